#!/bin/bash
# usage (GPU box, repo root): tools/pmc_cmp.sh <tag> [workload args...]
# Three counter passes over tools/prof_workload.py (counters only, never mixed
# with other trace domains) for the library QMCWALK_LIB selects; prints the
# per-wavefront numbers of the walker kernels from the last 16 dispatches.
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES"
P3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F64"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc$i -- python3 $R/tools/prof_workload.py "$@" > $out/pmc$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for f in glob.glob(os.path.join(root, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]
        vals[k][r['Counter_Name']].append((int(r.get('Dispatch_Id', 0) or 0), float(r['Counter_Value'])))
for f in glob.glob(os.path.join(root, 'pmc1', '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]
        dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k in vals:
    if 'step_kernel' not in k and 'evolve' not in k: continue
    c = {}
    for name, v in vals[k].items():
        v.sort(); v = v[-16:]
        c[name] = sum(x for _, x in v) / len(v)
    w = c['SQ_WAVES']
    d = dur[k][-16:]
    print(f"{os.path.basename(root)} {k}: {sum(d)/len(d)/1e3:.1f} us/launch, {w:.0f} waves")
    for name in sorted(c):
        if name == 'SQ_WAVES': continue
        print(f"    {name:30s} {c[name]/w:12.1f} per wave")
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8
    if cyc:
        print(f"    kernel cycles (GRBM/8) {cyc:.0f}; per wave and SIMD {cyc*1024/w:.0f}; "
              f"VALU busy {4*c['SQ_ACTIVE_INST_VALU']/1024/cyc:.3f}  scalar busy {4*c.get('SQ_ACTIVE_INST_SCA',0)/1024/cyc:.3f}")
PY
