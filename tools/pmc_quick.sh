#!/bin/bash
# usage (GPU box, repo root): tools/pmc_quick.sh <tag> [env assignments...] -- workload args
# One counter pass (instruction counts, cycles) + nothing else; prints per-wave numbers.
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc1 -- python3 $R/tools/prof_workload.py "$@" > $out/pmc1.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
cnt = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, 'pmc1', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]
        cnt[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k in cnt:
    if 'step_kernel' not in k and 'evolve' not in k: continue
    c = {x: cnt[k][x] / max(1, n[k][x]) for x in cnt[k]}
    w = c['SQ_WAVES']
    print(f"{os.path.basename(root):12s} {k}: VALU/wave {c['SQ_INSTS_VALU']/w:.1f} SALU/wave {c['SQ_INSTS_SALU']/w:.1f} "
          f"LDS/wave {c['SQ_INSTS_LDS']/w:.1f} active_valu_cyc/wave {4*c['SQ_ACTIVE_INST_VALU']/w:.0f} "
          f"wave_cyc/wave {4*c['SQ_WAVE_CYCLES']/w:.0f}")
PY
