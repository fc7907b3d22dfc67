#!/bin/bash
# usage (GPU box, repo root): QMCWALK_LIB=$PWD/build/variants/cuts/libqmcwalk.so tools/section_counts.sh <tag> [--bosons N]
# Per-section EXECUTED instruction counts of the VMC / DMC step kernels: two
# counter passes over tools/section_counts.py (counters only, never mixed with
# other trace domains), then the differences between successive cuts.
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
P2="SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/pmc$i -- python3 $R/tools/section_counts.py --plan $out/plan.json "$@" > $out/pmc$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
plan = json.load(open(os.path.join(root, 'plan.json')))
rows = defaultdict(lambda: defaultdict(dict))     # kernel -> dispatch -> counter -> value
for f in glob.glob(os.path.join(root, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = 'vmc' if 'vmc_step_kernel' in r['Kernel_Name'] else 'dmc' if 'dmc_evolve_kernel' in r['Kernel_Name'] else None
        if k is None:
            continue
        # dispatch ids differ between the two passes: rank them per pass
        rows[(k, f)][int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
seq = {'vmc': defaultdict(dict), 'dmc': defaultdict(dict)}
for (k, f), d in rows.items():
    for rank, did in enumerate(sorted(d)):
        seq[k][rank].update(d[did])
def table(kind, names, per_cut, first, waves_expected, title):
    cnts = ['SQ_INSTS_VALU', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_ADD_F64',
            'SQ_INSTS_VALU_TRANS_F64', 'SQ_INSTS_SALU', 'SQ_INSTS_BRANCH', 'SQ_INSTS_SMEM', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM']
    cum = []
    for i, nm in enumerate(names):
        acc = defaultdict(float)
        for j in range(per_cut):
            c = seq[kind][first + i * per_cut + j]
            w = c['SQ_WAVES']
            for x in cnts:
                acc[x] += c.get(x, 0.0) / w / per_cut
        cum.append(acc)
    print(f'== {title}: instructions per wavefront executed in each section (a wavefront = one walker)')
    print(f'  {"section":30s} ' + ' '.join(f'{x.replace("SQ_INSTS_", ""):>13s}' for x in cnts))
    for i in range(len(names) - 1):
        d = {x: cum[i + 1][x] - cum[i][x] for x in cnts}
        print(f'  {names[i]:30s} ' + ' '.join(f'{d[x]:13.1f}' for x in cnts))
    print(f'  {"TOTAL (cut at end)":30s} ' + ' '.join(f'{cum[-1][x]:13.1f}' for x in cnts))
nv = len(seq['vmc'])
first_vmc = nv - plan['vmc_tail_launches'] - len(plan['vmc']) * plan['vmc_launches_per_cut']
table('vmc', plan['vmc'], plan['vmc_launches_per_cut'], first_vmc, plan['walkers'],
      f"VMC N={plan['bosons']}, acceptance {plan['acceptance']:.3f}")
# DMC: per cut 2 full steps + 1 cut step
names = plan['dmc']
stride = plan['dmc_full_steps_before_each_cut'] + 1
# remap: the cut launch is the last of every group of `stride`
dm = seq['dmc']
sel = {}
for i in range(len(names)):
    sel[i] = dm[i * stride + stride - 1]
seq['dmc'] = defaultdict(dict, sel)
table('dmc', names, 1, 0, plan['maxw'], f"DMC N={plan['bosons']} (all launched wavefronts, incl. the ~6 % beyond the population that exit at once)")
PY
