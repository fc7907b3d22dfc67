#!/bin/bash
# usage (GPU box, repo root): tools/round_records.sh <outdir under gpurun_out>
# Everything a round's records need at the kernels in the tree, in one call
# (development tool; ~5 minutes): the counter passes of tools/round_profiles.sh,
# the per-section counts and lifetime shares on stationary ensembles (needs
# build/variants/cuts and build/variants/timing: tools/build_variant.sh cuts
# "-DQMC_CUTS", timing "-DQMC_TIMING"), the 400-model fuzz of the sorted-row
# kernels, the whole GPU suite, the soak run and the shape bench.  The bench
# line itself comes from a SECOND call (tools/round_profiles.sh <tag> bench)
# after profiles/traffic.json was rebuilt from this one's summaries
# (tools/make_traffic.py): bench.py reports the traffic of the kernels it loads.
set -u
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
$R/tools/round_profiles.sh $tag pmc 2>&1 | tail -12
python3 $R/tools/make_stationary.py --bosons 64 --out /tmp/s64.npy > $out/stat.log 2>&1
python3 $R/tools/make_stationary.py --bosons 128 --out /tmp/s128.npy >> $out/stat.log 2>&1
QMCWALK_LIB=$R/build/variants/cuts/libqmcwalk.so $R/tools/section_counts.sh $tag/sec64 --start-file /tmp/s64.npy > $out/sec64.txt 2>&1
QMCWALK_LIB=$R/build/variants/cuts/libqmcwalk.so $R/tools/section_counts.sh $tag/sec128 --bosons 128 --start-file /tmp/s128.npy > $out/sec128.txt 2>&1
QMCWALK_LIB=$R/build/variants/timing/libqmcwalk.so python3 $R/tools/section_times.py --start-file /tmp/s64.npy > $out/times64.txt 2>&1
QMCWALK_LIB=$R/build/variants/timing/libqmcwalk.so python3 $R/tools/section_times.py --bosons 128 --start-file /tmp/s128.npy > $out/times128.txt 2>&1
cd $R
QMC_FUZZ_SPECS=400 python3 -m pytest tests/test_gpu_sorted_pins.py -x -q -k random > $out/fuzz.txt 2>&1; echo "fuzz rc=$?"
python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "pytest rc=$?"; tail -n 2 $out/tests.log
timeout -k 10 600 python3 tools/soak.py > $out/soak.txt 2>&1; echo "soak rc=$?"
python3 tools/shape_bench.py > $out/shape_bench.txt 2>&1; echo "shape rc=$?"
