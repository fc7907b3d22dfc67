#!/bin/bash
# usage (GPU box, repo root): tools/round_profiles.sh <outdir under gpurun_out> [bench|pmc|all]
# The measurement records of a round at the kernels in the tree (development
# tool): the default bench line, the same command under `rocprofv3
# --kernel-trace --stats` (the program itself after `--`, counters never mixed
# with other trace domains), and the kernel-trace + counter passes of
# tools/profile.sh for the VMC step at N = 64 and the DMC step at N = 64 / 128.
set -u
tag=$1; what=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ "$what" = bench ] || [ "$what" = all ]; then
  python3 $R/bench.py --steps 20 --warmup 5 --fp32 --unrelaxed > $out/bench.json 2> $out/bench.err
  echo "bench rc=$?"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_prof -- \
      python3 $R/bench.py --steps 20 --warmup 5 --no-cpu > $out/bench_under_rocprof.json 2> $out/bench_prof.err)
  echo "bench under rocprofv3 rc=$?"
  f=$(ls $out/bench_prof/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $out/bench_kernel_stats.csv
fi
if [ "$what" = pmc ] || [ "$what" = all ]; then
  # the walkers start from the stationary state, as in bench.py (made outside
  # the profiler: 30 000 small launches are not what the counters are for)
  python3 $R/tools/make_stationary.py --bosons 64 --out /tmp/stationary_64.npy > $out/stationary.log 2>&1
  python3 $R/tools/make_stationary.py --bosons 128 --out /tmp/stationary_128.npy >> $out/stationary.log 2>&1
  PMC_LAST=16 $R/tools/profile.sh $tag/vmc64 vmc --walkers 262144 --steps 16 --launches 1 --equil 300 --start-file /tmp/stationary_64.npy > /dev/null 2>&1
  PMC_LAST=16 $R/tools/profile.sh $tag/dmc64 dmc --walkers 262144 --steps 16 --equil 100 --start-file /tmp/stationary_64.npy > /dev/null 2>&1
  PMC_LAST=16 $R/tools/profile.sh $tag/dmc128 dmc --bosons 128 --walkers 65536 --steps 16 --equil 100 --start-file /tmp/stationary_128.npy > /dev/null 2>&1
  for k in vmc64 dmc64 dmc128; do echo "== $k"; grep -A3 "kernel durations" $out/$k/summary.txt | head -4; tail -1 $out/$k/summary.txt; done
fi
