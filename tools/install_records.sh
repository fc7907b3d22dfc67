#!/bin/bash
# usage (repo root, after the gpurun call of tools/round_records.sh <tag> came
# back): tools/install_records.sh <tag> <round prefix, e.g. r04>
# Copies the summaries into profiles/ and rebuilds profiles/traffic.json
# (development tool).  The bench line follows from a second GPU call:
# tools/round_profiles.sh <tag2> bench, then copy its three files.
set -eu
T=$1; P=$2
cd "$(dirname "$0")/.."
for k in vmc64 dmc64 dmc128; do cp gpurun_out/$T/$k/summary.txt profiles/${P}_${k}_pmc_summary.txt; done
python3 tools/make_traffic.py N64,profiles/${P}_vmc64_pmc_summary.txt,vmc_step_kernel,chain_step \
    N64,profiles/${P}_dmc64_pmc_summary.txt,dmc_evolve_kernel,walker_step,1.0674 \
    N128,profiles/${P}_dmc128_pmc_summary.txt,dmc_evolve_kernel,walker_step,1.0703 > profiles/traffic.json
(grep '^#' profiles/${P}_dynamic_sections.txt; echo; cat gpurun_out/$T/sec64.txt gpurun_out/$T/sec128.txt; echo
 grep -v amdgpu gpurun_out/$T/times64.txt; grep -v amdgpu gpurun_out/$T/times128.txt) > /tmp/ds.$$ && mv /tmp/ds.$$ profiles/${P}_dynamic_sections.txt
grep -v amdgpu gpurun_out/$T/soak.txt > profiles/${P}_soak.txt
grep -v amdgpu gpurun_out/$T/shape_bench.txt > profiles/${P}_shape_bench.txt     # (stationary ensembles)
[ -f gpurun_out/fastmath_report.json ] && cp gpurun_out/fastmath_report.json profiles/${P}_fastmath_report.json
grep 'valu\|salu\|sha' profiles/traffic.json
