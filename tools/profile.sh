#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile.sh <tag> <workload args...>
# Runs a kernel-trace pass and separate PMC passes (counters never combined with
# other trace domains) and leaves CSVs under gpurun_out/<tag>/.
set -u
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
out=$R/gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/prof_workload.py "$@" > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc1 -- python3 $R/tools/prof_workload.py "$@" > $out/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $out/pmc2 -- python3 $R/tools/prof_workload.py "$@" > $out/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc3 -- python3 $R/tools/prof_workload.py "$@" > $out/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc4 -- python3 $R/tools/prof_workload.py "$@" > $out/pmc4.log 2>&1
python3 $R/tools/pmc_summary.py $out ${PMC_LAST:+--last $PMC_LAST} > $out/summary.txt 2>&1
# which kernels were measured (tools/make_traffic.py -> profiles/traffic.json)
grep -h '^kernel_source_sha=' $out/pmc4.log | tail -1 >> $out/summary.txt
cat $out/summary.txt
