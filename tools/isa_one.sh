#!/bin/bash
# usage: tools/isa_one.sh '<kernel instantiation>' [out.s]
#   e.g. tools/isa_one.sh 'vmc_step_kernel<64, 1, false, false, true>'
# Compiles ONE instantiation of a kernel of qmc_kernels.h for gfx950 (device
# only) and writes its ISA; prints the resource usage and instruction counts.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
inst=$1
out=${2:-$R/build_tmp/one.s}
mkdir -p "$(dirname "$out")"
tmp=$R/build_tmp/isa_one.hip
cat > "$tmp" <<SRC
#include "$R/phd_qmclib_amd/csrc/qmc_kernels.h"
template __global__ void $inst(${3:-const DevModel *, VmcArgs});
SRC
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -DQMC_SECTIONS ${QMC_EXTRA:-} \
    -Rpass-analysis=kernel-resource-usage -Wno-unused-value -fno-slp-vectorize -o "$out" "$tmp" 2>&1 |
    grep -E "remark:.*(Function Name|VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize)" | sed 's/.*remark: //; s/ \[-Rpass.*//' |
    awk -v k="${inst%%<*}" '/Function Name/ { show = index($0, k) > 0 } show' || true
python3 "$R/tools/isa_count.py" "$out" --only "${inst%%<*}" --ops
