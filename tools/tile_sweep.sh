#!/bin/bash
# Builds the N = 512 tile variants (own particles per pass x LDS table copies)
# into build/variants/ (run here), or runs them on the GPU box (`run`).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-build}" = build ]; then
    for pa in 2 4 8; do for dup in 1 2; do
        "$R/tools/build_variant.sh" pa${pa}_dup${dup} "-DQMC_PA8=$pa -DQMC_DUP8=$dup" | tail -1
    done; done
else
    for pa in 2 4 8; do for dup in 1 2; do
        lib=$R/build/variants/pa${pa}_dup${dup}/libqmcwalk.so
        [ -f "$lib" ] || continue
        QMCWALK_LIB=$lib python3 "$R/tools/tile_sweep.py" --tag "tile=$((64*pa)) dup=$dup" 2>&1 | grep -v amdgpu.ids
        QMCWALK_LIB=$lib python3 "$R/tools/tile_sweep.py" --fast --tag "tile=$((64*pa)) dup=$dup" 2>&1 | grep -v amdgpu.ids
    done; done
fi
