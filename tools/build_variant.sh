#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"
# Builds build/variants/<name>/libqmcwalk.so with extra -D flags for A/B runs
# (select it with QMCWALK_LIB=<path>; development tool).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
extra=${1:-}
out=$R/build/variants/$name
mkdir -p "$out"
make -s -C "$R/phd_qmclib_amd/csrc" -j8 OBJDIR="/tmp/qmc_variants/$name" OUT="$out/libqmcwalk.so" \
    HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-parameter -Wno-unused-value -fno-slp-vectorize $extra"
echo "$out/libqmcwalk.so"
