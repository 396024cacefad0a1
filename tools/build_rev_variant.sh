#!/bin/bash
# Developer tool: builds the library of another git revision into gpurun_variants/lib_<name>.so (A/B runs on one GPU
# box: TAMCMC_ACCEL_LIB=gpurun_variants/lib_<name>.so).  usage: tools/build_rev_variant.sh <rev> <name>
set -e
REV=${1:-HEAD}; NAME=${2:-base}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
git -C "$ROOT" archive "$REV" tamcmc-c-_amd/csrc include | tar -x -C "$W"
make -C "$W/tamcmc-c-_amd/csrc" -j8 ../libtamcmc_accel.so > "$W/build.log" 2>&1 || { tail -20 "$W/build.log"; exit 1; }
mkdir -p "$ROOT/gpurun_variants"
cp "$W/tamcmc-c-_amd/libtamcmc_accel.so" "$ROOT/gpurun_variants/lib_$NAME.so"
rm -rf "$W"
echo "built gpurun_variants/lib_$NAME.so from $REV"
