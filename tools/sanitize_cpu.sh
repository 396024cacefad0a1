#!/bin/bash
# Developer tool (this container, no GPU): the host-side C++ of the library (sampler + priors, readers, writers) built
# with AddressSanitizer + UndefinedBehaviorSanitizer and driven by the CPU test-suite.  GPU sanitizers are not available
# on the pool; the device objects are linked in as they are.  Usage: tools/sanitize_cpu.sh [pytest args]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/tamcmc-c-_amd/csrc
OUT=$SRC/variants/asan
mkdir -p "$OUT"
make -C "$SRC" >/dev/null
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
for f in tamcmc_sampler tamcmc_io tamcmc_outputs; do
    g++ $SAN -std=c++17 -fPIC -I"$ROOT/include" -I"$SRC" -Wall -ffp-contract=off -pthread -c "$SRC/$f.cpp" -o "$OUT/$f.o"
done
g++ $SAN -shared -fPIC -o "$OUT/libtamcmc_accel.so" "$OUT"/tamcmc_sampler.o "$OUT"/tamcmc_io.o "$OUT"/tamcmc_outputs.o \
    "$SRC"/tamcmc_api.o "$SRC"/tamcmc_setup.o "$SRC"/tamcmc_eval.o "$SRC"/tamcmc_fused.o "$SRC"/tamcmc_backward.o \
    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -pthread
cd "$ROOT"
# python itself is not instrumented: preload the runtime, and leave leak checking off (the interpreter never frees all)
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
TAMCMC_ACCEL_LIB="$OUT/libtamcmc_accel.so" \
python -m pytest tests -q -x -m "not gpu" -p no:cacheprovider "$@"
