#!/bin/bash
# Host-side AddressSanitizer + UBSan build of libpsa_hip (device code untouched: GPU sanitizers are not available
# on this pool) and a run of the plain-C GPU client against it.  Usage (on the GPU box): bash tools/host_sanitize.sh
set -euo pipefail
cd "$(dirname "$0")/.."
SRC=psa-simulation-ode-rk-mvp-dispersion_amd/csrc
OUT=${TMPDIR:-/tmp}/psa_san
mkdir -p "$OUT"
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
for f in psa_rk4_f64 psa_rk4_f32 psa_aux psa_capi; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC $SAN -c $SRC/$f.hip -o $OUT/$f.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $SAN -o $OUT/libpsa_hip.so $OUT/psa_rk4_f64.o $OUT/psa_rk4_f32.o $OUT/psa_aux.o $OUT/psa_capi.o
/opt/rocm/bin/hipcc -x c -std=c99 -Iinclude $SAN tests/c/abi_gpu_client.c -o $OUT/abi_gpu_client -L$OUT -lpsa_hip -Wl,-rpath,$OUT -lm
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $OUT/abi_gpu_client
