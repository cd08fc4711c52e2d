#!/bin/bash
# Host-side AddressSanitizer + UBSan build of libpsa_hip (device code untouched: GPU sanitizers are not available
# on this pool) and two runs of the plain-C GPU client against it: every host-buffer entry point incl. a two-chunk
# trajectory, then the same with a failure injected INSIDE the staging loop (-DPSA_FAULT_INJECTION + PSA_FAIL_CHUNK).
# Usage (on the GPU box): bash tools/host_sanitize.sh
set -euo pipefail
cd "$(dirname "$0")/.."
SRC=psa-simulation-ode-rk-mvp-dispersion_amd/csrc
OUT=${TMPDIR:-/tmp}/psa_san
mkdir -p "$OUT"
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
for f in psa_rk4_f64 psa_rk4_f32 psa_aux psa_dbeta psa_capi; do
  EXTRA=""
  [ $f = psa_dbeta ] && EXTRA="-ffp-contract=off"
  [ $f = psa_capi ] && EXTRA="-DPSA_FAULT_INJECTION"
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$SRC $SAN $EXTRA -c $SRC/$f.hip -o $OUT/$f.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $SAN -o $OUT/libpsa_hip.so $OUT/psa_rk4_f64.o $OUT/psa_rk4_f32.o $OUT/psa_aux.o $OUT/psa_dbeta.o $OUT/psa_capi.o
/opt/rocm/bin/hipcc -x c -std=c99 -Iinclude $SAN tests/c/abi_gpu_client.c -o $OUT/abi_gpu_client -L$OUT -lpsa_hip -Wl,-rpath,$OUT -lm
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== host ASan + UBSan build of $(git rev-parse --short HEAD 2>/dev/null || echo HEAD): client, clean run"
$OUT/abi_gpu_client
echo "== the same with the second staged chunk failing (PSA_FAIL_CHUNK=1), then a clean call on the same context"
PSA_FAIL_CHUNK=1 $OUT/abi_gpu_client
echo "== host sanitizers: no report"
