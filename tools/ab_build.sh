#!/bin/bash
# Developer tool: build A/B variants of libpsa_hip.so that differ only in how the float64 sweep TU is compiled.
#   tools/ab_build.sh <tag> [hipcc flags for psa_rk4_f64.hip ...]      ->  ab/libpsa_hip_<tag>.so
# e.g. tools/ab_build.sh ilp_w3 -mllvm -amdgpu-sched-strategy=max-ilp '-DPSA_SWEEP_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(3)))'
# The other objects are taken from the normal in-tree build (run `make -C .../csrc` first).  PSA_HIP_LIB=<path> selects
# a variant at run time (psa_amd._native).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/psa-simulation-ode-rk-mvp-dispersion_amd/csrc
TAG=$1; shift
mkdir -p $ROOT/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CSRC -Wall -Wno-unused-function "$@" \
    -c $CSRC/psa_rk4_f64.hip -o $ROOT/ab/psa_rk4_f64_$TAG.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/ab/libpsa_hip_$TAG.so $ROOT/ab/psa_rk4_f64_$TAG.o \
    $CSRC/psa_rk4_f32.o $CSRC/psa_aux.o $CSRC/psa_dbeta.o $CSRC/psa_capi.o
rm -f $ROOT/ab/psa_rk4_f64_$TAG.o
echo built ab/libpsa_hip_$TAG.so
