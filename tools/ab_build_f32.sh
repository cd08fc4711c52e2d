#!/bin/bash
# Developer tool: A/B variants of the float32 sweep TU:  tools/ab_build_f32.sh <tag> [hipcc flags ...]  ->  ab/libpsa_hip_<tag>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/psa-simulation-ode-rk-mvp-dispersion_amd/csrc
TAG=$1; shift
mkdir -p $ROOT/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CSRC -Wall -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp "$@" \
    -c $CSRC/psa_rk4_f32.hip -o $ROOT/ab/psa_rk4_f32_$TAG.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/ab/libpsa_hip_$TAG.so $ROOT/ab/psa_rk4_f32_$TAG.o \
    $CSRC/psa_rk4_f64.o $CSRC/psa_aux.o $CSRC/psa_dbeta.o $CSRC/psa_capi.o
rm -f $ROOT/ab/psa_rk4_f32_$TAG.o
echo built ab/libpsa_hip_$TAG.so
