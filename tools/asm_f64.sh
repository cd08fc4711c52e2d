#!/bin/bash
# Developer tool: device assembly of the float64 sweep TU -> /tmp/f64.s (then tools/isa_loop_stats.py /tmp/f64.s <substr>)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/psa-simulation-ode-rk-mvp-dispersion_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$ROOT/include -I$CSRC -mllvm -amdgpu-sched-strategy=max-ilp "$@" \
    -S --cuda-device-only $CSRC/${SRC:-psa_rk4_f64.hip} -o ${OUT:-/tmp/f64.s} 2>&1 | grep -E "error|warning: v" | head
