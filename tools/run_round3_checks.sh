#!/bin/bash
# Round 3, the long checks of HEAD's build on ONE box (bench numbers come from tools/run_round3_all.sh):
#   randomized differential soaks against the oracle (two seeds), device dbeta fuzz on 1e6 points, 16.8 M points in one launch,
#   bench.py --gpus 2 through its own launcher for c2 and c5 (gloo on the one-GPU box: two ranks share the GPU),
#   the reference's three scenarios through the package.
set -e
O=gpurun_out/${1:-r3checks}
mkdir -p $O
SOAK=${SOAK_SECONDS:-420}
timeout -k 10 $((SOAK + 60)) python3 tools/soak_differential.py $SOAK 31337 2>&1 | grep -v amdgpu.ids > $O/soak_31337.log || echo "SOAK 31337 FAILED"
tail -1 $O/soak_31337.log
timeout -k 10 $((SOAK + 60)) python3 tools/soak_differential.py $SOAK 4242 2>&1 | grep -v amdgpu.ids > $O/soak_4242.log || echo "SOAK 4242 FAILED"
tail -1 $O/soak_4242.log
timeout -k 10 300 python3 tools/dbeta_fuzz.py 2>&1 | grep -v amdgpu.ids > $O/dbeta_fuzz.log || echo "DBETA FUZZ FAILED"
tail -2 $O/dbeta_fuzz.log
timeout -k 10 300 python3 tools/big_n_check.py 2>&1 | grep -v amdgpu.ids > $O/big_n.log || echo "BIG N FAILED"
tail -2 $O/big_n.log
for c in c2 c5; do
  PSA_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${c}_self_launch_gloo2.json 2> $O/bench_${c}_self_launch_gloo2.err || echo "SELF-LAUNCH $c FAILED"
  python3 -c "
import json; d=json.load(open('$O/bench_${c}_self_launch_gloo2.json')); print('$c self-launched', d['n_gpus'], 'ranks: %.4g upd/s' % d['value'], '%.2f ms/step' % d['ms_per_step'], d['config']['parallelism'], 'max rel err vs oracle', d['verify']['max_rel_err_a_end'])"
done
timeout -k 10 200 python3 examples/reference_scenarios.py 2>&1 | grep -v amdgpu.ids > $O/reference_scenarios.log || echo "SCENARIOS FAILED"
tail -4 $O/reference_scenarios.log
