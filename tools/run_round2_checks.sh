set -e
mkdir -p gpurun_out/r2e
O=gpurun_out/r2e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || true
tail -4 $O/pytest_gpu.log
for c in c2 c5; do
PSA_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --config $c --steps 3 --warmup 1 > $O/bench_${c}_gloo2.json 2> $O/bench_${c}_gloo2.err || echo "FAILED $c"
python -c "
import json; d=json.load(open('$O/bench_${c}_gloo2.json')); print('$c', d['n_gpus'], '%.4g' % d['value'], d['ms_per_step'], d['config']['parallelism'], d['verify']['max_rel_err_a_end'])"
done
timeout -k 10 200 python bench.py --mode trajectory > $O/bench_traj.json 2> $O/bench_traj.err
python -c "
import json; d=json.load(open('$O/bench_traj.json')); print('traj', d['roofline']['achieved'], d['roofline']['kernel_ms_avg'])"
