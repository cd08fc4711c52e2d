set -e
mkdir -p gpurun_out/r2c
O=gpurun_out/r2c
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config c3 --steps 4 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --config c4 --steps 4 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --config c5 --one-lane --no-cpu-baseline > $O/bench_c5_one_lane.json 2> $O/bench_c5_one_lane.err
python bench.py --mode trajectory > $O/bench_traj.json 2> $O/bench_traj.err
for c in c2 c4 c5; do
PSA_BENCH_DIST_ON_ONE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${c}_rccl1.json 2> $O/bench_${c}_rccl1.err
done
for f in $O/*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f'))
r=d['roofline']
print('  value %.4g ms/step %.3f kern_ms %.3f frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms_avg'], r['frac']), d.get('verify'))
print('  cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('sample'))
"; done
