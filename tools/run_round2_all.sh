#!/bin/bash
# Everything that gets measured for a round, in one GPU-box call (tools/profile_summary.py condenses it afterwards):
#   tests -m gpu, smoke, bench for every configuration, rocprofv3 stats + PMC for every configuration's kernel,
#   the store/FP64 ceiling probes and the trajectory steady-state probe.
set -e
O=gpurun_out/${1:-r2final}
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || true
tail -3 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || true
tail -1 $O/smoke.log
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config c3 --steps 4 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --config c4 --steps 4 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
python bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --config c5 --one-lane --no-cpu-baseline > $O/bench_c5_one_lane.json 2> $O/bench_c5_one_lane.err
python bench.py --mode trajectory --steps 100 --warmup 30 > $O/bench_traj.json 2> $O/bench_traj.err
echo "== benches done"
bash tools/profile_r02.sh $O/prof c2 c3 c4 c5 c5one traj > $O/profile.log 2>&1
echo "== profiles done"
(timeout -k 10 100 tools/hbm_write_peak; timeout -k 10 100 tools/fp64_store_mix; timeout -k 10 150 python tools/traj_clock_probe.py 2>&1 | grep -v amdgpu.ids) > $O/store_ceiling.log 2>&1
timeout -k 10 100 python tools/config1_latency.py 2>&1 | grep -v amdgpu.ids > $O/config1.log
cat $O/config1.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.3f kern_ms %.3f frac %.3f" % (d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["frac"]))
PY
done
