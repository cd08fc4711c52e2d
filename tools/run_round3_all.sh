#!/bin/bash
# Round 3, everything measured on ONE box in one call (profiles come from tools/profile_r03.sh in separate calls):
#   bench JSON lines for every configuration incl. the per-step-check headline and all trajectory layouts, the store-only
#   ceilings of the same box, the lane-layout A/B around the cost model's switch points, the small-sweep latencies,
#   and the host ASan/UBSan run.
set -e
O=gpurun_out/${1:-r3all}
mkdir -p $O
python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
python3 bench.py --block-check --no-cpu-baseline > $O/bench_c2_block_check.json 2> $O/bench_c2_block_check.err
python3 bench.py --config c3 --steps 4 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
python3 bench.py --config c4 --steps 4 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err
python3 bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
python3 bench.py --config c5 --one-lane --no-cpu-baseline > $O/bench_c5_one_lane.json 2> $O/bench_c5_one_lane.err
echo "== summary benches done"
python3 bench.py --mode trajectory --steps 100 --warmup 30 > $O/bench_traj.json 2> $O/bench_traj.err
python3 bench.py --mode trajectory --config c4 --steps 100 --warmup 30 > $O/bench_traj_f32.json 2> $O/bench_traj_f32.err
python3 bench.py --mode trajectory --config c5 --steps 60 --warmup 30 > $O/bench_traj_six.json 2> $O/bench_traj_six.err
python3 bench.py --mode trajectory --config c2 --split --steps 100 --warmup 30 > $O/bench_traj_split4.json 2> $O/bench_traj_split4.err
python3 bench.py --mode trajectory --config c5 --split --steps 100 --warmup 30 > $O/bench_traj_split6.json 2> $O/bench_traj_split6.err
echo "== trajectory benches done"
(timeout -k 10 100 tools/hbm_write_peak; timeout -k 10 100 tools/hbm_write_peak 524288 201; timeout -k 10 100 tools/hbm_write_peak 32768 3201;
 timeout -k 10 100 tools/fp64_store_mix) > $O/store_ceiling.log 2>&1
echo "== ceilings done"
timeout -k 10 300 python3 tools/small_sweeps.py 2>&1 | grep -v amdgpu.ids > $O/small_sweeps.log
timeout -k 10 200 python3 tools/driver_overheads.py 2>&1 | grep -v amdgpu.ids > $O/driver_overheads.log
timeout -k 10 100 python3 tools/config1_latency.py 2>&1 | grep -v amdgpu.ids > $O/config1.log
echo "== latency probes done"
timeout -k 10 400 python3 tools/split_cliff.py 2>&1 | grep -v amdgpu.ids > $O/split_cliff.log
echo "== layout A/B done"
TMPDIR=/tmp timeout -k 10 600 bash tools/host_sanitize.sh > $O/host_sanitizer.log 2>&1 || echo "SANITIZER RUN FAILED"
tail -3 $O/host_sanitizer.log
for f in $O/bench_*.json; do python3 - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1].split("/")[-1], "value %.4g ms/step %.3f kern_ms %.3f frac %.3f" % (d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["frac"]),
      "issue_nominal %s" % r.get("issue_frac_nominal"))
PY
done
