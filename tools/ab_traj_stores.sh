set -e
O=gpurun_out/r3ab
mkdir -p $O
for v in head spread wb spread_wb; do
  for cfg in "c2 --split" "c5 --split" "c2" "c5"; do
    tag=$(echo $cfg | tr -d ' -')
    PSA_HIP_LIB=$PWD/ab/libpsa_hip_$v.so python3 bench.py --mode trajectory --config $cfg --steps 100 --warmup 30 > $O/${v}_$tag.json 2> $O/${v}_$tag.err
    python3 - "$O/${v}_$tag.json" "$v $cfg" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("%-22s kern_ms %.3f  %.0f GB/s  err %.1e" % (sys.argv[2], r["kernel_ms_avg"], r["achieved"], d["verify"]["max_rel_err"]), flush=True)
PY
  done
done
python3 tools/small_sweeps.py 2>&1 | grep -v amdgpu.ids | tee $O/small_sweeps_spin.log
