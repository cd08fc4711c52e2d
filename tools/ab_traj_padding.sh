#!/bin/bash
# A/B on one box: every-step trajectories with the dense device layout (ld = N) and with the padded leading dimension
# (psa_traj_ld: + 4 352 B where the wave regions would lie a multiple of 2 MiB apart), next to the store-only probe.
O=gpurun_out/r3pad; mkdir -p $O
for c in "c2" "c4" "c5"; do
  for dense in 1 0; do
    PSA_TRAJ_DENSE=$dense python3 bench.py --mode trajectory --config $c --steps 100 --warmup 30 > $O/traj_${c}_dense$dense.json 2>/dev/null
    python3 - $O/traj_${c}_dense$dense.json "$c $([ $dense = 1 ] && echo 'dense ld = N     ' || echo 'padded ld        ')" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("%s kernel %.3f ms  %.0f GB/s  frac %.3f  err %.1e" % (sys.argv[2], r["kernel_ms_avg"], r["achieved"], r["frac"], d["verify"]["max_rel_err"]), flush=True)
PY
  done
done
PEAK_QUICK=1 tools/hbm_write_peak 262144 401 | grep -E "ceiling|^non-temporal stores, 0|PADDED ld = n \+ 272"
PEAK_QUICK=1 tools/hbm_write_peak 524288 201 | grep -E "ceiling|^non-temporal stores, 0|PADDED ld = n \+ 272"
