# A/B on one box: the library built with the encoding-alignment step (default) against the plain hipcc build
#   make -C <copy of csrc> ENCODE=0  ->  ab/libpsa_hip_plain.so      (PSA_HIP_LIB selects the library)
# Prints kernel ms per configuration, alternating the two builds twice to expose drift.
for rep in 1 2; do
for v in plain cur; do
  if [ $v = cur ]; then unset PSA_HIP_LIB; else export PSA_HIP_LIB=$PWD/ab/libpsa_hip_$v.so; fi
  for c in c2 c4 c5 "c5 --one-lane" c3; do python3 bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$rep $v $c kern %.3f ms' % d['roofline']['kernel_ms_avg'])"; done
  for c in "c2" "c4" "c5" "c2 --split" "c5 --split"; do python3 bench.py --mode trajectory --config $c --steps 100 --warmup 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$rep $v traj $c kern %.3f ms' % d['roofline']['kernel_ms_avg'])"; done
  python3 tools/small_sweeps.py 2>/dev/null | grep "^G[123]" | sed "s/^/$rep $v /"
done
done
