for v in prev cur; do
  if [ $v = prev ]; then export PSA_HIP_LIB=$PWD/ab/libpsa_hip_prev.so; else unset PSA_HIP_LIB; fi
  for c in c2 c4 c5; do python3 bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $c kern %.3f ms' % d['roofline']['kernel_ms_avg'])"; done
done
