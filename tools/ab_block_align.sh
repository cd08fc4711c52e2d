for v in cur a5 a6 cur; do
  if [ $v = cur ]; then unset PSA_HIP_LIB; else export PSA_HIP_LIB=$PWD/ab/libpsa_hip_$v.so; fi
  for c in c2 c4 c5; do python3 bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $c kern %.3f ms' % d['roofline']['kernel_ms_avg'])"; done
done
