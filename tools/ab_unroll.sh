# A/B on one box: split / quad kernels with one taken back-edge per FOUR steps (make EXTRA=-DPSA_UNROLL4 -> ab/libpsa_hip_unroll4.so)
# against the shipped two-step loop.  Kernel ms; small sweeps: wall / C-ABI call / kernel.
for rep in 1 2; do
for v in unroll4 cur; do
  if [ $v = cur ]; then unset PSA_HIP_LIB; else export PSA_HIP_LIB=$PWD/ab/libpsa_hip_$v.so; fi
  for c in c5; do python3 bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$rep $v $c kern %.3f ms' % d['roofline']['kernel_ms_avg'])"; done
  python3 tools/small_sweeps.py 2>/dev/null | grep "^G[123]" | sed "s/^/$rep $v /"
  python3 tools/split_cliff.py 2>/dev/null | awk '$2==100 || $2==4096 || $2==16384 || $2==32768' | sed "s/^/$rep $v cliff /"
done
done
