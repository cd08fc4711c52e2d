for v in prev cap4 cap6 cap8 cap10 cap12 cur; do
  if [ $v = cur ]; then unset PSA_HIP_LIB; else export PSA_HIP_LIB=$PWD/ab/libpsa_hip_$v.so; fi
  python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c4 kern %.3f ms' % d['roofline']['kernel_ms_avg'], d['verify']['max_rel_err_a_end'])"
done
