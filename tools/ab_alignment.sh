# A/B against a build of an EARLIER commit: make it with  git worktree add /tmp/prev <commit> && make -C /tmp/prev/psa-*/csrc && cp /tmp/prev/psa-*/libpsa_hip.so ab/libpsa_hip_prev.so
# aligned build (HEAD) vs the previous commit's build (ab/libpsa_hip_prev.so: blocks where they fell) on one box
for v in prev cur; do
  if [ $v = cur ]; then unset PSA_HIP_LIB; else export PSA_HIP_LIB=$PWD/ab/libpsa_hip_$v.so; fi
  for c in c2 c3 c4 c5; do python3 bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $c kern %.3f ms' % d['roofline']['kernel_ms_avg'])"; done
  for c in "c2" "c4" "c5" "c2 --split" "c5 --split"; do python3 bench.py --mode trajectory --config $c --steps 100 --warmup 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v traj $c kern %.3f ms' % d['roofline']['kernel_ms_avg'])"; done
done
