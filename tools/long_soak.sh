set -e
O=gpurun_out/r3soak; mkdir -p $O
for seed in 20261005 97 5150; do
  timeout -k 10 380 python3 tools/soak_differential.py 330 $seed 2>&1 | grep -v amdgpu.ids > $O/soak_$seed.log || echo "SOAK $seed FAILED"
  tail -1 $O/soak_$seed.log
done
