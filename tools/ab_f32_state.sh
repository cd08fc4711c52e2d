set -e
mkdir -p gpurun_out/r3f
for v in kahan delta; do
  if [ $v = kahan ]; then export PSA_HIP_LIB=$PWD/ab/libpsa_hip_head.so; else unset PSA_HIP_LIB; fi
  echo "== $v"
  python3 tools/f32_accuracy.py 2>&1 | grep -v amdgpu.ids
  python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c4 ms/step %.3f kern %.3f frac %.3f' % (d['ms_per_step'], r['kernel_ms_avg'], r['frac']), d['verify'])"
done 2>&1 | tee gpurun_out/r3f/f32_state_ab.log
unset PSA_HIP_LIB
python3 -m pytest tests -m gpu -q -k "float32 or f32 or packed or trajectory_layouts or six_wave_reference or soak or sharded" 2>&1 | tail -3
