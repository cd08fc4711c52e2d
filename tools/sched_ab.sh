mkdir -p gpurun_out/r2i
for t in maxilp default iterilp; do
  for c in c5 c2; do
    PSA_HIP_LIB=$PWD/ab/libpsa_hip_$t.so python bench.py --config $c --no-cpu-baseline --steps 5 > gpurun_out/r2i/${t}_$c.json 2> gpurun_out/r2i/${t}_$c.err || echo FAIL $t $c
    python -c "
import json; d=json.load(open('gpurun_out/r2i/${t}_$c.json')); print('$t $c kern_ms %.3f' % d['roofline']['kernel_ms_avg'])"
  done
done
