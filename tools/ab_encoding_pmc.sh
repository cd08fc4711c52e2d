set -e
export TMPDIR=/tmp
for v in plain cur; do
  if [ $v = cur ]; then unset PSA_HIP_LIB; else export PSA_HIP_LIB=$PWD/ab/libpsa_hip_$v.so; fi
  for cfg in c2 c5; do
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES -d gpurun_out/r3encpmc/${v}_${cfg}_sq -- python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3encpmc/${v}_${cfg}_sq.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE -d gpurun_out/r3encpmc/${v}_${cfg}_grbm -- python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3encpmc/${v}_${cfg}_grbm.log 2>&1
  done
done
echo done
