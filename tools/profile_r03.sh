#!/bin/bash
# Round-3 profiling recipe (run on the GPU box through gpurun): for each bench configuration
#   1. rocprofv3 --kernel-trace --stats            -> per-kernel durations
#   2. rocprofv3 --kernel-trace --pmc <group>      -> one counter group per pass: SQ / GRBM / FETCH_SIZE / WRITE_SIZE / FLOPS
#      (FLOPS = SQ_INSTS_VALU_{FMA,MUL,ADD}_F64|F32 + the gfx950 SQ_INSTS_VALU_FLOPS_FP64|FP32 counter: executed work)
# The program itself follows `--` (python3 bench.py ...): no env/bash hop under the profiler; counters never share a run with
# a trace domain other than --kernel-trace.
# Usage: tools/profile_r03.sh <out-dir> <cfg> [<cfg> ...]
#   cfg in  c2 c2blk (--block-check) c3 c4 c5 c5one  traj trajf32 traj6 traj4s traj6s
set -e
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for cfg in "$@"; do
  F64=1
  case $cfg in
    c2)      ARGS="--config c2 --steps 5 --warmup 1 --no-cpu-baseline" ;;
    c2blk)   ARGS="--config c2 --block-check --steps 5 --warmup 1 --no-cpu-baseline" ;;
    c3)      ARGS="--config c3 --steps 2 --warmup 1 --no-cpu-baseline" ;;
    c4)      ARGS="--config c4 --steps 2 --warmup 1 --no-cpu-baseline"; F64=0 ;;
    c5)      ARGS="--config c5 --steps 5 --warmup 1 --no-cpu-baseline" ;;
    c5one)   ARGS="--config c5 --one-lane --steps 5 --warmup 1 --no-cpu-baseline" ;;
    traj)    ARGS="--mode trajectory --steps 40 --warmup 30" ;;
    trajf32) ARGS="--mode trajectory --config c4 --steps 40 --warmup 30"; F64=0 ;;
    traj6)   ARGS="--mode trajectory --config c5 --steps 40 --warmup 30" ;;
    traj4s)  ARGS="--mode trajectory --config c2 --split --steps 40 --warmup 30" ;;
    traj6s)  ARGS="--mode trajectory --config c5 --split --steps 40 --warmup 30" ;;
    *) echo "unknown cfg $cfg"; exit 2 ;;
  esac
  if [ $F64 = 1 ]; then FL="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FLOPS_FP64"
  else FL="SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT"; fi
  echo "== $cfg: stats"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${cfg}_stats -- python3 bench.py $ARGS > $OUT/${cfg}_stats.log 2>&1
  for grp in "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES" \
             "grbm:GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "flops:$FL"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    echo "== $cfg: pmc $name"
    rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d $OUT/${cfg}_pmc_$name -- python3 bench.py $ARGS > $OUT/${cfg}_pmc_$name.log 2>&1
  done
done
echo done
