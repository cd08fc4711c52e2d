#!/bin/bash
# Round-2 profiling recipe (run on the GPU box through gpurun): for each bench configuration
#   1. rocprofv3 --kernel-trace --stats            -> per-kernel durations
#   2. rocprofv3 --kernel-trace --pmc <group>      -> one counter group per pass (SQ / GRBM / FETCH_SIZE / WRITE_SIZE)
# The program itself follows `--` (python3 bench.py ...): no env/bash hop under the profiler.
# Usage: tools/profile_r02.sh <out-dir> <cfg> [<cfg> ...]    cfg in c2 c3 c4 c5 c5one traj
set -e
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for cfg in "$@"; do
  case $cfg in
    c2)    ARGS="--config c2 --steps 5 --warmup 1 --no-cpu-baseline" ;;
    c3)    ARGS="--config c3 --steps 2 --warmup 1 --no-cpu-baseline" ;;
    c4)    ARGS="--config c4 --steps 2 --warmup 1 --no-cpu-baseline" ;;
    c5)    ARGS="--config c5 --steps 5 --warmup 1 --no-cpu-baseline" ;;
    c5one) ARGS="--config c5 --one-lane --steps 5 --warmup 1 --no-cpu-baseline" ;;
    traj)  ARGS="--mode trajectory --steps 5 --warmup 1" ;;
  esac
  echo "== $cfg: stats"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${cfg}_stats -- python3 bench.py $ARGS > $OUT/${cfg}_stats.log 2>&1
  for grp in "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES" \
             "grbm:GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    echo "== $cfg: pmc $name"
    rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d $OUT/${cfg}_pmc_$name -- python3 bench.py $ARGS > $OUT/${cfg}_pmc_$name.log 2>&1
  done
done
echo done
