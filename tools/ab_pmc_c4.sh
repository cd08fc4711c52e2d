# A/B against a build of an EARLIER commit: make it with  git worktree add /tmp/prev <commit> && make -C /tmp/prev/psa-*/csrc && cp /tmp/prev/psa-*/libpsa_hip.so ab/libpsa_hip_prev.so
export TMPDIR=/tmp
O=gpurun_out/r3r; rm -rf $O; mkdir -p $O
for v in prev cur; do
  if [ $v = prev ]; then export PSA_HIP_LIB=$PWD/ab/libpsa_hip_prev.so; else unset PSA_HIP_LIB; fi
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $O/$v -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/$v.json 2> $O/$v.err
  python3 - $O/$v $v <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*_counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "rk4_sweep" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"])
v=list(acc.values())[-1]
print(sys.argv[2], {k: round(x/1024/1e6,2) for k,x in v.items() if k!="SQ_WAVES"}, "per wave-step; waves", v["SQ_WAVES"])
PY
done
