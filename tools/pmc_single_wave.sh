# Where one resident wave per SIMD spends its cycles: tools/pmc_lanes.py (4 096 points x 20 000 steps, one / two / four lanes per
# point, save_every = 10) under rocprofv3 --pmc; per wave and z-step: VALU and SALU instructions, taken branches, wave cycles, waits.
export TMPDIR=/tmp
O=gpurun_out/${1:-r3sw}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN -d $O/p -- python3 tools/pmc_lanes.py > $O/run.log 2>&1
python3 - $O/p <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*_counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); names={}
for r in csv.DictReader(open(f)):
    if "rk4_sweep" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"]); names[r["Dispatch_Id"]]=r["Kernel_Name"]
for d,v in acc.items():
    w=v["SQ_WAVES"]*20000
    print(names[d][:64], "waves", int(v["SQ_WAVES"]), "per wave-step: VALU %.1f  SALU %.1f  branches %.2f (taken %.2f)  wave cycles %.0f  waiting %.0f  -> %.3f cycles per VALU instruction"
          % (v["SQ_INSTS_VALU"]/w, v["SQ_INSTS_SALU"]/w, v["SQ_INSTS_BRANCH"]/w, v.get("SQ_INSTS_CBRANCH_TAKEN",float('nan'))/w, 4*v["SQ_WAVE_CYCLES"]/w, 4*v["SQ_WAIT_ANY"]/w, 4*v["SQ_WAVE_CYCLES"]/v["SQ_INSTS_VALU"]))
PY
