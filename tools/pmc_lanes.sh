export TMPDIR=/tmp
O=gpurun_out/r3v; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU -d $O/p -- python3 tools/pmc_lanes.py > $O/run.log 2>&1
python3 - $O/p <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*_counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); names={}
for r in csv.DictReader(open(f)):
    if "rk4_sweep" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"]); names[r["Dispatch_Id"]]=r["Kernel_Name"]
for d,v in acc.items():
    print(names[d][:70], "waves", int(v["SQ_WAVES"]), "VALU per wave-step %.1f" % (v["SQ_INSTS_VALU"]/v["SQ_WAVES"]/20000))
PY
python3 tools/quad_lane_probe.py 2>&1 | grep -v amdgpu.ids
python3 tools/split_cliff.py 2>&1 | grep -v amdgpu.ids | head -12
