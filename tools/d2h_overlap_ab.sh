export TMPDIR=/tmp
O=gpurun_out/r3pre; mkdir -p $O
for mode in overlap inline off; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES FETCH_SIZE -d $O/c3_$mode -- python3 bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --d2h $mode > $O/c3_$mode.json 2> $O/c3_$mode.err
  python3 - $O/c3_$mode $mode <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*_counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "rk4_sweep" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]]+=float(r["Counter_Value"])
print(sys.argv[2], [(int(v["SQ_WAVES"]), round(v["FETCH_SIZE"]*2048/1e6,1)) for v in acc.values()])
PY
done
for mode in overlap inline off; do python3 bench.py --config c3 --steps 4 --warmup 1 --no-cpu-baseline --d2h $mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode c3 ms/step %.3f kern %.3f resident %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['device_resident_ms_per_step']))"; done
for mode in overlap inline off; do python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --d2h $mode 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode c2 ms/step %.3f kern %.3f resident %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['device_resident_ms_per_step']))"; done
