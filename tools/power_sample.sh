# Samples rocm-smi (socket power, shader clock) while a long run of a bench configuration is in flight: is the clock the chip
# holds under this load a power cap?   bash tools/power_sample.sh [c2|c3|c4|c5|traj]
CFG=${1:-c3}
case $CFG in
  c3) ARGS="--config c3 --steps 12 --warmup 1 --no-cpu-baseline" ;;
  c4) ARGS="--config c4 --steps 16 --warmup 1 --no-cpu-baseline" ;;
  traj) ARGS="--mode trajectory --steps 6000 --warmup 30" ;;
  *) ARGS="--config $CFG --steps 200 --warmup 2 --no-cpu-baseline" ;;
esac
echo "idle: $(rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i 'Power (W)\|sclk' | sed 's/.*: //' | tr '\n' ' ')"
python3 bench.py $ARGS > /tmp/power_bench.json 2>/dev/null &
BPID=$!
sleep 6
for i in 1 2 3 4 5; do
  echo "$CFG in flight: $(rocm-smi --showpower --showclocks 2>/dev/null | grep -i 'Power (W)\|sclk' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 1
done
wait $BPID
python3 -c "import json; d=json.load(open('/tmp/power_bench.json')); print('$CFG kernel %.3f ms' % d['roofline']['kernel_ms_avg'])"
