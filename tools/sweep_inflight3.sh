for cfg in "8192 6" "8192 8" "12288 4" "16384 3" "16384 4"; do
  set -- $cfg
  python3 bench.py --no-cpu-baseline --no-extra-loops --batch $1 --inflight $2 --steps 12 --warmup 2 2>/tmp/err.log | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
if not t: print('batch $1 inflight $2 FAILED'); sys.exit(0)
b=json.loads(t[-1])
print('batch $1 inflight $2', b['value'], b['ms_per_step'])"
  grep -h "HBM in use" /tmp/err.log | tail -1
done
