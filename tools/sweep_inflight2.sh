for cfg in "4096 6" "4096 8" "8192 4" "4096 6" "4096 8" "8192 4" "8192 6"; do
  set -- $cfg
  python3 bench.py --no-cpu-baseline --no-extra-loops --batch $1 --inflight $2 --steps 16 --warmup 3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $1 inflight $2', b['value'], b['ms_per_step'])"
done
