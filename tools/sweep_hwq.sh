for cfg in "8 6 1" "16 6 1" "24 6 1" "16 8 1" "16 4 1" "16 6 0" "8 6 0"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 MFA_VIT_SIDE=$3 python3 bench.py --no-cpu-baseline --no-extra-loops --inflight $2 --steps 16 --warmup 3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hwq $1 inflight $2 side $3', b['value'], b['ms_per_step'])"
done
