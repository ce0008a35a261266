python3 -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for pr in "" "-1,0" "0,-1" "-1,0,1" "-1,-1,0,0,1,1" "0"; do
  python3 bench.py --no-cpu-baseline --no-extra-loops --steps 16 --warmup 3 --stream-priorities="$pr" 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prio [$pr]', b['value'], b['ms_per_step'])"
done
