#!/bin/bash
# A/B of two builds of libmfa_hip.so on the same box: bash tools/ab_value.sh <other.so> [rounds] → value / single-batch stage ms, alternating
set -eo pipefail
other=$1; rounds=${2:-2}
for i in $(seq $rounds); do
  for so in "" "$other"; do
    MFA_HIP_SO=$so python3 bench.py --no-cpu-baseline $AB_FLAGS --steps 12 --warmup 3 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${so:-default}', b['value'], b['ms_per_step'], (b.get('single_batch_in_flight') or {}).get('stage_ms_per_step'))"
  done
done
