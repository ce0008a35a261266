# lag mode on overflow-heavier settings: a wider first beam puts more than 64 tokens into some frames; MFA_VIT_LAG=0 keeps the
# large tier per window, MFA_VIT_LAG=1 (default) sends such utterances to the from-scratch list pass
for beam in 14 20; do
  for lag in 1 0; do
    MFA_VIT_LAG=$lag python3 bench.py --no-cpu-baseline --no-extra-loops --batch 4096 --beam $beam --retry-beam $((beam*4)) --steps 8 --warmup 3 2>/tmp/e.log | python3 -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
b=json.loads(t[-1]) if t else None
print('beam $beam lag $lag', (b['value'], b['ms_per_step'], b['aligned_fraction']) if b else 'FAILED')"
  done
done
