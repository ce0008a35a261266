#!/bin/bash
# Short bench runs of several variants on the GPU box; prints stage times per variant.
# usage: tools/variant_bench.sh OUTDIR "name|bench args[|ENV=1 ...]" ...
out=$1; shift
mkdir -p "$out"
for v in "$@"; do
  IFS='|' read -r name args envs <<< "$v"     # "name|bench args|VAR=1 VAR2=x" (third field optional)
  env $envs timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-loops $args > "$out/$name.json" 2> "$out/$name.err"
  rc=$?
  if grep -q "Memory access fault" "$out/$name.err"; then echo "$name: GPU FAULT"; exit 9; fi
  if [ $rc -ne 0 ]; then echo "$name rc=$rc"; tail -5 "$out/$name.err"; if [ $rc -gt 1 ]; then exit $rc; fi; continue; fi
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:28s} value {d['value']:9.0f}  ms/step {d['ms_per_step']:7.3f}  stages {d['stage_ms_per_step']}  launches(vit) {d['roofline'].get('launches_per_step')}")
PY
  grep -h "warmup done\|graph states" "$out/$name.err" | cut -c1-200
done
