#!/bin/bash
# Band-kernel phase accounting on the GPU box: rebuild with -DGMM_BAND_STAMPS (box copy only), one bench run per variant.
# usage: tools/gmm_phase_run.sh OUTDIR "name|ENV=1 ..." ...
out=$1; shift
mkdir -p "$out"
MFA_HIPCC_FLAGS="-DGMM_BAND_STAMPS $GMM_DIAG_FLAGS" python -c "from montreal_forced_aligner_amd import _lib; _lib.build_native(force=True)" > "$out/build.log" 2>&1 || { tail -20 "$out/build.log"; exit 1; }
for v in "$@"; do
  IFS='|' read -r name envs <<< "$v"
  env $envs MFA_GMM_STAMPS=1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extra-loops > "$out/$name.json" 2> "$out/$name.err"
  rc=$?
  if grep -q "Memory access fault" "$out/$name.err"; then echo "$name: GPU FAULT"; exit 9; fi
  if [ $rc -ne 0 ]; then echo "$name rc=$rc"; tail -5 "$out/$name.err"; exit $rc; fi
  echo "== $name: $(grep 'band kernel per wavefront' "$out/$name.err")"
  python - "$out/$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   stages", d["stage_ms_per_step"])
PY
done
