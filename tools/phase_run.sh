#!/bin/bash
# Decoder phase accounting on the GPU box: rebuild with -DVIT_STAMPS (box copy only), one bench step per batch size,
# print cycles per phase.  usage: tools/phase_run.sh OUTDIR [batch ...]
out=$1; shift
mkdir -p "$out"
MFA_HIPCC_FLAGS=-DVIT_STAMPS python -c "from montreal_forced_aligner_amd import _lib; _lib.build_native(force=True)" > "$out/build.log" 2>&1 || { tail -20 "$out/build.log"; exit 1; }
for b in "${@:-4096}"; do
  MFA_VIT_STAMPS=$out/stamps_$b.npy timeout -k 10 400 python bench.py --steps 2 --warmup 1 --inflight 1 --batch $b --no-cpu-baseline --no-extra-loops > "$out/phase_$b.json" 2> "$out/phase_$b.err"
  rc=$?
  if grep -q "Memory access fault" "$out/phase_$b.err"; then echo "batch $b: GPU FAULT"; exit 9; fi
  if [ $rc -ne 0 ]; then echo "batch $b rc=$rc"; tail -5 "$out/phase_$b.err"; exit $rc; fi
  echo "== batch $b"
  python tools/viterbi_phases.py $out/stamps_$b.npy
  python - "$out/phase_$b.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("stages", d["stage_ms_per_step"])
PY
done
