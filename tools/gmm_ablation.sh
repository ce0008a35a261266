#!/bin/bash
# Timing-only ablations of the scoring kernel (cdna_hip_programming.md §7 "Ablate"): rebuild libmfa_hip.so on the GPU box
# with one phase removed, time the bench's scoring launch, restore the real build.  Scores are wrong in these builds;
# only stage_ms_per_step.gmm matters.    bash tools/gmm_ablation.sh  (through gpurun, from the repo root)
set -eo pipefail
run() {
  MFA_GMM_DIAG=1 python bench.py --no-cpu-baseline --steps 3 $BENCH_ARGS 2>>gpurun_out/ablation.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['stage_ms_per_step']['gmm'])"
}
build() { MFA_HIPCC_FLAGS="$1" python -c "
import sys; sys.path.insert(0, '.')
from montreal_forced_aligner_amd import _lib; _lib.build_native(force=True)" 2>/dev/null; }
if [ "$1" = "bf16" ]; then
  export MFA_GMM_BF16=1
  run full-bf16
  for f in -DBF16_DIAG_NO_EPILOGUE -DBF16_DIAG_NO_FETCH -DBF16_DIAG_NO_FLUSH -DBF16_DIAG_NO_BARRIER -DBF16_DIAG_NO_INTERLEAVE \
           "-DBF16_DIAG_NO_EPILOGUE -DBF16_DIAG_NO_FETCH -DBF16_DIAG_NO_FLUSH" \
           "-DBF16_DIAG_NO_EPILOGUE -DBF16_DIAG_NO_FETCH -DBF16_DIAG_NO_FLUSH -DBF16_DIAG_NO_BARRIER"; do
    build "$f"; run "$f"
  done
else
  run full
  for f in -DGMM_DIAG_NO_EPILOGUE -DGMM_DIAG_NO_LOADS -DGMM_DIAG_NO_FLUSH "-DGMM_DIAG_NO_EPILOGUE -DGMM_DIAG_NO_LOADS -DGMM_DIAG_NO_FLUSH"; do
    build "$f"; run "$f"
  done
fi
build ""
