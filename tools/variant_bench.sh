#!/bin/bash
# A/B a compile-time variant of libmfa_hip.so on the GPU box:  bash tools/variant_bench.sh "<hipcc flags>" [bench args…]
# prints stage_ms_per_step of the default build and of the variant, then restores the default build.
set -eo pipefail
flags="$1"; shift || true
run() { python bench.py --no-cpu-baseline --steps 4 "$@" 2>>gpurun_out/variant.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['stage_ms_per_step'])"; }
build() { MFA_HIPCC_FLAGS="$1" python -c "
import sys; sys.path.insert(0, '.')
from montreal_forced_aligner_amd import _lib; _lib.build_native(force=True)" 2>/dev/null; }
tag=default; run "$@"
build "$flags"; tag="variant[$flags]"; run "$@"
build ""; tag=default-again; run "$@"
