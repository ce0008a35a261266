#!/bin/bash
# Profile pass for the bench workload on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>        → gpurun_out/prof_<tag>/{stats,pmc_sq,pmc_fetch,pmc_write}/… + summary json
# rocprofv3 rules of this pool: PMC passes carry --kernel-trace only; python3 directly after "--".
set -eo pipefail
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bench="$root/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 $bench > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
echo "[profile] kernel stats done"
small="--steps 1 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
  --output-format csv -d "$out/pmc_sq" -- python3 $bench $small > /dev/null 2> "$out/pmc_sq.err"
echo "[profile] SQ counters done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 $bench $small > /dev/null 2> "$out/pmc_fetch.err"
echo "[profile] FETCH_SIZE done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 $bench $small > /dev/null 2> "$out/pmc_write.err"
echo "[profile] WRITE_SIZE done"
cd "$root"
python3 tools/summarize_profile.py "$out" > "$out/summary.json"
tail -c 1500 "$out/summary.json"
