#!/bin/bash
# Profile pass for the bench workload on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>   → gpurun_out/prof_<tag>/{stats,pmc_sq,pmc_fetch,pmc_write}/… + summary.json
# rocprofv3 rules of this pool: PMC passes carry --kernel-trace only; python3 directly after "--".
# stats pass  = the default bench command (what the driver runs: several batches in flight);
# PMC passes  = one batch in flight, one timed step: per-dispatch counters of kernels that have the chip to themselves.
set -eo pipefail
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bench="$root/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 $bench --steps 6 --warmup 3 > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
echo "[profile] kernel stats done"
# the same with one batch in flight: per-kernel durations with the chip to themselves (what the bench line's `roofline` uses)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_single" -- python3 $bench --steps 3 --warmup 2 --inflight 1 --no-extra-loops > "$out/bench_single_under_rocprof.json" 2> "$out/stats_single.err"
echo "[profile] single-batch kernel stats done"
small="--steps 1 --warmup 3 --inflight 1 --no-extra-loops"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
  --output-format csv -d "$out/pmc_sq" -- python3 $bench $small > /dev/null 2> "$out/pmc_sq.err"
echo "[profile] SQ counters done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 $bench $small > /dev/null 2> "$out/pmc_fetch.err"
echo "[profile] FETCH_SIZE done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 $bench $small > /dev/null 2> "$out/pmc_write.err"
echo "[profile] WRITE_SIZE done"
cd "$root"
python3 tools/summarize_profile.py "$out" > "$out/summary.json"
# keep what gets committed small: the stats CSV of the kernel-stats pass and the summary
cp "$out"/stats/*/*kernel_stats.csv "$out/kernel_stats.csv" 2>/dev/null || true
cp "$out"/stats_single/*/*kernel_stats.csv "$out/kernel_stats_single.csv" 2>/dev/null || true
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*agent_info.csv" -delete
tail -c 1200 "$out/summary.json"
