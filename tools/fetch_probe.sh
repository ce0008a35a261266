#!/bin/bash
# Quick A/B probe on the GPU box: short bench (value + single-batch stage times) and one FETCH_SIZE pass (fabric-side bytes
# per step and stage).   bash tools/fetch_probe.sh <tag>  → gpurun_out/probe_<tag>/{bench.json,summary.json}
set -eo pipefail
tag=${1:-x}
root=$(pwd)
out=$root/gpurun_out/probe_$tag
mkdir -p "$out"
python3 bench.py --no-cpu-baseline --no-extra-loops --steps 10 --warmup 3 > "$out/bench.json" 2> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 $root/bench.py --no-cpu-baseline --steps 1 --warmup 3 --inflight 1 --no-extra-loops > /dev/null 2> "$out/pmc_fetch.err"
cd "$root"
python3 tools/summarize_profile.py "$out" > "$out/summary.json"
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*agent_info.csv" -delete
python3 - "$out" <<'PY'
import json, sys
o = sys.argv[1]
b = json.loads(open(o + "/bench.json").read().strip().splitlines()[-1])
s = json.load(open(o + "/summary.json"))
print("value", b["value"], "ms/step", b["ms_per_step"], "single", b.get("single_batch_in_flight", {}).get("stage_ms_per_step"))
for st, t in s.get("bench_roofline_traffic", {}).items():
    print(st, "fetch GB/step %.2f" % (t["fetch_bytes_per_step_corrected"] / 1e9))
for k, e in s["kernels"].items():
    d = e.get("derived", {})
    if "fetch_bytes_per_dispatch_raw" in d:
        print("  %-28s %6d dispatches  %8.3f MB/dispatch (x2)  %.3f ms" % (k, e["pmc_dispatches"]["pmc_fetch"], 2 * d["fetch_bytes_per_dispatch_raw"] / 1e6, e["pmc_dispatch_ms"]["pmc_fetch"]))
PY
