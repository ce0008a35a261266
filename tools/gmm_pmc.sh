#!/bin/bash
# Fabric-side traffic of the scoring kernels for a set of env variants (GPU box): FETCH_SIZE / WRITE_SIZE / L2 hit-miss.
# usage: bash tools/gmm_pmc.sh OUTDIR "name|ENV=1 ..." ...
set -eo pipefail
root=$(pwd)
out=$root/$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bench="$root/bench.py --no-cpu-baseline --steps 1 --warmup 2 --inflight 1 --no-extra-loops"
for v in "$@"; do
  IFS='|' read -r name envs <<< "$v"
  for e in $envs; do export "$e"; done
  for ctr in FETCH_SIZE WRITE_SIZE; do      # one counter per pass (more than that does not fit the TCC's counters)
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/$name/$ctr" -- python3 $bench > /dev/null 2> "$out/$name.$ctr.err" \
      || { echo "$name $ctr failed"; grep -v "^    @" "$out/$name.$ctr.err" | tail -5; exit 1; }
  done
  for e in $envs; do unset "${e%%=*}"; done
  python3 - "$out/$name" "$name" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
root, name = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    if "gmm" in k:
        print(name, k, "dispatches", len(cnt[k]), {c: f"{x:.4g}" for c, x in v.items()})
PY
  find "$out/$name" -name "*.csv" -size +1M -delete
done
