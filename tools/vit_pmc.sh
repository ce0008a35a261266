#!/bin/bash
# Instruction mix of the decoder kernel (GPU box, through gpurun): two PMC passes over one bench step, sums per kernel.
# usage: bash tools/vit_pmc.sh OUTDIR [bench args]
set -eo pipefail
root=$(pwd)
out=$root/$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bench="$root/bench.py --no-cpu-baseline --steps 1 --warmup 2 --inflight 1 --no-extra-loops $*"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$out/a" -- python3 $bench > /dev/null 2> "$out/a.err"
echo "[pmc] pass a done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU \
  --output-format csv -d "$out/b" -- python3 $bench > /dev/null 2> "$out/b.err"
echo "[pmc] pass b done"
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
for sub in ("a", "b"):
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
            k = m.group(1) if m else r["Kernel_Name"][:48]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
        print(sub, k, "dispatches", len(cnt[k]), {c: f"{x:.4g}" for c, x in v.items()})
PY
find "$out" -name "*.csv" -size +2M -delete
