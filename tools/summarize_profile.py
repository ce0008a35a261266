"""Merge the rocprofv3 outputs of tools/profile_round.sh into one JSON: per kernel, calls / average duration (kernel
trace), PMC sums per dispatch, effective clock, matrix-pipe busy fraction and fabric-side bytes per launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def rows(sub, suffix):
    out = []
    for f in glob.glob(os.path.join(root, sub, "**", f"*{suffix}"), recursive=True):
        with open(f, newline="") as fh:
            out.extend(csv.DictReader(fh))
    return out


def short(name):
    if "gmm_split_single_kernel" in name:   # <steps, pieces>: 2 = f16×2 pass, 3 = bf16×3 pass (or its redo sweep)
        pieces = name.split("gmm_split_single_kernel<")[-1].split(">")[0].replace(" ", "").split(",")[-1] if "<" in name else "?"
        return {"2": "gmm_split_single_kernel_f16", "3": "gmm_split_single_kernel_bf16"}.get(pieces, "gmm_split_single_kernel")
    for k in ("gmm_split_single_kernel", "gmm_bf16_single_kernel", "gmm_bf16_kernel", "gmm_kernel", "viterbi_kernel", "mfcc_kernel", "feats_lda_kernel", "feats_kernel", "cmvn_utt_kernel", "cmvn_spk_kernel",
              "arcnext_kernel", "collect_pending_kernel", "finalize_pending_kernel", "gmm_max_first_frame_kernel"):
        if k in name:
            return k
    return name.split("(")[0][-60:]


summary = {"source": os.path.basename(root), "kernels": {}}
# ---- kernel trace of the full bench run: duration per dispatch
dur = defaultdict(list)
for r in rows("stats", "kernel_trace.csv"):
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
total = sum(sum(v) for v in dur.values())
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6   # bench launches of a stage in the stats run (warm-up + timed)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    top = sorted(v, reverse=True)[:steps]          # the bench launches; the rest are the synthetic model's set-up calls
    summary["kernels"][k] = {"calls": len(v), "avg_ms": sum(v) / len(v), "total_ms": sum(v), "share": sum(v) / total,
                             "bench_launch_avg_ms": sum(top) / len(top)}
# ---- PMC passes (one bench step + one warm-up step each): per-dispatch sums
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    for r in rows(sub, "counter_collection.csv"):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
    tr = defaultdict(list)
    for r in rows(sub, "kernel_trace.csv"):
        tr[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k in acc:
        e = summary["kernels"].setdefault(k, {})
        pm = e.setdefault("pmc_per_dispatch", {})
        for c, v in acc[k].items():
            # rocprofv3 emits one row per dispatch (and per dimension for some counters): average over dispatches
            pm[c] = v / max(1, len(tr[k]))
        e.setdefault("pmc_dispatch_ms", {})[sub] = sum(tr[k]) / max(1, len(tr[k]))
        e.setdefault("pmc_dispatches", {})[sub] = len(tr[k])
for k, e in summary["kernels"].items():
    pm = e.get("pmc_per_dispatch", {})
    ms = e.get("pmc_dispatch_ms", {})
    d = {}
    if "GRBM_GUI_ACTIVE" in pm and ms.get("pmc_sq"):
        d["effective_clock_GHz"] = pm["GRBM_GUI_ACTIVE"] / 8 / (ms["pmc_sq"] * 1e-3) / 1e9
        if "SQ_VALU_MFMA_BUSY_CYCLES" in pm:
            # busy cycles are summed over the 1024 SIMDs' matrix pipes... reported per SE×CU: normalise by active cycles
            d["mfma_busy_over_gui_active"] = pm["SQ_VALU_MFMA_BUSY_CYCLES"] / (pm["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
    if "FETCH_SIZE" in pm:
        d["fetch_bytes_per_dispatch_raw"] = pm["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in pm:
        d["write_bytes_per_dispatch"] = pm["WRITE_SIZE"] * 1024
    if d:
        e["derived"] = d
print(json.dumps(summary, indent=1))
