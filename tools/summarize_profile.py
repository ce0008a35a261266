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
    if "gmm_band_f32_kernel" in name:
        return "gmm_band_f32_kernel"
    if "gmm_band_kernel" in name:            # <steps, pieces>: 2 = f16×2 pass, 3 = bf16×3 pass (its redo sweep when f16 is on)
        pieces = name.split("gmm_band_kernel<")[-1].split(">")[0].replace(" ", "").split(",")[-1] if "<" in name else "?"
        return {"2": "gmm_band_kernel_f16", "3": "gmm_band_kernel_bf16"}.get(pieces, "gmm_band_kernel")
    if "gmm_split_single_kernel" in name:   # <steps, pieces>: 2 = f16×2 pass, 3 = bf16×3 pass (or its redo sweep)
        pieces = name.split("gmm_split_single_kernel<")[-1].split(">")[0].replace(" ", "").split(",")[-1] if "<" in name else "?"
        return {"2": "gmm_split_single_kernel_f16", "3": "gmm_split_single_kernel_bf16"}.get(pieces, "gmm_split_single_kernel")
    for k in ("gmm_split_single_kernel", "gmm_bf16_single_kernel", "gmm_bf16_kernel", "gmm_presplit_kernel", "gmm_kernel",
              "viterbi_small_kernel", "viterbi_finish_kernel", "viterbi_kernel", "mfcc_kernel", "feats_lda_kernel", "feats_kernel", "cmvn_utt_kernel", "cmvn_spk_kernel",
              "arcnext_kernel", "collect_pending_kernel", "finalize_pending_kernel", "gmm_max_first_frame_kernel",
              "gmm_band_ranges_kernel", "gmm_col_rows_kernel"):
        if k in name:
            return k
    base = name.replace("(anonymous namespace)::", "")
    return (base.split("(")[0] or base)[-60:]


summary = {"source": os.path.basename(root), "kernels": {}}
# ---- kernel trace of the full bench run: duration per dispatch
dur = defaultdict(list)
for r in rows("stats", "kernel_trace.csv"):
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
total = sum(sum(v) for v in dur.values())
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6   # (round-1 field: the longest launches of a kernel)
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
# ---- fabric-side bytes per bench step and stage, from the PMC passes (ONE timed step after 3 warm-up steps and the fill
# probe, one batch in flight): per kernel, dispatches of the whole run ÷ steps run = dispatches per step.  FETCH_SIZE is
# doubled: on gfx950 it tallies the 128-byte requests of wide coalesced reads at 64 bytes (MI355X_MICROARCH.md, HBM section);
# WRITE_SIZE is exact for 16-byte-per-lane and dword-per-lane stores.  Infinity-Cache hits are counted (fabric side).
stage_of = {"gmm_band_kernel_f16": "gmm", "gmm_band_kernel_bf16": "gmm", "gmm_band_f32_kernel": "gmm",
            "gmm_split_single_kernel_f16": "gmm", "gmm_split_single_kernel_bf16": "gmm", "gmm_kernel": "gmm", "gmm_bf16_kernel": "gmm",
            "gmm_presplit_kernel": "gmm", "viterbi_kernel": "viterbi", "viterbi_small_kernel": "viterbi", "viterbi_finish_kernel": "viterbi",
            "mfcc_kernel": "mfcc", "feats_lda_kernel": "feats", "feats_kernel": "feats"}
pmc_steps = 5.0   # 3 warm-up + 1 fill probe + 1 timed step in each PMC run (the model-training set-up launches are tiny)
traffic = {}
for k, e in summary["kernels"].items():
    st = stage_of.get(k)
    d = e.get("derived", {})
    if st is None or "fetch_bytes_per_dispatch_raw" not in d:
        continue
    n_f = e.get("pmc_dispatches", {}).get("pmc_fetch", 0) / pmc_steps
    n_w = e.get("pmc_dispatches", {}).get("pmc_write", 0) / pmc_steps
    t = traffic.setdefault(st, {"bytes_per_step": 0.0, "fetch_bytes_per_step_corrected": 0.0, "write_bytes_per_step": 0.0, "kernels": []})
    fb = 2.0 * d["fetch_bytes_per_dispatch_raw"] * n_f
    wb = d.get("write_bytes_per_dispatch", 0.0) * n_w
    t["fetch_bytes_per_step_corrected"] += fb
    t["write_bytes_per_step"] += wb
    t["bytes_per_step"] += fb + wb
    t["kernels"].append(k)
for t in traffic.values():
    t["how"] = ("sum over the stage's kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 per dispatch x dispatches per step; "
                "separate rocprofv3 --pmc passes, one batch in flight; FETCH_SIZE doubled per the gfx950 correction")
summary["bench_roofline_traffic"] = traffic
print(json.dumps(summary, indent=1))
