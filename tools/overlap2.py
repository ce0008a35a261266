#!/usr/bin/env python3
"""Which kernels share the chip with several batches in flight (rocprofv3 --kernel-trace CSV of bench.py --no-extra-loops):
over the last `span_ms` of the trace, wall time by the set of WIDE kernel classes running (kernels that fill the chip when alone:
band scoring, first-tier decoder, MFCC, LDA, pre-split) — "thin" = only few-wavefront launches (large tier, retry, finish,
ranges, sweeps) or nothing at all.   usage: overlap2.py <dir> [span_ms]"""
import csv, glob, sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 150e6
rows = list(csv.DictReader(open(path)))


def cls(n, dur):
    if "gmm_band_kernel<5, 2>" in n or "gmm_band_kernel<6, 2>" in n:
        return "band" if dur > 60_000 else None          # (the flagged-list launches are thin)
    if "viterbi_small_kernel" in n: return "tier1"
    if "mfcc_kernel" in n: return "mfcc"
    if "feats_lda" in n or "feats_kernel" in n: return "lda"
    if "gmm_presplit" in n: return "presplit"
    return None


ev, thin = [], []
end = max(int(r["End_Timestamp"]) for r in rows)
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if b < end - span: continue
    c = cls(r["Kernel_Name"], b - a)
    a = max(a, end - span)
    if c is None:
        thin.append((a, 1)); thin.append((b, -1))
    else:
        ev.append((a, 1, c)); ev.append((b, -1, c))
allev = sorted([(t, d, c) for t, d, c in ev] + [(t, d, "~thin") for t, d in thin])
live = defaultdict(int); last = allev[0][0]; acc = defaultdict(float); conc = 0.0
gaps, gap_start = [], None        # maximal intervals without a wide kernel
for t, d, c in allev:
    wide = sorted(k for k, v in live.items() if v > 0 and k != "~thin")
    key = "+".join(wide) if wide else ("thin only" if live["~thin"] > 0 else "idle")
    acc[key] += t - last
    if not wide and gap_start is None and t > last: gap_start = last
    if wide and gap_start is not None: gaps.append((gap_start, last)); gap_start = None
    conc += (t - last) * sum(v for k, v in live.items() if k != "~thin")
    last = t
    live[c] += d
tot = sum(acc.values())
print(f"span {tot / 1e6:.1f} ms; average number of wide kernels running {conc / tot:.2f}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:16]:
    print(f"{k:34s} {v / 1e6:9.2f} ms  {100 * v / tot:5.1f} %")

# the intervals without a wide kernel: how long, and which thin kernels run in the long ones
bins = [(0, 20e3), (20e3, 100e3), (100e3, 500e3), (500e3, 1e12)]
print("intervals without a wide kernel:")
for lo, hi in bins:
    sel = [(a, b) for a, b in gaps if lo <= b - a < hi]
    print(f"  {lo / 1e3:6.0f} .. {hi / 1e3 if hi < 1e11 else float('inf'):6.0f} us: {len(sel):5d} intervals, {sum(b - a for a, b in sel) / 1e6:7.2f} ms")
long_ = [(a, b) for a, b in gaps if b - a >= 100e3]
names = defaultdict(float)
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cls(r["Kernel_Name"], b - a) is not None: continue
    for ga, gb in long_:
        o = min(b, gb) - max(a, ga)
        if o > 0: names[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:]] += o
for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:8]:
    print(f"    {k:42s} {v / 1e6:8.2f} ms of thin-kernel time inside the >= 100 us intervals")
