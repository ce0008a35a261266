#!/usr/bin/env python3
"""Concurrency of the stages in a rocprofv3 --kernel-trace CSV of the bench (several batches in flight): over the last
`span_ms` of the trace, the wall time during which each set of stages had a kernel running.
usage: overlap.py <dir with *_kernel_trace.csv> [span_ms]"""
import csv, glob, sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 150e6
rows = list(csv.DictReader(open(path)))
def cls(n):
    for k, v in (("viterbi_kernel", "vit"), ("gmm_", "gmm"), ("mfcc_kernel", "mfcc"), ("feats", "feats"), ("cmvn", "feats")):
        if k in n: return v
    return None
ev = []
end = max(int(r["End_Timestamp"]) for r in rows)
for r in rows:
    c = cls(r["Kernel_Name"])
    if c is None: continue
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if b < end - span: continue
    ev.append((max(a, end - span), 1, c)); ev.append((b, -1, c))
ev.sort()
live = defaultdict(int); last = ev[0][0]; acc = defaultdict(float)
for t, d, c in ev:
    key = "+".join(sorted(k for k, v in live.items() if v > 0)) or "idle"
    acc[key] += t - last; last = t
    live[c] += d
tot = sum(acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:24s} {v / 1e6:9.2f} ms  {100 * v / tot:5.1f} %")
