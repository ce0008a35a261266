#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name, the durations of its launches inside the LAST bench step
(the launches after the last mfcc_kernel), in launch order.  usage: trace_kernels.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "mfcc_kernel" in r["Kernel_Name"])
step = rows[last:]
t0 = int(step[0]["Start_Timestamp"])
out = {}
for r in step:
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    out.setdefault(name, []).append(((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
for name, v in out.items():
    tot = sum(d for _, d in v)
    print(f"{name:34s} n={len(v):3d} total {tot:8.3f} ms : " + " ".join(f"{d:.2f}" for _, d in v[:40]))
print(f"step span {(int(step[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
