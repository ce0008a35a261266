"""Timeline of ONE bench step from a rocprofv3 kernel trace (one batch in flight): every launch of the decoder/scoring kernels
in start order with its offset from the step's first kernel, duration and the idle gap before it.
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --inflight 1 --steps 2 --warmup 2 --no-extra-loops --no-cpu-baseline
  python tools/step_timeline.py DIR [step_index_from_end]"""
import csv
import glob
import os
import sys

root = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        rows.extend(csv.DictReader(fh))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


# steps start with mfcc_kernel launches of the full batch (the long ones)
starts = [i for i, r in enumerate(rows) if "mfcc_kernel" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1_000_000]
i0 = starts[-back]
i1 = starts[-back + 1] if back > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
tot = {}
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = short(r["Kernel_Name"])
    tot[name] = tot.get(name, 0) + (e - s)
    if e - s > 30_000 or "viterbi" in name:
        print(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f} us  {name}")
    prev_end = max(prev_end, e)
print("step wall %.3f ms" % ((prev_end - t0) / 1e6))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-46s %8.3f ms" % (k, v / 1e6))
