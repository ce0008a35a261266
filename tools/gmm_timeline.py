"""Summarise a scoring-kernel timeline written by `MFA_GMM_TRACE=file.npy python bench.py ...` (mfa_debug_gmm_trace).

Records: [utterance, 64-frame sub-tile, {start, end, hw id, blocks}]."""
import sys

import numpy as np

t = np.load(sys.argv[1]).astype(np.int64)
n_utt, subs, _ = t.shape
ran = t[..., 1] > 0
start, end, hw, blocks = (t[..., k] for k in range(4))
t0 = start[ran].min()
tick_us = 0.01                                      # wall_clock64: 100 MHz
span = (end[ran].max() - t0) * tick_us
print(f"utterances {n_utt}, sub-tile records {int(ran.sum())}, span {span / 1e3:.2f} ms")
dur = (end - start) * tick_us
for k in range(subs):
    m = ran[:, k]
    if not m.any():
        continue
    b, d = blocks[:, k][m], dur[:, k][m]
    print(f"sub-tile {k:2d}: {int(m.sum())} records, blocks mean {b.mean():.0f}, duration mean {d.mean():.0f} us "
          f"(p10 {np.percentile(d, 10):.0f}, p90 {np.percentile(d, 90):.0f}); us per block {d.mean() / max(b.mean(), 1):.2f}, "
          f"started at {((start[:, k][m] - t0) * tick_us).mean() / 1e3:.1f} ms on average")
xcc = (hw >> 32) & 0xF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
wv = hw & 0xF
home = xcc == (np.arange(n_utt) & 7)[:, None]
print(f"records run on their home XCD: {home[ran].mean():.4f}")
slot = ((xcc * 8 + se) * 16 + cu) * 64 + simd * 16 + wv
u, inv = np.unique(slot[ran], return_inverse=True)
busy = np.bincount(inv, weights=dur[ran])
print(f"wavefront slots used: {len(u)}; busy time per slot: min {busy.min() / 1e3:.2f} ms, mean {busy.mean() / 1e3:.2f} ms, "
      f"max {busy.max() / 1e3:.2f} ms  → occupancy {busy.mean() / span:.3f}")
last_end = np.zeros(len(u))
np.maximum.at(last_end, inv, (end[ran] - t0) * tick_us)
print(f"slots finish between {last_end.min() / 1e3:.2f} and {last_end.max() / 1e3:.2f} ms (mean {last_end.mean() / 1e3:.2f})")
tot_blocks = blocks[ran].sum()
print(f"32-row blocks walked {tot_blocks}: {tot_blocks / span / len(u) * 1e0:.4f} blocks/us/slot; "
      f"matrix-pipe share if a block is 80 MFMAs of 64 cycles at 2.4 GHz: {tot_blocks * 5120 / 2400 / (span * len(u) / 2):.3f}")
