"""Where a decoder frame spends its cycles.  Run on the GPU box, from the repo root:
    MFA_HIPCC_FLAGS=-DVIT_STAMPS python -c "from montreal_forced_aligner_amd import _lib; _lib.build_native(force=True)"
    MFA_VIT_STAMPS=gpurun_out/vit_stamps.npy python bench.py --no-cpu-baseline --steps 1
    python tools/viterbi_phases.py gpurun_out/vit_stamps.npy 1000
(then rebuild without the flag).  The stamps themselves cost ≈40 cycles each, nine per frame."""
import sys

import numpy as np

a = np.load(sys.argv[1]).astype(np.float64)
frames = float(sys.argv[2]) if len(sys.argv) > 2 else float(a[:, 11].mean())   # [11] = frames each utterance decoded
names = ["score row staged", "GetCutoff", "candidate layout (scan, owner map)", "arc gather + score + cost",
         "running cutoff (seed, prefix-min)", "claim / lower / winner", "general path + stash winners",
         "list order (bucket ranks, ordinal scan)", "new list, back-pointers, reset"]
per_frame = a[:, : len(names)].mean(0) / frames
tot = per_frame.sum()
for n, c in zip(names, per_frame):
    print(f"{n:42s} {c:8.0f} cycles/frame  {100 * c / tot:5.1f} %")
print(f"{'total':42s} {tot:8.0f} cycles/frame")
print(f"frames that ran the exact min_active selection: {a[:, 9].mean() / frames:.3f} of all; tokens per frame {a[:, 10].mean() / frames:.1f}")
