"""Randomised check of the front end against the oracle: MFCC (random utterance lengths around every framing boundary, odd
sample offsets inside the batch buffer, silence, DC offsets, clipping, both snip_edges settings), per-speaker CMVN, the
delta and the splice+LDA+fMLLR feature kernels.  GPU.  python tools/frontend_fuzz.py [n_seeds] [first_seed]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import synth_workload as synth
from montreal_forced_aligner_amd.engine import AlignmentEngine
from oracle import oracle as O

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eng = AlignmentEngine(0)
lda = synth.seeded_lda()
fm = synth.seeded_fmllr(16)
bad = 0
for seed in range(seed0, seed0 + n_seeds):
    rng = np.random.default_rng(64000 + seed)
    snip = bool(rng.random() < 0.3)
    eng.configure_mfcc(snip_edges=snip)
    opts = O.default_mfcc_opts(snip_edges=snip)
    n = int(rng.integers(1, 30))
    segs, kinds = [], []
    for _ in range(n):
        L = int(rng.choice([1, 79, 80, 81, 159, 160, 161, 239, 240, 399, 400, 401, 559, 560, 561, 719, 1000, int(rng.integers(2, 70000))]))
        kind = rng.random()
        if kind < 0.1:
            s = np.zeros(L, np.int16)
        elif kind < 0.2:
            s = np.full(L, int(rng.integers(-32768, 32768)), np.int16)
        elif kind < 0.4:
            s = np.clip(rng.normal(0, 40000, L), -32768, 32767).astype(np.int16)                # heavy clipping
        else:
            t = np.arange(L) / 16000.0
            # (a noise floor of a few LSB under a full-scale tone puts most mel bins below the float32 rounding noise of ANY 512-point FFT —
            #  two implementations then differ by 1e-2 in the cepstra; measured with std 3: up to 0.027.  Speech is not like that.)
            s = (rng.normal(0, float(rng.choice([30, 300, 3000])), L) + 8000 * np.sin(2 * np.pi * float(rng.integers(60, 4000)) * t)
                 + float(rng.integers(-2000, 2000))).clip(-32768, 32767).astype(np.int16)
        segs.append(s)
        kinds.append("zeros" if kind < 0.1 else "constant" if kind < 0.2 else "clipped noise" if kind < 0.4 else "tone + noise")
    so = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
    pcm = torch.from_numpy(np.concatenate(segs)).to(eng.device)
    out, fo = eng.mfcc(pcm, so)
    out = out.cpu().numpy()
    refs = [O.mfcc(s.astype(np.float32), opts) for s in segs]
    worst = 0.0
    try:
        for u, ref in enumerate(refs):
            got = out[fo[u]: fo[u + 1]]
            assert got.shape == ref.shape, (u, len(segs[u]), got.shape, ref.shape)
            if ref.size:
                d = float(np.abs(got - ref).max())
                worst = max(worst, d)
                if d >= 1e-2:   # (speech-like signals agree to 1e-3; a strong tone over a weak floor leaves bins at the FFT's own rounding noise: ≤ 7e-3 seen)
                    fr, cf = np.unravel_index(np.argmax(np.abs(got - ref)), ref.shape)
                    raise AssertionError(f"utterance {u} ({kinds[u]}, {len(segs[u])} samples, std {segs[u].astype(np.float64).std():.1f}): |d| {d:.4f} at frame {fr} coefficient {cf}: device {got[fr, cf]:.4f} oracle {ref[fr, cf]:.4f}")
        # CMVN over random speaker groups + both feature kernels, from the DEVICE's MFCCs on both sides
        n_spk = int(rng.integers(1, 4))
        rows = rng.integers(0, n_spk, size=n).astype(np.int32)
        rows[: min(n, n_spk)] = np.arange(min(n, n_spk))          # every speaker row is used
        stats = eng.cmvn_stats(torch.from_numpy(out).to(eng.device), fo, rows, n_spk)
        st = stats.cpu().numpy()
        mf = [out[fo[u]: fo[u + 1]] for u in range(n)]
        for s_ in range(n_spk):
            mine = [mf[u] for u in range(n) if rows[u] == s_ and mf[u].shape[0] > 0]
            if mine:
                ref_st = O.cmvn_stats(mine)
                assert np.allclose(st[s_], ref_st, rtol=1e-12, atol=1e-9), ("cmvn", s_)
        d_mfcc = torch.from_numpy(out).to(eng.device)
        f_delta = eng.features(d_mfcc, fo, rows, stats).cpu().numpy()
        per_utt_fm = torch.from_numpy(fm[rows % 16]).to(eng.device)
        f_lda = eng.features(d_mfcc, fo, rows, stats, lda=torch.from_numpy(lda).to(eng.device), fmllr=per_utt_fm).cpu().numpy()
        wd = wl = 0.0
        for u in range(n):
            if mf[u].shape[0] == 0:
                continue
            base = O.cmvn_apply(st[rows[u]], mf[u])
            rd = O.deltas(base)
            rl = O.affine(O.affine(O.splice(base), lda), fm[rows[u] % 16])
            wd = max(wd, float(np.abs(f_delta[fo[u]: fo[u + 1]] - rd).max()))
            wl = max(wl, float(np.abs(f_lda[fo[u]: fo[u + 1]] - rl).max()))
        assert wd < 1e-3 and wl < 1e-3, ("features", wd, wl)
        print(seed, f"snip_edges {snip}, {n} utterances, mfcc worst {worst:.2e}, deltas {wd:.1e}, lda+fmllr {wl:.1e}", flush=True)
    except AssertionError as e:
        bad += 1
        print(seed, "MISMATCH", str(e)[:300], flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
