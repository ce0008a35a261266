"""Randomised check of the fMLLR statistics (device accumulation, single- and two-model form) and the host solve against the
oracle: random mixture models of every slot class, random features, random transition-id "alignments" with unaligned frames,
random speaker maps, random silence weights.  GPU.  python tools/fmllr_fuzz.py [n_seeds] [first_seed]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from montreal_forced_aligner_amd import fmllr as F
from montreal_forced_aligner_amd.engine import AlignmentEngine, fmllr_statistics
from oracle import oracle as O
from tests import helpers

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fx = helpers.Fixtures()
eng = AlignmentEngine(0)
tm = fx.mono_tm
bad = 0
for seed in range(seed0, seed0 + n_seeds):
    rng = np.random.default_rng(88000 + seed)
    dim = int(rng.choice([39, 40]))
    sizes = [int(x) for x in rng.choice([1, 2, 3, 4, 5, 8, 9, 12, 16, 17, 26, 32, 40], size=tm.num_pdfs)]
    am = helpers.random_gmm(rng, dim, sizes)
    two = rng.random() < 0.4
    stats_am = helpers.random_gmm(rng, dim, sizes) if two else None     # same mixture sizes, other means and variances
    eng.load_gmm(am)
    n = int(rng.integers(1, 12))
    feats = [rng.normal(0, 2.0, size=(int(rng.integers(1, 400)), dim)).astype(np.float32) for _ in range(n)]
    alis = [rng.integers(1, tm.num_transition_ids + 1, size=f.shape[0]).astype(np.int32) for f in feats]
    for a in alis:
        if rng.random() < 0.3:
            a[rng.integers(0, len(a), size=max(1, len(a) // 10))] = 0
    if n > 1 and rng.random() < 0.3:
        alis[1][:] = 0                                              # a failed utterance: no weight at all
    utt2spk = rng.integers(0, int(rng.integers(1, 5)), size=n) * 3 + 2
    sil = [int(x) for x in rng.choice(np.arange(1, 6), size=2, replace=False)]
    sw = float(rng.choice([0.0, 0.0, 0.5, 1.0]))
    fo = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int64)
    try:
        ids, beta, K, G = fmllr_statistics(eng, torch.from_numpy(np.concatenate(feats)).to(eng.device), fo,
                                           torch.from_numpy(np.concatenate(alis)).to(eng.device), tm, utt2spk, sil, sw, stats_model=stats_am)
        assert ids.tolist() == sorted(set(utt2spk.tolist()))
        sm = stats_am if two else am
        worst = 0.0
        for k, spk in enumerate(ids):
            stats = None
            for u in range(n):
                if utt2spk[u] != spk:
                    continue
                a = alis[u]
                w = np.where(a == 0, 0.0, np.where(np.isin(tm.id2phone[a], sil), sw, 1.0)).astype(np.float32)
                if two:
                    stats = O.fmllr_acc(feats[u], np.maximum(tm.id2pdf[a], 0), w, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets,
                                        stats, stat_means_invvars=sm.means_invvars, stat_inv_vars=sm.inv_vars)
                else:
                    stats = O.fmllr_acc(feats[u], np.maximum(tm.id2pdf[a], 0), w, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, stats)
            rb, rK, rG = stats[0][0], stats[1], stats[2]
            assert abs(beta[k] - rb) < 1e-3 * max(1.0, rb), ("beta", beta[k], rb)
            assert np.allclose(K[k], rK, rtol=1e-4, atol=2e-2), ("K", float(np.abs(K[k] - rK).max()))
            assert np.allclose(G[k], rG, rtol=1e-4, atol=2e-2), ("G", float(np.abs(G[k] - rG).max()))
            if rb > 60.0:
                Wd, impr_d = F.compute_fmllr(beta[k], K[k], G[k], min_count=50.0)
                Wo, impr_o = O.fmllr_solve(rb, rK, rG, min_count=50.0)
                worst = max(worst, float(np.abs(Wd - Wo).max()))
                assert abs(impr_d - impr_o) < 1e-3 * max(1.0, abs(impr_o)) and worst < 2e-3, ("solve", impr_d, impr_o, worst)
        print(seed, f"dim {dim} two-model {two} utterances {n} speakers {len(ids)} silence weight {sw} transform diff {worst:.1e}", flush=True)
    except (AssertionError, TypeError) as e:
        bad += 1
        print(seed, "MISMATCH", repr(e)[:300], flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
