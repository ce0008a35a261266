"""Extended fuzz of the lazy (windowed, speculative) path beyond the test suite's seeds: random graphs (cycles, dead ends,
wide states), random mixture models with every slot class, random beams, window sizes and look-aheads; every output and
every written score cell must equal the dense path's bit for bit (the dense path is fuzzed against the oracle by
tools/decoder_fuzz.py).  GPU; from the repo root:  python tools/lazy_fuzz.py [n_seeds] [first_seed] [--eps]
--eps: a fifth of every graph's arcs become epsilon input arcs (the closure inside the windowed, lazily scored decoder)."""
import os
import sys

sys.path.insert(0, ".")
import numpy as np                                                     # noqa: E402
import torch                                                           # noqa: E402

from montreal_forced_aligner_amd.engine import AlignmentEngine         # noqa: E402
from tests import helpers                                              # noqa: E402
from tests.test_gpu_lazy import _both, _dev                            # noqa: E402
from tests.test_gpu_parity import _random_graph                        # noqa: E402

fx = helpers.Fixtures()
eng = AlignmentEngine(0)
tm = fx.mono_tm
_nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
n_seeds = _nums[0] if _nums else 24
seed0 = _nums[1] if len(_nums) > 1 else 0
EPS = "--eps" in sys.argv
bad = 0
for seed in range(seed0, seed0 + n_seeds):
    rng = np.random.default_rng(9100 + seed)
    dim = int(rng.choice([39, 40, 45]))
    # single-block classes only (1..32 Gaussians): lazy and dense cells are bit-identical there (multi-block pdfs agree to 1e-4)
    sizes = [int(x) for x in rng.choice([1, 2, 3, 4, 5, 8, 9, 12, 16, 17, 26, 32], size=tm.num_pdfs)]
    eng.load_gmm(helpers.random_gmm(rng, dim, sizes))
    fsts, feats = [], []
    for u in range(10):
        f = _random_graph(rng, tm, int(rng.choice([3, 8, 40, 150, 400, 900])))
        if EPS:
            from montreal_forced_aligner_amd import kaldi_io as K
            arcs = f.arcs.copy()
            eps = rng.random(len(arcs)) < 0.2
            arcs["ilabel"][eps] = 0
            arcs["olabel"][eps & (rng.random(len(arcs)) < 0.7)] = 0      # (few word labels on epsilon arcs: keeps words <= frames mostly)
            arcs["weight"][eps & (rng.random(len(arcs)) < 0.3)] = 0.0    # zero-weight epsilon arcs (cycles of them included): ties
            src = np.repeat(np.arange(f.num_states), np.diff(f.arc_offsets))
            neg = eps & (arcs["nextstate"] > src) & (rng.random(len(arcs)) < 0.2)
            arcs["weight"][neg] -= 0.5
            g = K.Fst(f.start, f.arc_offsets, arcs, f.final)
            if helpers.has_negative_eps_cycle(g):                        # Kaldi's closure does not terminate on one
                arcs["weight"][neg] += 0.5
                g = K.Fst(f.start, f.arc_offsets, arcs, f.final)
            if not eng.needs_general_decoder(g):
                f = g
        fsts.append(f)
        feats.append(rng.normal(0, 3.0, size=(int(rng.integers(2, 400)), dim)).astype(np.float32))
    fo = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int64)
    graphs = eng.pack_graphs(fsts, tm, groups=int(rng.choice([1, 2, 8])))
    beam = float(rng.choice([2.0, 8.0, 30.0]))
    retry = float(rng.choice([0.0, 4.0])) * beam
    window = int(rng.choice([64, 128]))
    look = int(rng.choice([4, 16, 40, 48, 63, 200]))
    os.environ["MFA_LAZY_LOOKAHEAD"] = str(look)
    try:
        dense, lazy, fill = _both(eng, graphs, _dev(eng, np.concatenate(feats)), fo, window=window, beam=beam, retry_beam=retry,
                                  max_tokens=2048, bp_tokens_per_frame=1100, acoustic_scale=0.1)
        st = np.unique(dense["status"].cpu().numpy(), return_counts=True)
        print(seed, f"dim {dim} beam {beam} retry {retry} window {window} look-ahead {look} fill {fill:.2f}", st, flush=True)
    except AssertionError as e:
        bad += 1
        print(seed, "MISMATCH", str(e)[:200], flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
