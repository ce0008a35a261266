"""Extended decoder fuzz beyond the test suite's seeds (GPU; run from the repo root: python tools/decoder_fuzz.py [--eps]).
Random graphs, random or tie-heavy scores, random beams; every utterance must match the oracle bit for bit.
--eps: a fifth of every graph's arcs become epsilon input arcs (chains, zero-weight cycles, a quarter of the forward ones with
negative weights) — ProcessNonemitting inside the wavefront-parallel decoder, tie-heavy scores included."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from tests import helpers
from tests.test_gpu_parity import _random_graph, _align_case
from montreal_forced_aligner_amd.engine import AlignmentEngine
fx = helpers.Fixtures()
eng = AlignmentEngine(0)
tm = fx.mono_tm
from montreal_forced_aligner_amd import kaldi_io as K
EPS = "--eps" in sys.argv
bad = 0
for seed in range(3, 23):
    rng = np.random.default_rng(5000 + seed)
    fsts, lls = [], []
    for u in range(16):
        S = int(rng.choice([2, 5, 17, 64, 129, 300, 700]))
        f = _random_graph(rng, tm, S)
        if EPS:
            arcs = f.arcs.copy()
            eps = rng.random(len(arcs)) < 0.2
            arcs["ilabel"][eps] = 0
            src = np.repeat(np.arange(f.num_states), np.diff(f.arc_offsets))
            neg = eps & (arcs["nextstate"] > src) & (rng.random(len(arcs)) < 0.25)
            arcs["weight"][neg] -= 0.5
            f = K.Fst(f.start, f.arc_offsets, arcs, f.final)
            if eng.needs_general_decoder(f):
                f = _random_graph(rng, tm, 5)
        fsts.append(f)
        T = int(rng.integers(1, 140))
        if rng.random() < 0.4:
            ll = (rng.integers(-240, -160, size=(T, tm.num_pdfs)) * 0.25).astype(np.float32)
        else:
            ll = rng.normal(-60.0, float(rng.choice([1.0, 5.0, 25.0, 60.0])), size=(T, tm.num_pdfs)).astype(np.float32)
        lls.append(ll)
    beam = float(rng.choice([0.25, 1.0, 4.0, 10.0, 30.0]))
    retry = float(rng.choice([0.0, 4.0])) * beam
    try:
        res = _align_case(eng, tm, fx.mono_am, fsts, lls, beam, retry, max_tokens=1024, bp_tokens=700)
        print(seed, beam, retry, np.unique(res["status"], return_counts=True), flush=True)
    except AssertionError as e:
        bad += 1
        print("MISMATCH", seed, beam, retry, str(e)[:300], flush=True)
print("mismatches:", bad)
