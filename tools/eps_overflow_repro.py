"""Repro of decoder_fuzz --eps seed 325 (utterance 7 reported a token-capacity status the oracle does not have): the same
batch at growing capacities.  GPU; python tools/eps_overflow_repro.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
from tests import helpers
from tests.test_gpu_parity import _random_graph, _align_case
from montreal_forced_aligner_amd.engine import AlignmentEngine
from montreal_forced_aligner_amd import kaldi_io as K
fx = helpers.Fixtures()
eng = AlignmentEngine(0)
tm = fx.mono_tm
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 325
rng = np.random.default_rng(5000 + seed)
fsts, lls = [], []
for u in range(16):
    S = int(rng.choice([2, 5, 17, 64, 129, 300, 700]))
    f = _random_graph(rng, tm, S)
    arcs = f.arcs.copy()
    eps = rng.random(len(arcs)) < 0.2
    arcs["ilabel"][eps] = 0
    src = np.repeat(np.arange(f.num_states), np.diff(f.arc_offsets))
    neg = eps & (arcs["nextstate"] > src) & (rng.random(len(arcs)) < 0.25)
    arcs["weight"][neg] -= 0.5
    f = K.Fst(f.start, f.arc_offsets, arcs, f.final)
    if helpers.has_negative_eps_cycle(f):              # Kaldi's closure does not terminate on one: take the negative weights back
        arcs["weight"][neg] += 0.5
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
print("beam", beam, retry, "states", [f.num_states for f in fsts], "arcs", [f.num_arcs for f in fsts],
      "eps arcs", [int((f.arcs["ilabel"] == 0).sum()) for f in fsts], "T", [l.shape[0] for l in lls])
for mt, bp in ((1024, 700), (1024, 1400), (2048, 1400), (4096, 2800)):
    try:
        res = _align_case(eng, tm, fx.mono_am, fsts, lls, beam, retry, max_tokens=mt, bp_tokens=bp)
        print("max_tokens", mt, "bp", bp, "OK", np.unique(res["status"], return_counts=True))
    except AssertionError as e:
        print("max_tokens", mt, "bp", bp, "MISMATCH", str(e)[:200])
