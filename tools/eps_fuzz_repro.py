"""Repro of one seed of tools/decoder_fuzz.py --eps with every output compared field by field.  GPU; python tools/eps_fuzz_repro.py SEED"""
import sys
sys.path.insert(0, '.')
import numpy as np
from tests import helpers
from tests.test_gpu_parity import _random_graph, _dev
from montreal_forced_aligner_amd.engine import AlignmentEngine
from montreal_forced_aligner_amd import kaldi_io as K
fx = helpers.Fixtures()
eng = AlignmentEngine(0)
tm = fx.mono_tm
seed = int(sys.argv[1])
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
print("seed", seed, "beam", beam, retry)
eng.load_gmm(fx.mono_am)
graphs = eng.pack_graphs(fsts, tm)
frame_off = np.concatenate([[0], np.cumsum([l.shape[0] for l in lls])]).astype(np.int64)
cols = [lls[u][:, graphs.pdf_lists_host[u]] for u in range(len(fsts))]
ll_off = np.concatenate([[0], np.cumsum([c.size for c in cols])]).astype(np.int64)
d_ll = _dev(eng, np.concatenate([c.reshape(-1) for c in cols]).astype(np.float32))
ll_cols = _dev(eng, np.array([c.shape[1] for c in cols], dtype=np.int32))
res = eng.align(graphs, d_ll, ll_off, ll_cols, frame_off, beam=beam, retry_beam=retry, acoustic_scale=0.1, max_tokens=1024,
                bp_tokens_per_frame=700, want_frame_likes=True)
res = {k: (v.cpu().numpy() if v is not None else None) for k, v in res.items()}
for u, f in enumerate(fsts):
    ref = helpers.oracle_align(tm, f, cols[u], graphs.pdf_lists_host[u], acoustic_scale=0.1, beam=beam, retry_beam=retry)
    a, b = frame_off[u], frame_off[u + 1]
    st = int(res["status"][u])
    line = f"utt {u}: S={f.num_states} T={b - a} eps={int((f.arcs['ilabel'] == 0).sum())} status gpu {st} ref {ref['status']}"
    if st == ref["status"] and st in (0, 1):
        nw = int(res["n_words"][u])
        same_ali = np.array_equal(res["ali"][a:b], ref["ali"])
        same_w = np.array_equal(res["words"][a: a + nw], ref["words"])
        same_l = res["like"][u] == np.float32(ref["like"])
        same_f = np.array_equal(res["frame_like"][a:b], ref["per_frame"])
        line += f" ali {same_ali} words {same_w} like {same_l} frame_like {same_f}"
        if not same_w:
            line += f"\n    words gpu {res['words'][a: a + nw].tolist()} ref {ref['words'].tolist()}"
        if not same_f:
            bad = np.nonzero(res["frame_like"][a:b] != ref["per_frame"])[0]
            line += f"\n    frame_like differs at {bad[:8].tolist()} gpu {res['frame_like'][a:b][bad[:4]].tolist()} ref {ref['per_frame'][bad[:4]].tolist()}"
        if not same_l:
            line += f"\n    like gpu {res['like'][u]!r} ref {ref['like']!r}"
    print(line)
