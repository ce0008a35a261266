"""Graphs with epsilon input arcs (and states wider than 64 arcs) through mfa_align_general_batch — FasterDecoder with
ProcessNonemitting, one GPU thread per utterance — against the oracle, which implements the same rules
(oracle/mfa_oracle.cpp: ProcessNonemitting, HashList).  kalpy-compiled training graphs (FstArchive handed to
export_alignments, MFA/alignment/multiprocessing.py:846-853) can hold such arcs.  Bit-exact or it fails."""
import numpy as np
import pytest
import torch

from montreal_forced_aligner_amd import graph as G
from montreal_forced_aligner_amd import kaldi_io as K
from oracle import oracle as O
from tests import helpers
from tests.test_gpu_parity import _random_graph

pytestmark = pytest.mark.gpu


def _dev(e, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(e.device)


def _with_eps(rng, f, frac=0.3):
    """An equivalent graph with epsilon input arcs: a share of the arcs s -il:ol/w-> d is split into
    s -eps:ol/w-> m -il:eps/0-> d through a fresh state m (the word label and the weight travel on the epsilon arc)."""
    S = f.num_states
    arcs_by_state = [[] for _ in range(S)]
    for s in range(S):
        for a in range(int(f.arc_offsets[s]), int(f.arc_offsets[s + 1])):
            arcs_by_state[s].append(tuple(f.arcs[a]))
    extra = []
    final = list(f.final)
    for s in range(S):
        new = []
        for (il, ol, w, d) in arcs_by_state[s]:
            if il != 0 and d != s and rng.random() < frac:
                m = S + len(extra)
                extra.append([(il, 0, 0.0, d)])
                final.append(np.inf)
                new.append((0, ol, w, m))
            else:
                new.append((il, ol, w, d))
        arcs_by_state[s] = new
    allst = arcs_by_state + extra
    offs = np.concatenate([[0], np.cumsum([len(x) for x in allst])]).astype(np.int64)
    arr = np.zeros(int(offs[-1]), dtype=K.ARC_DTYPE)
    k = 0
    for lst in allst:
        for t in lst:
            arr[k] = t
            k += 1
    return K.Fst(f.start, offs, arr, np.asarray(final, dtype=np.float32))


def _oracle(tm, am, f, x, beam, retry):
    emit = f.arcs["ilabel"] > 0
    pl = np.unique(tm.id2pdf[f.arcs["ilabel"][emit]])
    ll = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl)
    return helpers.oracle_align(tm, f, ll, pl, beam=beam, retry_beam=retry)


def _check(engine, tm, am, fsts, feats, beam, retry):
    engine.load_gmm(am)
    fo = np.concatenate([[0], np.cumsum([x.shape[0] for x in feats])]).astype(np.int64)
    g = engine.pack_graphs_general(fsts, tm)
    res = engine.align_general(g, _dev(engine, np.concatenate(feats)), fo, beam=beam, retry_beam=retry, want_frame_likes=True)
    res = {k: v.cpu().numpy() for k, v in res.items() if v is not None}
    seen = set()
    for u, f in enumerate(fsts):
        ref = _oracle(tm, am, f, feats[u], beam, retry)
        want = helpers.device_status(ref, feats[u].shape[0])
        assert int(res["status"][u]) == want, (u, int(res["status"][u]), want)
        seen.add(want)
        if want in (0, 1):
            a, b = int(fo[u]), int(fo[u + 1])
            assert np.array_equal(res["ali"][a:b], ref["ali"]), u
            nw = int(res["n_words"][u])
            assert np.array_equal(res["words"][a: a + nw], ref["words"]), u
            # scores: the device's f32 kernel vs the oracle's chain are bit-identical for single Gaussians
            assert res["like"][u] == np.float32(ref["like"]), (u, res["like"][u], ref["like"])
            assert np.array_equal(res["frame_like"][a:b], ref["per_frame"]), u
    return seen


def test_training_graphs_with_epsilon_arcs_match_the_oracle(engine, fx, monkeypatch):
    """The reference's recording and plumbing model; training graphs rewritten with epsilon arcs on a third of their
    arcs: same alignment as the oracle on the same graph, and — the rewriting being an equivalence — the same transition-ids
    as the epsilon-free graph gives on the fast path when nothing is pruned."""
    tm, am = fx.mono_tm, fx.mono_am
    rng = np.random.default_rng(2)
    sr = 16000
    cuts = [(0.0, 4.2), (4.0, 6.5), (23.5, 26.72)]
    texts = ["this is the acoustic corpus i'm talking pretty fast here", "there's nothing going else going on",
             "um and that should be all thanks"]
    feats = [fx.mono_feats(fx.pcm[int(a * sr): int(b * sr)]) for a, b in cuts]
    plain = [fx.mono_graph(t) for t in texts]
    fsts = [_with_eps(rng, f) for f in plain]
    assert all((f.arcs["ilabel"] == 0).any() for f in fsts)
    for beam, retry in ((100.0, 400.0), (10.0, 40.0), (1.0e4, 0.0)):
        _check(engine, tm, am, fsts, feats, beam, retry)
    # a workspace cap that holds one utterance at a time: the batch is decoded in three launches, same results
    monkeypatch.setenv("MFA_GENERAL_WS_GIB", "0.001")
    _check(engine, tm, am, fsts, feats, 10.0, 40.0)
    monkeypatch.delenv("MFA_GENERAL_WS_GIB")
    # equivalence with the epsilon-free graphs on the fast path (no pruning: beam 1e4)
    engine.load_gmm(am)
    fo = np.concatenate([[0], np.cumsum([x.shape[0] for x in feats])]).astype(np.int64)
    d_feats = _dev(engine, np.concatenate(feats))
    gp = engine.pack_graphs(plain, tm)
    fast = engine.align_features(gp, d_feats, fo, beam=1.0e4, retry_beam=0.0, max_tokens=gp.max_states, bp_tokens_per_frame=gp.max_states)
    gen = engine.align_general(engine.pack_graphs_general(fsts, tm), d_feats, fo, beam=1.0e4, retry_beam=0.0)
    assert torch.equal(fast["ali"], gen["ali"]) and torch.equal(fast["n_words"], gen["n_words"])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_general_decoder_fuzz_with_epsilon_arcs(engine, fx, seed):
    """Random graphs (cycles, dead ends, out-degrees up to 90, exact cost ties) in which a fifth of the arcs are epsilon arcs —
    epsilon chains and zero-weight epsilon cycles included — against the oracle: hash-list order, LIFO closure, min_active,
    retry beam, failures."""
    tm, am = fx.mono_tm, fx.mono_am
    rng = np.random.default_rng(3000 + seed)
    fsts, feats = [], []
    for u in range(16):
        S = int(rng.choice([3, 8, 40, 150, 400, 1100]))
        f = _random_graph(rng, tm, S)
        arcs = f.arcs.copy()
        eps = rng.random(len(arcs)) < 0.2
        arcs["ilabel"][eps] = 0
        if u % 4 == 0 and S >= 8:          # a state wider than the fast decoder's 64 arcs
            wide = np.zeros(90, dtype=K.ARC_DTYPE)
            for k in range(90):
                wide[k] = (int(rng.integers(1, tm.num_transition_ids + 1)), 0, float(rng.integers(0, 8)) * 0.25, int(rng.integers(0, S)))
            a0 = int(f.arc_offsets[1])
            arcs = np.concatenate([arcs[:a0], wide, arcs[a0:]])
            offs = f.arc_offsets.copy()
            offs[2:] += 90
            f = K.Fst(f.start, offs, arcs, f.final)
        else:
            f = K.Fst(f.start, f.arc_offsets, arcs, f.final)
        fsts.append(f)
        feats.append(rng.normal(0, 3.0, size=(int(rng.integers(2, 120)), 39)).astype(np.float32))
    beam, retry = [(1.0, 4.0), (8.0, 32.0), (50.0, 0.0)][seed]
    seen = _check(engine, tm, am, fsts, feats, beam, retry)
    assert seen & {0, 1}


def test_kalpy_facade_routes_epsilon_graphs(engine, fx):
    """GmmAligner.align_utterance(fst, feats) takes an epsilon graph like any other (the general decoder behind it), also
    in a batch mixed with epsilon-free graphs."""
    from montreal_forced_aligner_amd import kalpy_api as KA

    tm, am = fx.mono_tm, fx.mono_am
    rng = np.random.default_rng(8)
    x = fx.mono_feats(fx.pcm[: 16000 * 4])
    plain = fx.mono_gc.compile_fst("this is the acoustic corpus")
    eps = _with_eps(rng, plain)
    al = KA.GmmAligner.__new__(KA.GmmAligner)
    al.acoustic_model_path = "mono"; al.transition_model, al.acoustic_model = tm, am
    al.beam, al.retry_beam = 100.0, 400.0
    al.transition_scale, al.acoustic_scale, al.self_loop_scale = 1.0, 0.1, 0.1
    al.disambiguation_symbols = []
    al._scaled = tm.scaled_log_probs(1.0, 0.1)
    al._loaded = False
    KA._ENGINE = engine
    out = al.align_utterances([plain, eps, plain], [x, x, x], ["a", "b", "c"])
    assert all(o is not None for o in out)
    ref = _oracle(tm, am, G.add_transition_probs(eps, al._scaled), x, 100.0, 400.0)
    assert out[1].alignment == ref["ali"].tolist() and out[1].words == ref["words"].tolist()
    assert out[0].alignment == out[2].alignment


def test_epsilon_closure_budget_hands_over_to_the_general_decoder(engine, fx, monkeypatch):
    """Kaldi's ProcessNonemitting has no budget; the wavefront kernel's closure has one (64 pops per token slot and frame:
    tools/decoder_fuzz.py --eps seed 325 — 700 states, a fifth of the arcs epsilon, a quarter of those with negative weights,
    beam 30 — exceeds it and comes back with a capacity status).  The product path must not report that as a failed
    alignment: after the hard-bounds redo the utterance goes to the general decoder.  Forced here with a budget of one pop."""
    from montreal_forced_aligner_amd import kalpy_api as KA

    tm, am = fx.mono_tm, fx.mono_am
    rng = np.random.default_rng(11)
    x = fx.mono_feats(fx.pcm[: 16000 * 4])
    plain = fx.mono_gc.compile_fst("this is the acoustic corpus")
    eps = _with_eps(rng, plain)
    al = KA.GmmAligner.__new__(KA.GmmAligner)
    al.acoustic_model_path = "mono"; al.transition_model, al.acoustic_model = tm, am
    al.beam, al.retry_beam = 100.0, 400.0
    al.transition_scale, al.acoustic_scale, al.self_loop_scale = 1.0, 0.1, 0.1
    al.disambiguation_symbols = []
    al._scaled = tm.scaled_log_probs(1.0, 0.1)
    al._loaded = False
    KA._ENGINE = engine
    scaled = G.add_transition_probs(eps, al._scaled)
    ref = _oracle(tm, am, scaled, x, 100.0, 400.0)
    assert ref["status"] in (0, 1)
    monkeypatch.setenv("MFA_VIT_EPS_POPS", "1")
    # the engine itself reports the capacity status, also with the hard bounds ...
    engine.load_gmm(am)
    g = engine.pack_graphs([scaled], tm)
    assert g.has_eps and g.hard_bounds() == (g.max_states, g.max_states + g.max_states // 2 + 2)
    fo = np.array([0, x.shape[0]], dtype=np.int64)
    mt, bp = g.hard_bounds()
    r = engine.align_features(g, _dev(engine, x), fo, beam=100.0, retry_beam=400.0, max_tokens=mt, bp_tokens_per_frame=bp)
    assert int(r["status"].cpu().numpy()[0]) == 3
    # ... and the kalpy-shaped aligner hands the utterance over: same alignment as with the default budget and as the oracle's
    out = al.align_utterances([plain, eps], [x, x], ["a", "b"])
    assert out[0] is not None and out[1] is not None
    assert out[1].alignment == ref["ali"].tolist() and out[1].words == ref["words"].tolist()
    monkeypatch.delenv("MFA_VIT_EPS_POPS")
    again = al.align_utterances([eps], [x], ["b"])
    assert again[0].alignment == out[1].alignment and again[0].likelihood == out[1].likelihood


def test_more_word_labels_than_frames_is_reported_not_truncated(engine, fx):
    """d_words holds one entry per frame.  A path over epsilon arcs that carry output labels can have more (found by
    tools/decoder_fuzz.py --eps: the words came back cut to the number of frames, silently): status 7 from both decoders, and
    the host layer reports it as this utterance's failure with the reason spelt out."""
    from montreal_forced_aligner_amd import _lib

    tm, am = fx.mono_tm, fx.mono_am
    x = fx.mono_feats(fx.pcm[: 16000 // 2])[:3]                # three frames
    tid = int(np.flatnonzero(tm.id2pdf >= 0)[0])
    # 0 -eps:11-> 1 -eps:12-> 2 -tid:13-> 3 -tid:14-> 4 -tid:0-> 5 (final), one self-loop-free chain: 4 word labels, 3 frames
    arcs = np.zeros(5, dtype=K.ARC_DTYPE)
    arcs["ilabel"] = [0, 0, tid, tid, tid]
    arcs["olabel"] = [11, 12, 13, 14, 0]
    arcs["weight"] = 0.25
    arcs["nextstate"] = [1, 2, 3, 4, 5]
    off = np.array([0, 1, 2, 3, 4, 5, 5], dtype=np.int64)
    final = np.array([np.inf] * 5 + [0.0], dtype=np.float32)
    f = K.Fst(0, off, arcs, final)
    ref = _oracle(tm, am, f, x, 100.0, 0.0)
    assert ref["status"] == 0 and ref["words"].tolist() == [11, 12, 13, 14] and len(ref["ali"]) == 3
    engine.load_gmm(am)
    fo = np.array([0, 3], dtype=np.int64)
    g = engine.pack_graphs([f], tm)
    r = engine.align_features(g, _dev(engine, x), fo, beam=100.0, retry_beam=0.0)
    assert int(r["status"].cpu()[0]) == 7 and int(r["n_words"].cpu()[0]) == 0
    gg = engine.pack_graphs_general([f], tm)
    r = engine.align_general(gg, _dev(engine, x), fo, beam=100.0, retry_beam=0.0)
    assert int(r["status"].cpu()[0]) == 7
    assert "more word labels" in _lib.status_reason(7)
    # one label fewer fits: identical to the oracle
    arcs2 = arcs.copy(); arcs2["olabel"][1] = 0
    f2 = K.Fst(0, off, arcs2, final)
    ref2 = _oracle(tm, am, f2, x, 100.0, 0.0)
    g2 = engine.pack_graphs([f2], tm)
    r2 = engine.align_features(g2, _dev(engine, x), fo, beam=100.0, retry_beam=0.0)
    assert int(r2["status"].cpu()[0]) == 0 and r2["words"].cpu().numpy()[:3].tolist() == ref2["words"].tolist() == [11, 13, 14]


def test_negative_epsilon_cycle_ends_as_a_failed_utterance(engine, fx):
    """An invalid graph — a cycle of epsilon arcs with negative total weight: Kaldi's ProcessNonemitting never terminates on it.
    Here every decoder is bounded (pop budget, token pool): the utterance comes back failed, the rest of the batch aligned."""
    from montreal_forced_aligner_amd import kalpy_api as KA

    tm, am = fx.mono_tm, fx.mono_am
    x = fx.mono_feats(fx.pcm[: 16000 * 2])
    good = G.add_transition_probs(fx.mono_gc.compile_fst("this is"), tm.scaled_log_probs(1.0, 0.1))
    tid = int(np.flatnonzero(tm.id2pdf >= 0)[0])
    arcs = np.array([(0, 0, -0.5, 1), (0, 0, 0.1, 0), (tid, 7, 0.0, 2), (tid, 0, 0.0, 2)], dtype=K.ARC_DTYPE)
    bad = K.Fst(0, np.array([0, 1, 3, 4], dtype=np.int64), arcs, np.array([np.inf, np.inf, 0.0], dtype=np.float32))
    assert helpers.has_negative_eps_cycle(bad)
    al = KA.GmmAligner.__new__(KA.GmmAligner)
    al.acoustic_model_path = "mono"; al.transition_model, al.acoustic_model = tm, am
    al.beam, al.retry_beam = 100.0, 400.0
    al.transition_scale, al.acoustic_scale, al.self_loop_scale = 1.0, 0.1, 0.1
    al.disambiguation_symbols = []
    al._scaled = np.zeros_like(tm.scaled_log_probs(1.0, 0.1))      # (graphs given with their weights in place)
    al._loaded = False
    KA._ENGINE = engine
    out = al.align_utterances([good, bad, good], [x, x, x], ["a", "b", "c"])
    assert out[1] is None and out[0] is not None and out[2] is not None
    assert out[0].alignment == out[2].alignment
    ref = _oracle(tm, am, good, x, 100.0, 400.0)
    assert out[0].alignment == ref["ali"].tolist()


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: epsilon input arcs on the wavefront-parallel decoder (pack_graphs stores every state's arcs [emitting | epsilon],
# the kEps instantiation of viterbi_kernel runs ProcessNonemitting after every frame) — the lazy-scored, windowed product path.
def _check_fast(engine, tm, am, fsts, feats, beam, retry, dense=False, **caps):
    keep = [u for u, f in enumerate(fsts) if not engine.needs_general_decoder(f)]
    assert keep
    fsts, feats = [fsts[u] for u in keep], [feats[u] for u in keep]
    engine.load_gmm(am)
    fo = np.concatenate([[0], np.cumsum([x.shape[0] for x in feats])]).astype(np.int64)
    g = engine.pack_graphs(fsts, tm)
    assert "state_nemit" in g.tensors
    d_feats = _dev(engine, np.concatenate(feats))
    kw = dict(beam=beam, retry_beam=retry, want_frame_likes=True, max_tokens=caps.get("max_tokens", g.max_states),
              bp_tokens_per_frame=caps.get("bp", max(g.max_states, 64)))
    if dense:
        ll, ll_off, ll_cols = engine.score(d_feats, fo, g.pdf_list, g.pdf_off_host, g.class_counts)
        res = engine.align(g, ll, ll_off, ll_cols, fo, **kw)
    else:
        res = engine.align_features(g, d_feats, fo, **kw)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in res.items() if isinstance(v, torch.Tensor) and k != "loglikes"}
    seen = set()
    for u, f in enumerate(fsts):
        ref = _oracle(tm, am, f, feats[u], beam, retry)
        want = helpers.device_status(ref, feats[u].shape[0])
        assert int(res["status"][u]) == want, (u, int(res["status"][u]), want)
        seen.add(want)
        if want in (0, 1):
            a, b = int(fo[u]), int(fo[u + 1])
            assert np.array_equal(res["ali"][a:b], ref["ali"]), u
            nw = int(res["n_words"][u])
            assert np.array_equal(res["words"][a: a + nw], ref["words"]), u
            assert res["like"][u] == np.float32(ref["like"]), (u, res["like"][u], ref["like"])
            assert np.array_equal(res["frame_like"][a:b], ref["per_frame"]), u
    return seen, len(keep)


def test_epsilon_training_graphs_on_the_wavefront_decoder(engine, fx):
    """The same rewritten training graphs as above, through the product path: bit-identical to the oracle with the default
    capacities (tiers, windows, retry beam) and with scores given densely."""
    tm, am = fx.mono_tm, fx.mono_am
    rng = np.random.default_rng(2)
    sr = 16000
    cuts = [(0.0, 4.2), (4.0, 6.5), (23.5, 26.72)]
    texts = ["this is the acoustic corpus i'm talking pretty fast here", "there's nothing going else going on",
             "um and that should be all thanks"]
    feats = [fx.mono_feats(fx.pcm[int(a * sr): int(b * sr)]) for a, b in cuts]
    fsts = [_with_eps(rng, fx.mono_graph(t)) for t in texts]
    for beam, retry in ((100.0, 400.0), (10.0, 40.0), (1.0e4, 0.0)):
        _seen, n = _check_fast(engine, tm, am, fsts, feats, beam, retry, max_tokens=1024, bp=512)
        assert n == 3
        _check_fast(engine, tm, am, fsts, feats, beam, retry, dense=True, max_tokens=1024, bp=512)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4])
def test_wavefront_decoder_fuzz_with_epsilon_arcs(engine, fx, seed):
    """Random graphs with a fifth of their arcs turned into epsilon arcs — chains, zero-weight cycles, exact cost ties,
    several epsilon arcs of one state into the same destination, more states than hash buckets — against the oracle:
    closure costs, LIFO insertion order, hash-list order of the states the closure creates, back-pointers on ties."""
    tm, am = fx.mono_tm, fx.mono_am
    rng = np.random.default_rng(5000 + seed)
    fsts, feats = [], []
    while len(fsts) < 24:
        S = int(rng.choice([3, 8, 40, 150, 400, 1100, 1500]))
        f = _random_graph(rng, tm, S)
        arcs = f.arcs.copy()
        eps = rng.random(len(arcs)) < 0.2
        arcs["ilabel"][eps] = 0
        if seed == 4:                                    # negative epsilon weights too (no negative cycles: forward arcs only)
            src = np.repeat(np.arange(f.num_states), np.diff(f.arc_offsets))
            fwd = eps & (arcs["nextstate"] > src)
            arcs["weight"][fwd] -= 0.5
        f = K.Fst(f.start, f.arc_offsets, arcs, f.final)
        if engine.needs_general_decoder(f) or not eps.any() or helpers.has_negative_eps_cycle(f):   # (Kaldi's closure would not end)
            continue
        fsts.append(f)
        feats.append(rng.normal(0, 3.0, size=(int(rng.integers(2, 140)), 39)).astype(np.float32))
    beam, retry = [(1.0, 4.0), (8.0, 32.0), (50.0, 0.0), (3.0, 12.0), (20.0, 80.0)][seed]
    seen, n = _check_fast(engine, tm, am, fsts, feats, beam, retry)
    assert n == 24 and seen & {0, 1}
    _check_fast(engine, tm, am, fsts, feats, beam, retry, dense=True)
