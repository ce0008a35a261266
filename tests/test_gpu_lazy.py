"""Lazy (windowed) scoring — mfa_align_features_batch — against the dense path and the oracle.

Kaldi's decodable is lazy: a score exists only if a live token's arc asked for it (GmmAligner.align_utterance,
MFA/alignment/multiprocessing.py:846-853).  The device path decodes in windows of K frames and, before each window,
scores only the pdfs that arcs within K arcs of the live tokens can emit.  That must change NOTHING:
  * every output (transition-ids, words, likelihood, per-frame likelihoods, status) equals the dense path's bit for bit,
  * every cell it writes carries exactly the dense kernel's bits,
  * and it writes far fewer cells.
The dense path itself is checked against the oracle in test_gpu_parity.py / test_gpu_headline_config.py; the last test
here closes the loop directly (lazy path vs oracle from PCM)."""
import numpy as np
import pytest
import torch

from montreal_forced_aligner_amd import graph as G
from oracle import oracle as O
from tests import helpers, synth
from tests.test_gpu_parity import _random_graph

pytestmark = pytest.mark.gpu


def _dev(e, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(e.device)


def _both(engine, graphs, feats, frame_off, window=64, **kw):
    ll, ll_off, ll_cols = engine.score(feats, frame_off, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts)
    dense = engine.align(graphs, ll, ll_off, ll_cols, frame_off, want_frame_likes=True, **kw)
    lazy = engine.align_features(graphs, feats, frame_off, want_frame_likes=True, window=window, **kw)
    torch.cuda.synchronize()
    for k in ("status", "ali", "words", "n_words", "like", "frame_like"):
        assert torch.equal(dense[k], lazy[k]), f"{k} differs between lazy and dense scoring"
    d, s = ll.cpu().numpy(), lazy["loglikes"].cpu().numpy()
    written = s != 0.0
    assert np.array_equal(d[written], s[written]), "a lazily scored cell differs from the dense kernel's value"
    return dense, lazy, float(written.mean())


@pytest.fixture(scope="module")
def tri(engine):
    world = synth.SynthWorld.build()
    engine.configure_mfcc()
    lda = synth.seeded_lda()
    fm = synth.seeded_fmllr(16)
    d_lda = torch.from_numpy(lda).to(engine.device)

    def feats_of(pcm_list, spks):
        so = np.concatenate([[0], np.cumsum([len(p) for p in pcm_list])]).astype(np.int64)
        mfcc, fo = engine.mfcc(torch.from_numpy(np.concatenate(pcm_list)).to(engine.device), so)
        own = np.arange(len(pcm_list), dtype=np.int32)
        stats = engine.cmvn_stats(mfcc, fo, own, len(pcm_list))
        per_utt = torch.from_numpy(fm[np.asarray(spks) % 16]).to(engine.device)
        return engine.features(mfcc, fo, own, stats, lda=d_lda, fmllr=per_utt), fo

    model = synth.train_triphone(world, lambda pcm, spk: feats_of([pcm], [spk])[0].cpu().numpy(), n_train=40, n_gauss=32,
                                 n_classes=2)
    return world, model, lda, fm, feats_of


@pytest.mark.parametrize("window", [64, 128])
def test_lazy_equals_dense_headline_shape(engine, tri, window):
    """configs[2] shape (32-Gaussian pdfs → f16×2 band kernel), ragged lengths incl. < 1 window and a non-multiple."""
    world, model, lda, fm, feats_of = tri
    engine.load_gmm(model.am)
    shapes = [(30, 160000), (30, 160000), (3, 16000), (8, 43200), (1, 8000), (30, 160001), (45, 240000)]
    utts = [world.utterance(7100 + i, n_words=nw, samples=ns) for i, (nw, ns) in enumerate(shapes)]
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    feats, fo = feats_of([u[0] for u in utts], [u[3] for u in utts])
    graphs = engine.pack_graphs(fsts, model.tm)
    dense, lazy, fill = _both(engine, graphs, feats, fo, window=window, beam=10.0, retry_beam=40.0, max_tokens=1024,
                              bp_tokens_per_frame=256)
    assert set(dense["status"].cpu().tolist()) <= {0, 1}
    assert fill < 0.45, f"lazy scoring wrote {fill:.2f} of the matrix"


def test_lazy_small_tables_growth_and_retry_passes(engine, tri):
    """The rare passes run the same windowed loop: token-capacity growth (tiny first-tier tables) and the retry beam (a beam
    so narrow that the first pass fails)."""
    world, model, lda, fm, feats_of = tri
    engine.load_gmm(model.am)
    utts = [world.utterance(7200 + i) for i in range(4)]
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    feats, fo = feats_of([u[0] for u in utts], [u[3] for u in utts])
    graphs = engine.pack_graphs(fsts, model.tm)
    # (min_active = 20 keeps a narrow beam from failing on these graphs; the retry pass proper is exercised on the
    #  reference's recording below, where beam 10 does fail)
    _both(engine, graphs, feats, fo, beam=0.05, retry_beam=40.0, max_tokens=1024, bp_tokens_per_frame=256)
    dense, _, _ = _both(engine, graphs, feats, fo, beam=60.0, retry_beam=0.0, max_tokens=2048, bp_tokens_per_frame=1024)
    assert set(dense["status"].cpu().tolist()) <= {0, 4}   # beam 60 keeps > 128 tokens alive: the growth pass ran


def test_lazy_single_gaussian_model_and_real_audio(engine, fx):
    """mono_model (single-Gaussian pdfs → bit-exact f32 band kernel) on the reference's recording, beam 100/400 as the
    reference's own tests use, and beam 10/40 (first pass fails on this plumbing model: retry / failure statuses)."""
    tm, am = fx.mono_tm, fx.mono_am
    sr = 16000
    cuts = [(0.0, 4.2), (4.0, 6.5), (0.0, 26.72), (23.5, 26.72)]
    texts = ["this is the acoustic corpus i'm talking pretty fast here", "there's nothing going else going on", fx.text,
             "um and that should be all thanks"]
    segs = [fx.pcm[int(a * sr): int(b * sr)] for a, b in cuts]
    engine.configure_mfcc()
    engine.load_gmm(am)
    so = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
    mfcc, fo = engine.mfcc(_dev(engine, np.concatenate(segs)), so)
    u2s = np.arange(len(segs), dtype=np.int32)
    feats = engine.features(mfcc, fo, u2s, engine.cmvn_stats(mfcc, fo, u2s, len(segs)))
    fsts = [fx.mono_graph(t) for t in texts]
    graphs = engine.pack_graphs(fsts, tm)
    seen = set()
    for beam, retry in ((100.0, 400.0), (10.0, 40.0)):
        dense, _, _ = _both(engine, graphs, feats, fo, beam=beam, retry_beam=retry, max_tokens=2048, bp_tokens_per_frame=1024)
        seen |= set(dense["status"].cpu().tolist())
    assert 1 in seen                                        # the retry-beam pass really ran (windowed, on its own list)


@pytest.mark.parametrize("seed,dim", [(0, 39), (1, 39), (0, 48), (1, 45)])
def test_lazy_random_graphs_all_slot_classes(engine, fx, seed, dim, monkeypatch):
    """Graphs with cycles, skips, dead ends and unreachable states over a model with every slot class (1, 4, 8, 16, 32 rows,
    multi-block) — both band kernels at once, for operand widths of 80 (dim 39) and 96 (dim 45, 48) columns.  (The band rule
    itself is brute-forced on such graphs on the CPU: tests/test_score_plan_cpu.py.)"""
    rng = np.random.default_rng(4000 + seed)
    tm = fx.mono_tm
    sizes = [int(x) for x in rng.choice([1, 2, 3, 4, 5, 8, 9, 12, 16, 17, 26, 32, 40, 70], size=tm.num_pdfs)]
    am = helpers.random_gmm(rng, dim, sizes)
    engine.load_gmm(am)
    fsts, feats = [], []
    for u in range(12):
        S = int(rng.choice([3, 8, 40, 150, 400, 1100]))
        fsts.append(_random_graph(rng, tm, S))
        T = int(rng.integers(2, 300))
        feats.append(rng.normal(0, 3.0, size=(T, dim)).astype(np.float32))
    fo = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int64)
    graphs = engine.pack_graphs(fsts, tm)
    beam, retry = [(8.0, 32.0), (30.0, 0.0)][seed]
    d_feats = _dev(engine, np.concatenate(feats))
    kw = dict(beam=beam, retry_beam=retry, max_tokens=2048, bp_tokens_per_frame=1100, acoustic_scale=0.1)
    # Same arithmetic on both sides → bit-identical: with MFA_GMM_BF16=0 every class is scored by the f32 kernels, dense and
    # lazy alike.
    monkeypatch.setenv("MFA_GMM_BF16", "0")
    _both(engine, graphs, d_feats, fo, **kw)
    monkeypatch.delenv("MFA_GMM_BF16")
    # Default arithmetic: the band kernel scores the 32-row single-block class and the 16/8/4-row classes with the dense
    # split-operand kernels' own expressions (bit-identical cells), single Gaussians go to the f32 band kernel (bit-identical
    # to the dense f32 kernel); only pdfs of more than 32 Gaussians may differ — both paths merge their blocks on the f16
    # pipe, but the dense kernel with its own block schedule: same scores to ≤ 1e-4·scale, not necessarily the same bits.
    ll, ll_off, ll_cols = engine.score(d_feats, fo, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts)
    dense = engine.align(graphs, ll, ll_off, ll_cols, fo, **kw)
    lazy = engine.align_features(graphs, d_feats, fo, **kw)
    d, s_ = ll.cpu().numpy(), lazy["loglikes"].cpu().numpy()
    cc = graphs.class_counts.cpu().numpy()
    P = np.diff(graphs.pdf_off_host)
    exact_cells = 0
    for u in range(len(fsts)):
        T = int(fo[u + 1] - fo[u])
        du = d[ll_off[u]: ll_off[u + 1]].reshape(T, P[u])
        su = s_[ll_off[u]: ll_off[u + 1]].reshape(T, P[u])
        multi = np.zeros(P[u], dtype=bool)
        multi[cc[u, 0]: cc[u, 0] + cc[u, 1]] = True
        w = su != 0.0
        assert np.array_equal(du[:, ~multi][w[:, ~multi]], su[:, ~multi][w[:, ~multi]]), f"utterance {u}"
        exact_cells += int(w[:, ~multi].sum())
        if w[:, multi].any():
            a_, b_ = du[:, multi][w[:, multi]], su[:, multi][w[:, multi]]
            assert np.abs(a_ - b_).max() <= 1e-4 * max(1.0, float(np.abs(a_).max()))
    assert exact_cells > 0
    same = 0
    for u in range(len(fsts)):
        a, b = int(fo[u]), int(fo[u + 1])
        if int(dense["status"][u]) == int(lazy["status"][u]) and torch.equal(dense["ali"][a:b], lazy["ali"][a:b]):
            same += 1
            if int(dense["status"][u]) in (0, 1):
                assert abs(float(dense["like"][u]) - float(lazy["like"][u])) / (b - a) < 1e-3
    assert same >= len(fsts) - 1
    # pdfs of several blocks: the block count packed into the column's row word (default) and looked up per column
    # (MFA_GMM_PACK_NB=0, the path of models with more than 1 024 Gaussians in a pdf) score the same cells to the same bits
    monkeypatch.setenv("MFA_GMM_PACK_NB", "0")
    lookup = engine.align_features(graphs, d_feats, fo, **kw)
    monkeypatch.delenv("MFA_GMM_PACK_NB")
    assert torch.equal(lookup["loglikes"], lazy["loglikes"]) and torch.equal(lookup["ali"], lazy["ali"])
    assert torch.equal(lookup["status"], lazy["status"])


def test_lazy_path_matches_oracle_from_pcm(engine, tri):
    """The default product path (front end → mfa_align_features_batch) against the oracle's whole path from the same PCM:
    boundaries frame-identical; log-likelihood within 1e-3 per frame (the value MFA stores, MFA/alignment/mixins.py:351-357)
    and the total difference reported."""
    world, model, lda, fm, feats_of = tri
    engine.load_gmm(model.am)
    utts = [world.utterance(7400 + i) for i in range(4)]
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    feats, fo = feats_of([u[0] for u in utts], [u[3] for u in utts])
    graphs = engine.pack_graphs(fsts, model.tm)
    res = engine.align_features(graphs, feats, fo, beam=10.0, retry_beam=40.0, max_tokens=256, bp_tokens_per_frame=128)
    res = {k: v.cpu().numpy() for k, v in res.items() if isinstance(v, torch.Tensor)}
    am = model.am
    for u, (pcm, text, segs, spk) in enumerate(utts):
        pl = graphs.pdf_lists_host[u]
        mf = O.mfcc(pcm.astype(np.float32), O.default_mfcc_opts())
        x = O.affine(O.affine(O.splice(O.cmvn_apply(O.cmvn_stats([mf]), mf)), lda), fm[spk % 16])
        ll = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl)
        ref = helpers.oracle_align(model.tm, fsts[u], ll, pl, beam=10.0, retry_beam=40.0)
        a, b = fo[u], fo[u + 1]
        assert res["status"][u] == ref["status"] == 0
        assert np.array_equal(res["ali"][a:b], ref["ali"])
        nw = int(res["n_words"][u])
        assert np.array_equal(res["words"][a: a + nw], ref["words"])
        total = abs(float(res["like"][u]) - ref["like"])
        print(f"utterance {u}: |delta log-likelihood| total {total:.4f}, per frame {total / (b - a):.2e}")
        assert total / (b - a) < 1e-3


def test_beam_10_40_result_is_the_unpruned_best_path(engine, tri, fx):
    """graph.py's training graphs are equivalent to Kaldi's but not determinised, and FasterDecoder's pruning
    depends on topology — so the claim worth checking is that pruning does not matter on the configurations the numbers are
    quoted on: beam 10 / retry 40 returns the same path as a decoder that prunes nothing (beam 1e4, tables at their hard
    upper bounds).  With no search error both graph forms give the same answer."""
    world, model, lda, fm, feats_of = tri
    engine.load_gmm(model.am)
    utts = [world.utterance(7500 + i) for i in range(12)]
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    feats, fo = feats_of([u[0] for u in utts], [u[3] for u in utts])
    graphs = engine.pack_graphs(fsts, model.tm)
    pruned = engine.align_features(graphs, feats, fo, beam=10.0, retry_beam=40.0, max_tokens=256, bp_tokens_per_frame=128)
    full = engine.align_features(graphs, feats, fo, beam=1.0e4, retry_beam=0.0, max_tokens=graphs.max_states,
                                 bp_tokens_per_frame=graphs.max_states)
    assert set(full["status"].cpu().tolist()) == {0}
    same = sum(int(torch.equal(pruned["ali"][fo[u]: fo[u + 1]], full["ali"][fo[u]: fo[u + 1]])) for u in range(len(utts)))
    assert same == len(utts), f"beam 10/40 left the unpruned best path on {len(utts) - same} of {len(utts)} utterances"
    # the reference's recording with its plumbing model, beams as the reference's own tests set them (100 / 400)
    tm, am = fx.mono_tm, fx.mono_am
    engine.load_gmm(am)
    sr = 16000
    segs = [fx.pcm[int(a * sr): int(b * sr)] for a, b in ((0.0, 4.2), (4.0, 6.5), (23.5, 26.72))]
    texts = ["this is the acoustic corpus i'm talking pretty fast here", "there's nothing going else going on",
             "um and that should be all thanks"]
    so = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
    mfcc, fo2 = engine.mfcc(_dev(engine, np.concatenate(segs)), so)
    u2s = np.arange(len(segs), dtype=np.int32)
    f2 = engine.features(mfcc, fo2, u2s, engine.cmvn_stats(mfcc, fo2, u2s, len(segs)))
    g2 = engine.pack_graphs([fx.mono_graph(t) for t in texts], tm)
    a_ = engine.align_features(g2, f2, fo2, beam=100.0, retry_beam=400.0, max_tokens=2048, bp_tokens_per_frame=1024)
    b_ = engine.align_features(g2, f2, fo2, beam=1.0e4, retry_beam=0.0, max_tokens=g2.max_states, bp_tokens_per_frame=g2.max_states)
    assert set(b_["status"].cpu().tolist()) == {0}
    for u in range(len(segs)):
        assert torch.equal(a_["ali"][fo2[u]: fo2[u + 1]], b_["ali"][fo2[u]: fo2[u + 1]]), u


def test_plan_grouping_and_decoder_tier_do_not_change_results(engine, tri, monkeypatch):
    """Two layout / launch choices that must be invisible in the outputs: the score plan's grouping of the single-block
    pdfs (one run per XCD vs one run) and the first decoder tier (dedicated 64-token kernel vs the general kernel,
    MFA_VIT_LEAN=0).  Alignments, words, likelihoods and statuses are compared bit for bit; so are the scores of every
    column both layouts wrote (columns are matched through the arcs' column maps)."""
    world, model, lda, fm, feats_of = tri
    engine.load_gmm(model.am)
    utts = [world.utterance(7300 + i, n_words=nw, samples=ns) for i, (nw, ns) in enumerate([(30, 160000), (12, 70000), (40, 200000)])]
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    feats, fo = feats_of([u[0] for u in utts], [u[3] for u in utts])
    kw = dict(beam=10.0, retry_beam=40.0, max_tokens=1024, bp_tokens_per_frame=256, want_frame_likes=True)
    g8 = engine.pack_graphs(fsts, model.tm, groups=8)
    g1 = engine.pack_graphs(fsts, model.tm, groups=1)
    assert g8.groups == 8 and g8.group_counts is not None and g1.group_counts is None
    assert int(g8.group_counts.sum()) == int(g8.class_counts[:, 0].sum())
    ref = engine.align_features(g8, feats, fo, **kw)
    one = engine.align_features(g1, feats, fo, **kw)
    monkeypatch.setenv("MFA_VIT_LEAN", "0")
    general = engine.align_features(g8, feats, fo, **kw)
    monkeypatch.delenv("MFA_VIT_LEAN")
    # ... and how a failed speculation is redone: by the first tier itself one window later (lag mode, default) or by a
    # large-tier launch right away (MFA_VIT_LAG=0) — with a short look-ahead so that windows really are redone
    monkeypatch.setenv("MFA_LAZY_LOOKAHEAD", "8")
    short = engine.align_features(g8, feats, fo, **kw)
    monkeypatch.setenv("MFA_VIT_LAG", "0")
    short_nolag = engine.align_features(g8, feats, fo, **kw)
    monkeypatch.delenv("MFA_LAZY_LOOKAHEAD")
    nolag = engine.align_features(g8, feats, fo, **kw)
    monkeypatch.delenv("MFA_VIT_LAG")
    torch.cuda.synchronize()
    assert set(ref["status"].cpu().tolist()) <= {0, 1}
    for other, what in ((one, "ungrouped plan"), (general, "general kernel as first tier"), (short, "8-arc look-ahead, lag mode"),
                        (short_nolag, "8-arc look-ahead, large-tier redo"), (nolag, "large-tier redo")):
        for k in ("status", "ali", "words", "n_words", "like", "frame_like"):
            assert torch.equal(ref[k], other[k]), f"{k} differs with the {what}"
    # same cells, same values: column c of the grouped layout is the ungrouped layout's column of the same arcs
    a8, a1 = g8.tensors["arc_col"].cpu().numpy(), g1.tensors["arc_col"].cpu().numpy()
    ab = g8.tensors["arc_base"].cpu().numpy()
    l8, l1 = ref["loglikes"].cpu().numpy(), one["loglikes"].cpu().numpy()
    off8 = off1 = 0
    for u in range(len(fsts)):
        T, P = int(fo[u + 1] - fo[u]), int(g8.pdf_off_host[u + 1] - g8.pdf_off_host[u])
        m8, m1 = l8[off8: off8 + T * P].reshape(T, P), l1[off1: off1 + T * P].reshape(T, P)
        perm = np.zeros(P, dtype=np.int64)
        perm[a8[ab[u]: ab[u + 1]]] = a1[ab[u]: ab[u + 1]]
        both = (m8 != 0.0) & (m1[:, perm] != 0.0)
        assert both.any() and np.array_equal(m8[both], m1[:, perm][both])
        off8 += T * P; off1 += T * P


def test_lazy_window_edges_and_empty_transcript(engine, tri):
    """Utterance lengths around the window size (1, 63, 64, 65, 128, 129 frames), an empty transcript (a graph of optional
    silence only) and a transcript far too long for its audio (no path reaches a final state: failure status, no crash) —
    lazy and dense agree on everything, whatever the status."""
    world, model, lda, fm, feats_of = tri
    engine.load_gmm(model.am)
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    frames = [1, 63, 64, 65, 128, 129, 40, 30]
    texts = []
    pcm = []
    for i, T in enumerate(frames):
        u = world.utterance(7400 + i, n_words=2 if T < 100 else 4, samples=160 * T)
        pcm.append(u[0]); texts.append(u[1])
    texts[6] = ""                                                   # optional silence only
    texts[7] = " ".join(world.words[:40])                           # 40 words in 0.3 s
    fsts = [G.add_transition_probs(gc.compile_fst(t), scaled) for t in texts]
    feats, fo = feats_of(pcm, [0] * len(pcm))
    assert [int(fo[i + 1] - fo[i]) for i in range(len(frames))] == frames
    graphs = engine.pack_graphs(fsts, model.tm)
    dense, lazy, _ = _both(engine, graphs, feats, fo, beam=10.0, retry_beam=40.0, max_tokens=1024, bp_tokens_per_frame=256)
    st = dense["status"].cpu().tolist()
    assert st[6] in (0, 1)                                           # silence-only graph aligns
    assert st[7] == 2                                                # too much text for the audio: failed, reported as such
    assert all(s in (0, 1, 2) for s in st)


@pytest.mark.parametrize("lookahead", [6, 24])
def test_speculative_lookahead_failures_fall_back_to_the_proven_band(engine, tri, lookahead, monkeypatch):
    """The first-beam windows are scored for a look-ahead of 32 arcs instead of the proven 63; a decoder that reads a score
    outside what was scored hands the WINDOW over: it is scored again with the proven band and redone from the state parked
    at its start by the large tier.  Forced here with look-aheads far too short (6: nearly every window fails; 24: a good
    part of them): every output still equals the dense path's (``_both`` compares them)."""
    world, model, lda, fm, feats_of = tri
    engine.load_gmm(model.am)
    utts = [world.utterance(7500 + i, n_words=nw, samples=ns) for i, (nw, ns) in enumerate([(30, 160000), (10, 60000), (2, 9000), (35, 200000)])]
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    feats, fo = feats_of([u[0] for u in utts], [u[3] for u in utts])
    graphs = engine.pack_graphs(fsts, model.tm)
    kw = dict(beam=10.0, retry_beam=40.0, max_tokens=1024, bp_tokens_per_frame=256)
    _, _, fill_default = _both(engine, graphs, feats, fo, **kw)
    monkeypatch.setenv("MFA_LAZY_LOOKAHEAD", str(lookahead))
    dense, lazy, fill_forced = _both(engine, graphs, feats, fo, **kw)
    monkeypatch.setenv("MFA_LAZY_LOOKAHEAD", "63")           # speculation off: the proven band
    _, _, fill_proven = _both(engine, graphs, feats, fo, **kw)
    assert set(dense["status"].cpu().tolist()) <= {0, 1}
    assert fill_default < fill_proven                         # the default look-ahead scores fewer cells than the proven band
    if lookahead == 6:      # nearly every window was redone with the proven band: about as many cells as without speculation
        assert fill_forced > fill_default and fill_forced > 0.8 * fill_proven
    else:                   # some windows held: fewer cells than the proven band, more than the look-ahead alone would write
        assert fill_forced < fill_proven
