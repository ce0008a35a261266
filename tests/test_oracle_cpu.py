"""CPU tests: readers against the facts the reference's fixtures hold, the C++ oracle against the
independent numpy restatement, host tables against the oracle's, and the oracle decoder against exhaustive Viterbi.
PARITY STATUS: the reference ships no golden vectors at the kalpy boundary — parity unpinned."""
import io

import numpy as np
import pytest

from montreal_forced_aligner_amd import graph as G
from montreal_forced_aligner_amd import kaldi_io as K
from oracle import np_oracle as N
from oracle import oracle as O
from tests import helpers


def test_model_dimensions_match_reference_fixtures(fx):
    # facts read from tests/data/am/*.zip of the reference
    assert fx.mono_am.dim == 39 and fx.mono_am.num_pdfs == 132 and fx.mono_am.num_gauss == 132
    assert fx.mono_tm.tuples.shape[0] == 543 and fx.mono_tm.log_probs.shape[0] == 1207
    assert len(fx.mono_tm.topo.phones) == 171 and fx.mono_tm.num_transition_ids == 1206
    assert fx.g2p_am.dim == 40 and fx.g2p_am.num_pdfs == 80 and fx.g2p_am.num_gauss == 500
    assert fx.g2p_tm.tuples.shape[0] == 244 and fx.g2p_tm.log_probs.shape[0] == 505
    assert fx.g2p_lda.shape == (40, 91)
    assert fx.mono_tree.context_width == 1 and fx.mono_tree.central_position == 0
    assert fx.g2p_tree.context_width == 3 and fx.g2p_tree.central_position == 1
    assert len(set(fx.mono_tree.to_pdf.leaves())) == 132 and len(set(fx.g2p_tree.to_pdf.leaves())) == 80
    g = [o1 - o0 for o0, o1 in zip(fx.g2p_am.pdf_offsets[:-1], fx.g2p_am.pdf_offsets[1:])]
    assert min(g) == 1 and max(g) == 26


def test_topology_matches_expected_topo(fx):
    # tests/data/dictionaries/expected/topo of the reference: 3-state Bakis (0.75/0.25) and the 5-state silence topology
    bakis, sil = fx.mono_tm.topo.entries[0], fx.mono_tm.topo.entries[1]
    assert [s.transitions for s in bakis[:3]] == [[(0, 0.75), (1, 0.25)], [(1, 0.75), (2, 0.25)], [(2, 0.75), (3, 0.25)]]
    assert sil[0].transitions == [(0, 0.25), (1, 0.25), (2, 0.25), (3, 0.25)]
    assert sil[4].transitions == [(4, 0.75), (5, 0.25)] and sil[5].transitions == []


def test_model_roundtrip(fx):
    buf = io.BytesIO()
    raw_tm = fx.g2p_tm.raw
    raw_am = K.read_model(fx.g2p_archive["final.mdl"])[1]
    K.write_model(buf, raw_tm, raw_am)
    assert buf.getvalue() == fx.g2p_archive["final.mdl"]


def test_fst_roundtrip(fx):
    data = fx.g2p_archive["english_us_mfa.fst"]
    f = K.read_fst(K.BinaryReader(data))
    assert (f.num_states, f.num_arcs, f.start) == (8478, 18602, 1)
    buf = io.BytesIO()
    K.write_fst(buf, f)
    f2 = K.read_fst(K.BinaryReader(buf.getvalue()))
    assert np.array_equal(f.arcs, f2.arcs) and np.array_equal(f.final, f2.final) and f2.start == 1


@pytest.mark.parametrize("snip", [0, 1])
def test_mfcc_oracle_vs_numpy(fx, snip):
    w = fx.pcm[: 16000 * 4].astype(np.float32)
    a = O.mfcc(w, O.default_mfcc_opts(snip_edges=snip))
    b = N.mfcc(w, snip_edges=bool(snip))
    assert a.shape == b.shape == ((400, 13) if not snip else (398, 13))
    assert np.abs(a - b).max() < 1e-3  # float32 FFT vs float64 FFT on values up to ~100


def test_mfcc_short_and_edge_cases(fx):
    o = O.default_mfcc_opts(snip_edges=1)
    assert O.mfcc_num_frames(399, o) == 0 and O.mfcc_num_frames(400, o) == 1 and O.mfcc_num_frames(160000, o) == 998
    o = O.default_mfcc_opts(snip_edges=0)
    assert O.mfcc_num_frames(160000, o) == 1000 and O.mfcc_num_frames(79, o) == 0 and O.mfcc_num_frames(80, o) == 1
    w = fx.pcm[5000:5300].astype(np.float32)  # shorter than a window: reflection on both sides
    a, b = O.mfcc(w, o), N.mfcc(w, snip_edges=False)
    assert a.shape == (2, 13) and np.abs(a - b).max() < 1e-3
    z = O.mfcc(np.zeros(1600, np.float32), o)  # digital silence: mel energies floored at FLT_EPSILON
    assert np.allclose(z, N.mfcc(np.zeros(1600), snip_edges=False), atol=1e-4)


def test_feature_chain_oracle_vs_numpy(fx):
    mf = O.mfcc(fx.pcm[: 16000 * 2].astype(np.float32), O.default_mfcc_opts())
    st = O.cmvn_stats([mf[:100], mf[100:]])
    assert st[0, 13] == mf.shape[0]
    assert np.allclose(st[0, :13], mf.astype(np.float64).sum(axis=0), rtol=1e-12)
    c = O.cmvn_apply(st, mf)
    assert np.abs(c - N.cmvn(mf.astype(np.float64))).max() < 1e-4
    assert np.allclose(O.delta_scales()[1, 2:7], np.array([-2, -1, 0, 1, 2]) / 10.0, atol=1e-7)
    assert np.allclose(O.delta_scales()[2], np.array([4, 4, 1, -4, -10, -4, 1, 4, 4]) / 100.0, atol=1e-7)
    assert np.abs(O.deltas(c) - N.deltas(c)).max() < 1e-4
    sp = O.splice(c)
    assert sp.shape == (c.shape[0], 91) and np.array_equal(sp, N.splice(c).astype(np.float32))
    lda = O.affine(sp, fx.g2p_lda)
    assert np.abs(lda - N.affine(sp.astype(np.float64), fx.g2p_lda.astype(np.float64))).max() < 1e-3
    rng = np.random.default_rng(0)
    fm = np.concatenate([np.eye(40) + 0.05 * rng.normal(size=(40, 40)), 0.1 * rng.normal(size=(40, 1))], axis=1).astype(np.float32)
    assert np.abs(O.affine(lda, fm) - N.affine(lda.astype(np.float64), fm.astype(np.float64))).max() < 1e-3


def test_gmm_oracle_vs_numpy(fx):
    x = fx.mono_feats(fx.pcm[: 16000 * 1])
    am = fx.mono_am
    pdfs = np.arange(am.num_pdfs, dtype=np.int32)
    a = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pdfs)
    b = N.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pdfs)
    assert np.abs(a - b).max() < 2e-3 and np.abs((a - b) / b).max() < 1e-5
    rng = np.random.default_rng(1)
    am = helpers.random_gmm(rng, 40, [1, 3, 4, 7, 8, 12, 16, 26, 32, 33, 70])
    x = rng.normal(0, 3, size=(50, 40)).astype(np.float32)
    pdfs = np.arange(am.num_pdfs, dtype=np.int32)[::-1].copy()
    a = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pdfs)
    b = N.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pdfs)
    assert np.abs((a - b) / b).max() < 1e-5


@pytest.mark.parametrize("which", ["mono", "g2p"])
def test_transition_model_tables_host_vs_oracle(fx, which):
    tm = fx.mono_tm if which == "mono" else fx.g2p_tm
    state2id, id2state, id2pdf, isl, isf = O.tm_derive(*tm.flat_topology(), tm.tuples)
    assert np.array_equal(state2id[1:], tm.state2id[1:])
    assert np.array_equal(id2state, tm.id2state) and np.array_equal(id2pdf[1:], tm.id2pdf[1:])
    assert np.array_equal(isl, tm.is_self_loop) and np.array_equal(isf, tm.is_final)
    for ts_, sl_ in ((1.0, 0.1), (1.0, 1.0), (0.0, 0.0)):
        a = O.tm_scaled_logprobs(tm.state2id, tm.id2state, tm.is_self_loop, tm.log_probs, ts_, sl_)
        assert np.array_equal(a, tm.scaled_log_probs(ts_, sl_))


def test_add_transition_probs_host_vs_oracle(fx):
    f = fx.mono_gc.compile_fst("this is the acoustic corpus")
    sc = fx.mono_tm.scaled_log_probs(1.0, 0.1)
    assert np.array_equal(O.add_transition_probs(f.arcs, sc), G.add_transition_probs(f, sc).arcs)


def test_graph_structure(fx):
    f = fx.mono_graph("this is")
    tm = fx.mono_tm
    il = f.arcs["ilabel"]
    assert np.all(il > 0), "compiled graphs are epsilon-free"
    src = np.repeat(np.arange(f.num_states), np.diff(f.arc_offsets))
    # AddSelfLoopsReorder invariants: every incoming arc of a state has one transition-state; the self-loop (if that
    # transition-state has one) is the LAST arc of the state and carries no word label
    ts_in = {}
    for a, s in zip(f.arcs, src):
        if a["nextstate"] != s:
            ts_in.setdefault(int(a["nextstate"]), set()).add(int(tm.id2state[a["ilabel"]]))
    assert all(len(v) == 1 for v in ts_in.values())
    for s, v in ts_in.items():
        a0, a1 = f.arc_offsets[s], f.arc_offsets[s + 1]
        sl = int(tm.self_loop_of[next(iter(v))])
        loops = [a for a in f.arcs[a0:a1] if a["nextstate"] == s]
        if sl:
            assert len(loops) == 1 and loops[0]["ilabel"] == sl and f.arcs[a1 - 1]["nextstate"] == s
            assert loops[0]["olabel"] == 0
        else:
            assert not loops
    assert np.isfinite(f.final).sum() >= 1


def _random_loglikes(rng, T, P):
    return rng.normal(-60.0, 12.0, size=(T, P)).astype(np.float32)


def test_oracle_decoder_wide_beam_is_exact_viterbi(fx):
    rng = np.random.default_rng(3)
    tm = fx.mono_tm
    f = fx.mono_graph("this is the acoustic corpus")
    pdfs = np.arange(tm.num_pdfs, dtype=np.int32)
    ll = _random_loglikes(rng, 180, tm.num_pdfs)
    r = helpers.oracle_align(tm, f, ll, pdfs, beam=1.0e4, retry_beam=0.0)
    assert r["status"] == 0
    tid2col = np.maximum(tm.id2pdf, 0)
    best, ali = N.viterbi_exact(f.num_states, f.start, f.arc_offsets, f.arcs, f.final, (-0.1 * ll.astype(np.float32)).astype(np.float32), tid2col)
    assert np.array_equal(r["ali"], ali)
    assert abs(-0.1 * r["like"] - best) < 1e-3 * max(1.0, abs(best)) * 1e-2
    words = [fx.mono_lex.word_table.find(int(w)) for w in r["words"]]
    assert words == "this is the acoustic corpus".split()
    ph, ok = O.split_to_phones(r["ali"], tm.id2state, tm.is_self_loop, tm.is_final, tm.tuples)
    assert ok and ph[:, 1].sum() == 180 and np.all(ph[:, 1] > 0)


def test_oracle_decoder_beam_retry_and_failure(fx):
    rng = np.random.default_rng(4)
    tm = fx.mono_tm
    f = fx.mono_graph("this is the acoustic corpus")
    pdfs = np.arange(tm.num_pdfs, dtype=np.int32)
    # 70 frames for 17 phones (51 emitting states minimum): little slack, so narrow beams lose the only viable path
    ll = rng.normal(-60.0, 30.0, size=(70, tm.num_pdfs)).astype(np.float32)
    wide = helpers.oracle_align(tm, f, ll, pdfs, beam=1e4, retry_beam=0.0)
    statuses = set()
    for beam, retry in ((0.5, 2.0), (2.0, 8.0), (10.0, 40.0), (1.0, 1000.0)):
        r = helpers.oracle_align(tm, f, ll, pdfs, beam=beam, retry_beam=retry, want_stats=True)
        statuses.add(r["status"])
        if r["status"] in (0, 1):
            assert r["like"] <= wide["like"] + 1e-2  # a pruned search can never beat the exact optimum
    assert 1 in statuses and 2 in statuses  # both the retry and the failure path are exercised
    too_short = helpers.oracle_align(tm, f, ll[:10], pdfs)  # 10 frames cannot cover 14 phones
    assert too_short["status"] == 2


def test_real_audio_plumbing_with_reference_test_beams(fx):
    """mono_model is the reference's deliberately tiny plumbing model (occupancies of a few hundred frames); like the
    reference's own tests (tests/conftest.py:1035-1037) it needs beam 100 / retry 400 to align acoustic_corpus.wav."""
    tm, am = fx.mono_tm, fx.mono_am
    x = fx.mono_feats(fx.pcm)
    pdfs = np.arange(am.num_pdfs, dtype=np.int32)
    ll = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pdfs)
    f = fx.mono_graph(fx.text)
    assert helpers.oracle_align(tm, f, ll, pdfs, beam=10.0, retry_beam=40.0)["status"] == 2
    r = helpers.oracle_align(tm, f, ll, pdfs, beam=100.0, retry_beam=400.0)
    assert r["status"] == 0 and r["ali"].shape[0] == 2672
    words = [fx.mono_lex.word_table.find(int(w)) for w in r["words"]]
    expect = [w if fx.mono_lex.word_table.member(w) else "<unk>" for w in fx.text.split()]
    assert words == expect


def test_lazy_decodable_equals_dense_scores_then_decode(fx):
    """orc_align_feats evaluates a (frame, pdf) score when a live token's arc first asks for it (Kaldi's
    DecodableAmDiagGmmUnmapped cache); everything it returns must equal scoring the whole matrix first — on the mixture
    model of the reference's g2p fixture and on the monophone fixture, first beam, retry beam and failure alike — and it
    must touch far fewer cells than the matrix holds."""
    rng = np.random.default_rng(11)
    tm, am = fx.mono_tm, fx.mono_am
    seg = fx.pcm[: 16000 * 5]
    x = fx.mono_feats(seg)
    f = fx.mono_graph("this is the acoustic corpus i'm talking pretty fast here")
    pdfs = np.arange(am.num_pdfs, dtype=np.int32)
    ll = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pdfs)
    for beam, retry in ((100.0, 400.0), (10.0, 400.0), (2.0, 8.0)):
        dense = helpers.oracle_align(tm, f, ll, pdfs, beam=beam, retry_beam=retry)
        lazy = helpers.oracle_align_feats(tm, f, x, am, beam=beam, retry_beam=retry)
        assert lazy["status"] == dense["status"]
        if dense["status"] in (0, 1):
            assert np.array_equal(lazy["ali"], dense["ali"]) and np.array_equal(lazy["words"], dense["words"])
            assert lazy["like"] == dense["like"] and np.array_equal(lazy["per_frame"], dense["per_frame"])
        assert 0 < lazy["cells"] < (2 if dense["status"] else 1) * ll.size
    # mixtures (1..26 Gaussians per pdf) in a 40-dim space, random features
    tm2, am2 = fx.g2p_tm, fx.g2p_am
    x2 = rng.normal(0, 1.0, size=(120, am2.dim)).astype(np.float32)
    # a left-to-right chain over the model's first phones' transition-ids is enough to drive the decodable
    from montreal_forced_aligner_amd import kaldi_io as K
    fwd = [t for t in range(1, tm2.num_transition_ids + 1) if not tm2.is_self_loop[t]][:40]
    arcs, offs = [], [0]
    for s, t in enumerate(fwd):
        loops = [u for u in range(1, tm2.num_transition_ids + 1) if tm2.is_self_loop[u] and tm2.id2state[u] == tm2.id2state[t]]
        for u in loops[:1]:
            arcs.append((u, 0, 0.1, s))
        arcs.append((t, 0, 0.2, s + 1))
        offs.append(len(arcs))
    offs.append(len(arcs))          # the last state: final, no arcs
    fst = K.Fst(0, np.asarray(offs, np.int64), np.asarray(arcs, dtype=O.ARC_DTYPE),
                np.asarray([np.inf] * len(fwd) + [0.0], np.float32))
    pdfs2 = np.arange(am2.num_pdfs, dtype=np.int32)
    ll2 = O.gmm_loglikes(x2, am2.gconsts, am2.means_invvars, am2.inv_vars, am2.pdf_offsets, pdfs2)
    d2 = helpers.oracle_align(tm2, fst, ll2, pdfs2, beam=50.0, retry_beam=5000.0)
    l2 = helpers.oracle_align_feats(tm2, fst, x2, am2, beam=50.0, retry_beam=5000.0)
    assert d2["status"] == l2["status"] and d2["status"] in (0, 1)
    assert np.array_equal(d2["ali"], l2["ali"]) and d2["like"] == l2["like"]


def test_parallel_formulation_of_the_decoder_matches_sequential_oracle(fx):
    """tests/viterbi_emul.py restates the GPU kernel's data-parallel formulation (prefix-min cutoff, first-creator
    ordering, bucket order) in numpy; it must reproduce the sequential decoder exactly, including under tight beams,
    ties, and graphs with more states than hash buckets."""
    from tests import viterbi_emul as E

    tm = fx.mono_tm
    rng = np.random.default_rng(21)
    words = [w for w in fx.text.split() if fx.mono_lex.word_table.member(w)]
    pdfs = np.arange(tm.num_pdfs, dtype=np.int32)
    tid2col = np.maximum(tm.id2pdf, 0).astype(np.int32)
    cases = []
    for u in range(10):
        text = " ".join(rng.choice(words, size=int(rng.integers(1, 7))))
        nph = sum(len(fx.mono_lex.word_pronunciations(w)[0].pronunciation.split()) for w in text.split())
        T = int(3 * nph * rng.uniform(1.0, 2.5)) + 2
        sd = float(rng.choice([0.0, 3.0, 12.0, 40.0]))
        cases.append((fx.mono_graph(text), rng.normal(-60.0, sd, size=(T, tm.num_pdfs)).astype(np.float32)))
    big = fx.mono_graph(" ".join(fx.text.split()[:44]))
    assert big.num_states > 1000
    cases.append((big, rng.normal(-60.0, 6.0, size=(600, tm.num_pdfs)).astype(np.float32)))
    n_ok = 0
    for f, ll in cases:
        for beam in (0.7, 3.0, 10.0, 200.0):
            ref = helpers.oracle_align(tm, f, ll, pdfs, beam=beam, retry_beam=0.0)
            got = E.decode(f.num_states, f.start, f.arc_offsets, f.arcs, f.final, ll, tid2col, 0.1, beam)
            assert got["status"] == ref["status"], (beam, got["status"], ref["status"])
            if ref["status"] == 0:
                n_ok += 1
                assert np.array_equal(got["ali"], ref["ali"]) and np.array_equal(got["words"], ref["words"])
                assert np.float32(got["like"]) == np.float32(ref["like"])
    assert n_ok >= 20
