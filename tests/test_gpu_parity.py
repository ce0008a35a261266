"""GPU parity tests: every kernel of libmfa_hip.so, called through the C ABI (ctypes → engine), against the CPU oracle on
the same inputs.  Integer/index results (alignments, words, status) must be identical; floating point within the stated
tolerance (north_star: log-likelihoods within 1e-3).  PARITY STATUS of the oracle itself: unpinned."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import helpers

pytestmark = pytest.mark.gpu


def _dev(e, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(e.device)


def _segments(fx):
    sr = 16000
    cuts = [(0.0, 3.0), (3.0, 3.21), (5.0, 12.5), (12.5, 12.62), (14.0, 26.7)]
    return [fx.pcm[int(a * sr): int(b * sr)] for a, b in cuts]


@pytest.mark.parametrize("snip", [0, 1])
def test_mfcc_matches_oracle(engine, fx, snip):
    segs = _segments(fx)
    engine.configure_mfcc(snip_edges=snip)
    sample_off = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
    pcm = _dev(engine, np.concatenate(segs).astype(np.int16))
    out, frame_off = engine.mfcc(pcm, sample_off)
    out = out.cpu().numpy()
    opts = O.default_mfcc_opts(snip_edges=snip)
    worst = 0.0
    for u, s in enumerate(segs):
        ref = O.mfcc(s.astype(np.float32), opts)
        got = out[frame_off[u]: frame_off[u + 1]]
        assert got.shape == ref.shape
        if ref.size:
            worst = max(worst, float(np.abs(got - ref).max()))
    # two float32 FFTs of different radix on int16-scale audio: values reach ~100, agreement ~1e-4
    assert worst < 2e-3, worst
    engine.configure_mfcc()


@pytest.mark.parametrize("raw,floor", [(1, 0.0), (0, 0.0), (1, 5.0e7)])
def test_mfcc_use_energy_matches_oracle(engine, fx, raw, floor):
    """use_energy (MFA/corpus/features.py:780-820 passes it through; default false): C0 is the frame's log energy — before
    pre-emphasis and window with raw_energy, after them without, floored at log(energy_floor) — the other coefficients
    unchanged (Kaldi feat/feature-mfcc.cc, feature-window.cc ProcessWindow)."""
    segs = _segments(fx) + [np.zeros(2000, np.int16)]
    so = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
    pcm = _dev(engine, np.concatenate(segs).astype(np.int16))
    engine.configure_mfcc()
    plain, fo = engine.mfcc(pcm, so)
    plain = plain.cpu().numpy()
    engine.configure_mfcc(use_energy=1, raw_energy=raw, energy_floor=floor)
    out, fo2 = engine.mfcc(pcm, so)
    out = out.cpu().numpy()
    engine.configure_mfcc()
    assert np.array_equal(fo, fo2) and np.array_equal(out[:, 1:], plain[:, 1:])      # only C0 changes
    opts = O.default_mfcc_opts(use_energy=1, raw_energy=raw, energy_floor=floor)
    floored = 0
    for u, s in enumerate(segs):
        ref = O.mfcc(s.astype(np.float32), opts)
        got = out[fo[u]: fo[u + 1]]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 2e-3, (u, float(np.abs(got - ref).max()))
        if floor > 0:
            floored += int(np.isclose(ref[:, 0], np.log(floor), rtol=0, atol=1e-5).sum())
    assert floor == 0.0 or floored > 0                                                  # the floor really was applied somewhere
    assert np.abs(out[:, 0] - plain[:, 0]).max() > 1.0                                 # and C0 really is something else


def test_mfcc_digital_silence_and_full_scale(engine):
    engine.configure_mfcc()
    rng = np.random.default_rng(7)
    segs = [np.zeros(4000, np.int16), np.full(4000, 32767, np.int16), rng.integers(-32768, 32767, 8000).astype(np.int16)]
    sample_off = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
    out, frame_off = engine.mfcc(_dev(engine, np.concatenate(segs)), sample_off)
    out = out.cpu().numpy()
    for u, s in enumerate(segs):
        ref = O.mfcc(s.astype(np.float32), O.default_mfcc_opts())
        assert np.abs(out[frame_off[u]: frame_off[u + 1]] - ref).max() < 5e-3


def test_cmvn_and_delta_features_match_oracle(engine, fx):
    segs = _segments(fx)
    mf = [O.mfcc(s.astype(np.float32), O.default_mfcc_opts()) for s in segs]
    frame_off = np.concatenate([[0], np.cumsum([m.shape[0] for m in mf])]).astype(np.int64)
    utt2spk = np.array([0, 1, 0, 1, 2], dtype=np.int32)
    d_mf = _dev(engine, np.concatenate(mf))
    stats = engine.cmvn_stats(d_mf, frame_off, utt2spk, 3)
    feats = engine.features(d_mf, frame_off, utt2spk, stats).cpu().numpy()
    stats = stats.cpu().numpy()
    exact = total = 0
    for spk in range(3):
        ref = O.cmvn_stats([mf[u] for u in range(5) if utt2spk[u] == spk])
        assert np.allclose(stats[spk], ref, rtol=1e-13, atol=1e-9)
    for u in range(5):
        ref_stats = O.cmvn_stats([mf[v] for v in range(5) if utt2spk[v] == utt2spk[u]])
        ref = O.deltas(O.cmvn_apply(ref_stats, mf[u]))
        got = feats[frame_off[u]: frame_off[u + 1]]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 1e-4
        exact += int((got == ref).sum()); total += ref.size
    assert exact / total > 0.999  # same fmaf chain: bit-identical except where a CMVN offset rounds the other way


def test_lda_fmllr_features_match_oracle(engine, fx):
    segs = _segments(fx)[:3]
    mf = [O.mfcc(s.astype(np.float32), O.default_mfcc_opts(snip_edges=1)) for s in segs]
    frame_off = np.concatenate([[0], np.cumsum([m.shape[0] for m in mf])]).astype(np.int64)
    utt2spk = np.array([0, 1, 0], dtype=np.int32)
    rng = np.random.default_rng(0)
    fm = np.stack([np.concatenate([np.eye(40) + 0.05 * rng.normal(size=(40, 40)), 0.1 * rng.normal(size=(40, 1))], axis=1)
                   for _ in range(2)]).astype(np.float32)
    d_mf = _dev(engine, np.concatenate(mf))
    stats = engine.cmvn_stats(d_mf, frame_off, utt2spk, 2)
    lda = _dev(engine, fx.g2p_lda)
    no_fmllr = engine.features(d_mf, frame_off, utt2spk, stats, lda=lda).cpu().numpy()
    with_fmllr = engine.features(d_mf, frame_off, utt2spk, stats, lda=lda, fmllr=_dev(engine, fm)).cpu().numpy()
    for u in range(3):
        ref_stats = O.cmvn_stats([mf[v] for v in range(3) if utt2spk[v] == utt2spk[u]])
        y = O.affine(O.splice(O.cmvn_apply(ref_stats, mf[u])), fx.g2p_lda)
        assert np.abs(no_fmllr[frame_off[u]: frame_off[u + 1]] - y).max() < 1e-4
        z = O.affine(y, fm[utt2spk[u]])
        assert np.abs(with_fmllr[frame_off[u]: frame_off[u + 1]] - z).max() < 1e-4


def _score(engine, am, feats_list, pdf_lists):
    engine.load_gmm(am)
    frame_off = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats_list])]).astype(np.int64)
    sorted_lists, counts = zip(*[engine.sort_pdf_list(p) for p in pdf_lists])
    pdf_off = np.concatenate([[0], np.cumsum([len(p) for p in sorted_lists])]).astype(np.int64)
    ll, ll_off, _ = engine.score(_dev(engine, np.concatenate(feats_list).astype(np.float32)), frame_off,
                                 _dev(engine, np.concatenate(sorted_lists).astype(np.int32)), pdf_off,
                                 _dev(engine, np.stack(counts).astype(np.int32)))
    ll = ll.cpu().numpy()
    out = []
    for u, f in enumerate(feats_list):
        P = len(sorted_lists[u])
        out.append(ll[ll_off[u]: ll_off[u + 1]].reshape(f.shape[0], P))
    return out, sorted_lists


def test_gmm_single_gaussian_is_bit_exact(engine, fx):
    """mono_model has one Gaussian per pdf: LL is the MFMA's fmaf chain itself, which must equal the oracle's bit for bit."""
    am = fx.mono_am
    feats = [fx.mono_feats(s) for s in _segments(fx)[:3]]
    rng = np.random.default_rng(1)
    lists = [np.arange(am.num_pdfs, dtype=np.int32), rng.permutation(am.num_pdfs)[:37].astype(np.int32),
             np.array([5], dtype=np.int32)]
    got, sorted_lists = _score(engine, am, feats, lists)
    for u in range(3):
        ref = O.gmm_loglikes(feats[u], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, sorted_lists[u])
        assert np.array_equal(got[u], ref), float(np.abs(got[u] - ref).max())


def test_gmm_real_mixture_model(engine, fx, monkeypatch):
    monkeypatch.setenv("MFA_GMM_BF16", "0")   # this test pins the f32 MFMA kernel to the ulp
    am = fx.g2p_am  # 80 pdfs with 1..26 Gaussians: every slot class
    mf = [O.mfcc(s.astype(np.float32), O.default_mfcc_opts(snip_edges=1)) for s in _segments(fx)[:3]]
    feats = [O.affine(O.splice(O.cmvn_apply(O.cmvn_stats([m]), m)), fx.g2p_lda) for m in mf]
    rng = np.random.default_rng(2)
    lists = [np.arange(am.num_pdfs, dtype=np.int32), rng.permutation(am.num_pdfs)[:33].astype(np.int32),
             rng.permutation(am.num_pdfs)[:7].astype(np.int32)]
    got, sorted_lists = _score(engine, am, feats, lists)
    for u in range(3):
        ref = O.gmm_loglikes(feats[u], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, sorted_lists[u])
        err = np.abs(got[u] - ref)
        # the device exponential (exp2-based, ≈1.5 ulp) and log differ from libm in the last bits; everything before the
        # log-sum-exp is bit-identical, so the disagreement stays at the ulp level of a float32 log-likelihood (~1e-5)
        assert err.max() < 1e-4 * max(1.0, np.abs(ref).max()), err.max()
        assert err.max() <= 4 * np.spacing(np.float32(np.abs(ref).max())), err.max()
        assert (got[u] == ref).mean() > 0.5


@pytest.mark.parametrize("dim", [40, 39, 45])
def test_gmm_random_models_all_slot_classes(engine, dim):
    rng = np.random.default_rng(dim)
    sizes = [1, 1, 2, 3, 4, 4, 5, 7, 8, 8, 9, 12, 16, 16, 17, 26, 32, 32, 33, 64, 70, 1, 4, 8]
    am = helpers.random_gmm(rng, dim, sizes)
    feats = [rng.normal(0, 3, size=(t, dim)).astype(np.float32) for t in (1, 63, 64, 65, 300)]
    lists = [rng.permutation(am.num_pdfs)[:n].astype(np.int32) for n in (24, 24, 5, 1, 17)]
    got, sorted_lists = _score(engine, am, feats, lists)
    for u in range(len(feats)):
        ref = O.gmm_loglikes(feats[u], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, sorted_lists[u])
        assert np.abs(got[u] - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def _align_case(engine, tm, fx_am, fsts, lls_full, beam, retry, max_tokens=1024, scale=0.1, bp_tokens=512):
    """lls_full[u]: [T, num_pdfs] oracle scores for every pdf; the GPU gets the utterance's own columns."""
    engine.load_gmm(fx_am)  # pdf lists are ordered by the loaded model's slot classes
    graphs = engine.pack_graphs(fsts, tm)
    frame_off = np.concatenate([[0], np.cumsum([l.shape[0] for l in lls_full])]).astype(np.int64)
    cols = [lls_full[u][:, graphs.pdf_lists_host[u]] for u in range(len(fsts))]
    ll_off = np.concatenate([[0], np.cumsum([c.size for c in cols])]).astype(np.int64)
    d_ll = _dev(engine, np.concatenate([c.reshape(-1) for c in cols]).astype(np.float32))
    ll_cols = _dev(engine, np.array([c.shape[1] for c in cols], dtype=np.int32))
    res = engine.align(graphs, d_ll, ll_off, ll_cols, frame_off, beam=beam, retry_beam=retry, acoustic_scale=scale,
                       max_tokens=max_tokens, bp_tokens_per_frame=bp_tokens, want_frame_likes=True)
    res = {k: (v.cpu().numpy() if v is not None else None) for k, v in res.items()}
    for u, f in enumerate(fsts):
        ref = helpers.oracle_align(tm, f, cols[u], graphs.pdf_lists_host[u], acoustic_scale=scale, beam=beam, retry_beam=retry)
        want = helpers.device_status(ref, int(frame_off[u + 1] - frame_off[u]))
        assert res["status"][u] == want, (u, res["status"][u], want)
        if want in (0, 1):
            a, b = frame_off[u], frame_off[u + 1]
            if not np.array_equal(res["ali"][a:b], ref["ali"]):
                bad = np.nonzero(res["ali"][a:b] != ref["ali"])[0]
                raise AssertionError(f"utt {u}: alignment differs at {bad.size}/{b - a} frames, first {bad[:5]}, "
                                     f"gpu {res['ali'][a:b][bad[:5]]} ref {ref['ali'][bad[:5]]} beam {beam}")
            nw = res["n_words"][u]
            assert np.array_equal(res["words"][a: a + nw], ref["words"])
            assert res["like"][u] == np.float32(ref["like"]), (res["like"][u], ref["like"])
            assert np.array_equal(res["frame_like"][a:b], ref["per_frame"])
    return res


def test_viterbi_real_audio_identical_to_oracle(engine, fx):
    tm, am = fx.mono_tm, fx.mono_am
    texts = ["this is the acoustic corpus i'm talking pretty fast here", "there's nothing going else going on",
             fx.text, "um and that should be all thanks"]
    sr = 16000
    cuts = [(0.0, 4.2), (4.0, 6.5), (0.0, 26.72), (23.5, 26.72)]
    feats = [fx.mono_feats(fx.pcm[int(a * sr): int(b * sr)]) for a, b in cuts]
    pdfs = np.arange(am.num_pdfs, dtype=np.int32)
    lls = [O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pdfs) for x in feats]
    fsts = [fx.mono_graph(t) for t in texts]
    assert fsts[2].num_states > 1000  # more states than the decoder's 1000 hash buckets: bucket-order path
    for beam, retry in ((100.0, 400.0), (10.0, 40.0), (30.0, 120.0)):
        res = _align_case(engine, tm, am, fsts, lls, beam, retry, max_tokens=2048)
        assert set(res["status"].tolist()) <= {0, 1, 2}


def test_viterbi_random_scores_all_paths(engine, fx):
    tm = fx.mono_tm
    rng = np.random.default_rng(11)
    words = [w for w in fx.text.split() if fx.mono_lex.word_table.member(w)]
    for trial in range(4):
        fsts, lls = [], []
        for u in range(24):
            n = int(rng.integers(1, 9))
            text = " ".join(rng.choice(words, size=n))
            f = fx.mono_graph(text)
            nph = sum(len(fx.mono_lex.word_pronunciations(w)[0].pronunciation.split()) for w in text.split())
            T = int(3 * nph * rng.uniform(0.9, 3.0)) + int(rng.integers(0, 5))
            sd = float(rng.choice([3.0, 12.0, 30.0, 80.0]))
            fsts.append(f)
            lls.append(rng.normal(-60.0, sd, size=(max(T, 1), tm.num_pdfs)).astype(np.float32))
        beam, retry = [(0.5, 2.0), (2.0, 8.0), (10.0, 40.0), (1.0, 1000.0)][trial]
        res = _align_case(engine, tm, fx.mono_am, fsts, lls, beam, retry)
        if trial == 3:
            assert 1 in res["status"].tolist()  # the retry launch really ran


def _random_graph(rng, tm, n_states):
    """Arbitrary epsilon-free graph over the model's transition-ids: nothing HMM-shaped about it — cycles, skips, dead ends,
    out-degrees from 1 to 64, weights on a coarse grid (exact cost ties), a few final states."""
    from montreal_forced_aligner_amd import kaldi_io as K

    offs, arcs = [0], []
    for s in range(n_states):
        r = rng.random()
        deg = 64 if r < 0.01 else (int(rng.integers(9, 30)) if r < 0.05 else int(rng.integers(1, 5)))
        if s == n_states - 1 and rng.random() < 0.5:
            deg = 0   # a dead end
        for _ in range(deg):
            q = rng.random()
            dst = s if q < 0.3 else (min(n_states - 1, s + int(rng.integers(1, 4))) if q < 0.8 else int(rng.integers(0, n_states)))
            arcs.append((int(rng.integers(1, tm.num_transition_ids + 1)), int(rng.integers(0, 50)) if rng.random() < 0.2 else 0,
                         float(rng.integers(0, 12)) * 0.25, dst))
        offs.append(len(arcs))
    a = np.zeros(len(arcs), dtype=K.ARC_DTYPE)
    for i, (il, ol, w, d) in enumerate(arcs):
        a[i] = (il, ol, w, d)
    final = np.full(n_states, np.inf, dtype=np.float32)
    for s in rng.choice(n_states, size=max(1, n_states // 10), replace=False):
        final[s] = float(rng.integers(0, 4)) * 0.5
    return K.Fst(0, np.asarray(offs, dtype=np.int64), a, final)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_viterbi_random_graph_fuzz(engine, fx, seed):
    """Decoder against the oracle on graphs that share nothing with a training graph's shape: exercises list-order rules
    (cycles re-create states, many arcs into one state), exact ties, wide states (token-per-lane general path, stash),
    token-capacity escalation (hundreds of live tokens), pruning to failure and the retry launch.  Bit-exact or it fails."""
    tm = fx.mono_tm
    rng = np.random.default_rng(1000 + seed)
    fsts, lls = [], []
    for u in range(20):
        S = int(rng.choice([3, 8, 40, 150, 400, 1100])) if u else 1100
        fsts.append(_random_graph(rng, tm, S))
        T = int(rng.integers(2, 90))
        if rng.random() < 0.3:   # quantised scores: ties everywhere
            ll = (rng.integers(-240, -160, size=(T, tm.num_pdfs)) * 0.25).astype(np.float32)
        else:
            ll = rng.normal(-60.0, float(rng.choice([2.0, 10.0, 40.0])), size=(T, tm.num_pdfs)).astype(np.float32)
        lls.append(ll)
    beam, retry = [(1.0, 4.0), (8.0, 32.0), (50.0, 0.0)][seed]
    # capacities at their hard upper bounds (one token per state), so the only statuses left are the decoder's own
    res = _align_case(engine, tm, fx.mono_am, fsts, lls, beam, retry, max_tokens=2048, bp_tokens=1100)
    assert set(res["status"].tolist()) <= {0, 1, 2}
    assert (res["status"] != 2).any()


def test_viterbi_ties_and_duplicate_paths(engine, fx):
    """Constant scores make many paths cost exactly the same: the winner is decided purely by Kaldi's first-come order."""
    tm = fx.mono_tm
    fsts = [fx.mono_graph("words words words"), fx.mono_graph("this is"), fx.mono_graph("um")]
    lls = [np.full((T, tm.num_pdfs), -50.0, dtype=np.float32) for T in (90, 40, 30)]
    _align_case(engine, tm, fx.mono_am, fsts, lls, 10.0, 40.0)
    _align_case(engine, tm, fx.mono_am, fsts, lls, 1.0e4, 0.0)


def test_viterbi_token_capacity_overflow_is_reported(engine, fx):
    tm = fx.mono_tm
    rng = np.random.default_rng(5)
    f = fx.mono_graph(fx.text)
    ll = rng.normal(-60.0, 1.0, size=(400, tm.num_pdfs)).astype(np.float32)
    engine.load_gmm(fx.mono_am)
    graphs = engine.pack_graphs([f], tm)
    cols = ll[:, graphs.pdf_lists_host[0]]
    frame_off = np.array([0, 400], dtype=np.int64)
    res = engine.align(graphs, _dev(engine, cols.reshape(-1)), np.array([0, cols.size], dtype=np.int64),
                       _dev(engine, np.array([cols.shape[1]], dtype=np.int32)), frame_off, beam=1000.0, retry_beam=0.0,
                       max_tokens=64)
    assert int(res["status"].cpu()[0]) == 3  # loud, per-utterance, never a silent wrong answer


def test_end_to_end_pcm_to_alignment(engine, fx):
    """Whole device pipeline from int16 PCM versus the whole oracle pipeline."""
    tm, am = fx.mono_tm, fx.mono_am
    sr = 16000
    cuts = [(0.0, 4.2), (4.0, 6.5), (23.5, 26.72)]
    texts = ["this is the acoustic corpus i'm talking pretty fast here", "there's nothing going else going on",
             "um and that should be all thanks"]
    segs = [fx.pcm[int(a * sr): int(b * sr)] for a, b in cuts]
    engine.configure_mfcc()
    engine.load_gmm(am)
    sample_off = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
    mfcc, frame_off = engine.mfcc(_dev(engine, np.concatenate(segs)), sample_off)
    utt2spk = np.arange(3, dtype=np.int32)
    stats = engine.cmvn_stats(mfcc, frame_off, utt2spk, 3)
    feats = engine.features(mfcc, frame_off, utt2spk, stats)
    fsts = [fx.mono_graph(t) for t in texts]
    graphs = engine.pack_graphs(fsts, tm)
    ll, ll_off, ll_cols = engine.score(feats, frame_off, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts)
    res = engine.align(graphs, ll, ll_off, ll_cols, frame_off, beam=100.0, retry_beam=400.0)
    res = {k: (v.cpu().numpy() if v is not None else None) for k, v in res.items()}
    for u in range(3):
        x = fx.mono_feats(segs[u])
        pl = graphs.pdf_lists_host[u]
        ref_ll = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl)
        ref = helpers.oracle_align(tm, fsts[u], ref_ll, pl, beam=100.0, retry_beam=400.0)
        assert res["status"][u] == ref["status"] and ref["status"] in (0, 1)
        a, b = frame_off[u], frame_off[u + 1]
        assert np.array_equal(res["ali"][a:b], ref["ali"])  # frame-identical boundaries
        assert abs(res["like"][u] - ref["like"]) / (b - a) < 1e-3  # per-frame log-likelihood (what MFA reports) within 1e-3


@pytest.mark.parametrize("dim", [40, 39, 45])
def test_gmm_bf16_split_kernel_within_float32_rounding(engine, dim, monkeypatch):
    """Default scoring: 32-row pdfs (17+ Gaussians; more than 32 = several blocks merged by an online log-sum-exp) go
    through the bf16×3 MFMA kernel (products exact to 2^-24 per term, accumulation order different from the oracle's fmaf
    chain); the small-slot classes stay on the f32 kernel.  Scores must
    stay within float32 rounding noise of the oracle — an order of magnitude inside north_star's 1e-3 — and must not depend
    on the tile an utterance's frames fall in."""
    rng = np.random.default_rng(100 + dim)
    sizes = [32] * 40 + [17, 20, 31, 32, 29] + [1, 4, 8, 16, 33, 64, 70, 100, 128]
    am = helpers.random_gmm(rng, dim, sizes)
    feats = [rng.normal(0, 3, size=(t, dim)).astype(np.float32) for t in (1, 63, 64, 65, 257, 700)]
    lists = [rng.permutation(am.num_pdfs)[:n].astype(np.int32) for n in (54, 30, 45, 1, 54, 40)]
    monkeypatch.setenv("MFA_GMM_BF16", "0")
    f32_scores, sorted_lists = _score(engine, am, feats, lists)
    monkeypatch.delenv("MFA_GMM_BF16")          # default: bf16×3 for the single-block 32-row class
    got, sorted_lists2 = _score(engine, am, feats, lists)
    n_gauss = np.diff(am.pdf_offsets)
    changed = 0
    for u in range(len(feats)):
        assert np.array_equal(sorted_lists[u], sorted_lists2[u])
        ref = O.gmm_loglikes(feats[u], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, sorted_lists[u])
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got[u] - ref).max() < 4e-6 * scale, (u, float(np.abs(got[u] - ref).max()), scale)
        single = n_gauss[sorted_lists[u]] > 1    # everything but single-Gaussian pdfs takes a split-operand kernel
        # single-Gaussian columns are produced by the f32 kernel in both runs: identical bits
        assert np.array_equal(got[u][:, ~single], f32_scores[u][:, ~single])
        changed += int((got[u][:, single] != f32_scores[u][:, single]).sum())
    assert changed > 0   # the bf16 path really ran


def test_gmm_real_mixture_model_default_path(engine, fx):
    """The reference's own fixture model (80 pdfs, 1..26 Gaussians) on features computed from its fixture audio, scored
    the way a user gets it (f16×2 for every pdf of 2+ Gaussians, f32 for single Gaussians): every cell within
    1e-4 + 2e-6·|score| of the oracle (north_star: 1e-3), and the same cells again within 1e-4 when the features are 40× out of the model's range, where
    every tile is declined by the f16 pass and scored by the bf16×3 pass."""
    am = fx.g2p_am
    mf = [O.mfcc(s.astype(np.float32), O.default_mfcc_opts(snip_edges=1)) for s in _segments(fx)[:3]]
    feats = [O.affine(O.splice(O.cmvn_apply(O.cmvn_stats([m]), m)), fx.g2p_lda) for m in mf]
    lists = [np.arange(am.num_pdfs, dtype=np.int32)] * 3
    n_gauss = np.diff(am.pdf_offsets)
    assert (n_gauss > 16).any() and n_gauss.max() <= 32
    for scale_up, tol in ((1.0, 1e-4), (40.0, None)):
        fs = [(f * scale_up).astype(np.float32) for f in feats]
        got, sorted_lists = _score(engine, am, fs, lists)
        for u in range(3):
            ref = O.gmm_loglikes(fs[u], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, sorted_lists[u])
            err = np.abs(got[u] - ref)
            if tol is not None:   # 1e-4 on ordinary scores; the fixture model also has cells of magnitude 1e4 (far-off pdfs)
                assert (err <= tol + 2e-6 * np.abs(ref)).all(), (u, float(err.max()), float(np.abs(ref).max()))
            else:   # scores of magnitude 1e5: float32 spacing is 8e-3 there
                assert (err / np.maximum(1.0, np.abs(ref))).max() < 1e-6, (u, float((err / np.abs(ref)).max()))


@pytest.mark.parametrize("dim", [40, 45])
def test_gmm_f16_split_kernel_within_tolerance_and_range_fallback(engine, dim, monkeypatch):
    """Default scoring of a model without multi-block pdfs: every class of 2..32 Gaussians goes through the f16×2 MFMA
    kernels (three products per term, operands scaled by model-derived powers of two; worst case 3·2^-22 per term; the
    16/8/4-row classes as gathered virtual blocks), single-Gaussian pdfs through the bit-exact f32 kernel.  Scores stay within
    2e-5 × scale of the oracle per (frame, pdf) cell — north_star allows 1e-3 on a log-likelihood — and a 256-frame tile
    holding a feature value that leaves the f16 range after scaling is scored by the bf16×3 kernel instead: its cells are
    bit-identical to a run with MFA_GMM_F16=0."""
    monkeypatch.delenv("MFA_GMM_BF16", raising=False)   # this test is about the default path, whatever the caller's shell says
    rng = np.random.default_rng(300 + dim)
    sizes = [32] * 40 + [17, 20, 31, 32, 29] + [1, 4, 8, 16]
    am = helpers.random_gmm(rng, dim, sizes)
    feats = [rng.normal(0, 3, size=(t, dim)).astype(np.float32) for t in (1, 63, 64, 65, 257, 700)]
    feats[5][300, 3] = 3.0e5        # tile 1 of the last utterance (frames 256..511) leaves the f16 range
    lists = [rng.permutation(am.num_pdfs)[:n].astype(np.int32) for n in (49, 30, 45, 1, 49, 40)]
    monkeypatch.setenv("MFA_GMM_F16", "0")
    bf16_scores, sorted_lists = _score(engine, am, feats, lists)
    monkeypatch.delenv("MFA_GMM_F16")
    got, sorted_lists2 = _score(engine, am, feats, lists)
    n_gauss = np.diff(am.pdf_offsets)
    changed, worst = 0, 0.0
    for u in range(len(feats)):
        assert np.array_equal(sorted_lists[u], sorted_lists2[u])
        ref = O.gmm_loglikes(feats[u], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, sorted_lists[u])
        scale = np.maximum(1.0, np.abs(ref).max(axis=1, keepdims=True))     # per frame: the outlier frame is its own scale
        err = np.abs(got[u] - ref) / scale
        worst = max(worst, float(err.max()))
        at = np.unravel_index(np.argmax(np.nan_to_num(err, nan=np.inf)), err.shape)
        assert err.max() < 2e-5, (u, at, float(got[u][at]), float(ref[at]), float(bf16_scores[u][at]))
        single = n_gauss[sorted_lists[u]] > 1    # every class but the single-Gaussian one takes the split-operand kernels
        assert np.array_equal(got[u][:, ~single], bf16_scores[u][:, ~single])   # f32 kernel in both runs: bit-exact class
        if u == 5:
            assert np.array_equal(got[u][256:512], bf16_scores[u][256:512])     # the declined tile
            keep = np.r_[0:256, 512:700]
            changed += int((got[u][keep][:, single] != bf16_scores[u][keep][:, single]).sum())
        else:
            changed += int((got[u][:, single] != bf16_scores[u][:, single]).sum())
    assert changed > 0   # the f16 path really ran
    print(f"f16x2 worst |err|/scale = {worst:.3g}")


@pytest.mark.parametrize("sizes", [[1, 2, 3, 4, 5, 8, 9, 12, 16, 17, 26, 32, 32, 4, 8, 16] * 4,
                                   [1, 4, 8, 16, 20, 32, 33, 64, 70, 2, 6, 11] * 4])
def test_reachability_keys_across_all_slot_classes(engine, sizes):
    """Scoring with per-pdf first-frame keys on models that exercise every kernel family at once (single Gaussians on
    the f32 kernel; 4-, 8-, 16-row classes as gathered virtual blocks; single 32-row blocks; multi-block pdfs): every
    cell at or after its pdf's key must carry exactly the bits of the dense run, every cell written at all must be the
    dense run's value, and cells are in fact skipped."""
    rng = np.random.default_rng(len(sizes) + sizes[5])
    dim = 40
    am = helpers.random_gmm(rng, dim, sizes)
    engine.load_gmm(am)
    Ts = (700, 257, 64, 1)
    feats = [rng.normal(0, 3, size=(t, dim)).astype(np.float32) for t in Ts]
    lists, keys, counts = [], [], []
    for t in Ts:
        pl = rng.permutation(am.num_pdfs)[: max(1, am.num_pdfs - 5)].astype(np.int32)
        ff = rng.integers(0, t + 40, size=len(pl)).astype(np.int32)   # some pdfs are never reachable in this utterance
        pl, cc, ff = engine.sort_pdf_list(pl, first_frame=ff)
        lists.append(pl); keys.append(ff); counts.append(cc)
    frame_off = np.concatenate([[0], np.cumsum(Ts)]).astype(np.int64)
    pdf_off = np.concatenate([[0], np.cumsum([len(p) for p in lists])]).astype(np.int64)
    d_feats = _dev(engine, np.concatenate(feats))
    d_list, d_cc = _dev(engine, np.concatenate(lists)), _dev(engine, np.stack(counts).astype(np.int32))
    dense, ll_off, _ = engine.score(d_feats, frame_off, d_list, pdf_off, d_cc)
    sparse, _, _ = engine.score(d_feats, frame_off, d_list, pdf_off, d_cc, pdf_first_frame=_dev(engine, np.concatenate(keys)))
    dense, sparse = dense.cpu().numpy(), sparse.cpu().numpy()
    skipped = 0
    for u, t in enumerate(Ts):
        P = len(lists[u])
        du, su = dense[ll_off[u]: ll_off[u + 1]].reshape(t, P), sparse[ll_off[u]: ll_off[u + 1]].reshape(t, P)
        ref = O.gmm_loglikes(feats[u], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, lists[u])
        assert np.abs(du - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
        tt = np.arange(t)[:, None]
        needed = tt >= keys[u][None, :]
        assert np.array_equal(su[needed], du[needed])
        written = su != 0.0                                      # a kernel may score a little more than it must (pdfs
        assert np.array_equal(su[written], du[written])           # sharing a block): whatever it writes is the real score
        early = (tt // 64 + 1) * 64 - 1 < keys[u][None, :]        # the whole 64-frame tile lies before the key
        skipped += int((early & ~written).sum())
    assert skipped > 0


def test_reachability_bounded_scoring_changes_nothing(engine, fx):
    """Scoring with pdf_first_frame skips (frame, pdf) cells no decoder token can ask for: every cell it does write is
    bit-identical to the dense matrix, every cell at or after the pdf's first frame is written, and the alignment that
    comes out of the sparse matrix is identical to the one from the dense matrix."""
    tm, am = fx.mono_tm, fx.mono_am
    sr = 16000
    cuts = [(0.0, 4.2), (4.0, 6.5), (23.5, 26.72)]
    texts = ["this is the acoustic corpus i'm talking pretty fast here", "there's nothing going else going on",
             "um and that should be all thanks"]
    segs = [fx.pcm[int(a * sr): int(b * sr)] for a, b in cuts]
    engine.configure_mfcc()
    engine.load_gmm(am)
    sample_off = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
    mfcc, frame_off = engine.mfcc(_dev(engine, np.concatenate(segs)), sample_off)
    utt2spk = np.arange(3, dtype=np.int32)
    feats = engine.features(mfcc, frame_off, utt2spk, engine.cmvn_stats(mfcc, frame_off, utt2spk, 3))
    fsts = [fx.mono_graph(t) for t in texts]
    graphs = engine.pack_graphs(fsts, tm, cluster_gap=None)   # one column per pdf: the key check below assumes it
    dense, ll_off, ll_cols = engine.score(feats, frame_off, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts)
    sparse, _, _ = engine.score(feats, frame_off, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts,
                                pdf_first_frame=graphs.pdf_first_frame)
    d, s = dense.cpu().numpy(), sparse.cpu().numpy()
    skipped = 0
    for u in range(3):
        T, P = int(frame_off[u + 1] - frame_off[u]), len(graphs.pdf_lists_host[u])
        du, su = d[ll_off[u]: ll_off[u + 1]].reshape(T, P), s[ll_off[u]: ll_off[u + 1]].reshape(T, P)
        ff = graphs.pdf_first_frame_host[u]
        # host-side check of the keys themselves: state depths by a plain numpy relaxation
        f = fsts[u]
        depth = np.full(f.num_states, 1 << 30, dtype=np.int64)
        depth[f.start] = 0
        src = np.repeat(np.arange(f.num_states), np.diff(f.arc_offsets))
        for _ in range(f.num_states):
            nd = depth.copy()
            np.minimum.at(nd, f.arcs["nextstate"], depth[src] + 1)
            if np.array_equal(nd, depth):
                break
            depth = nd
        col_of_arc = {int(p): i for i, p in enumerate(graphs.pdf_lists_host[u])}
        want = np.full(P, 1 << 30, dtype=np.int64)
        for a in range(f.num_arcs):
            c = col_of_arc[int(tm.id2pdf[f.arcs["ilabel"][a]])]
            want[c] = min(want[c], depth[src[a]])
        assert np.array_equal(want, ff)
        needed = np.arange(T)[:, None] >= ff[None, :]
        assert np.array_equal(du[needed], su[needed])          # written cells are the dense values, bit for bit
        written = su != 0.0
        assert np.array_equal(du[written], su[written])
        skipped += int((~written).sum())
    assert skipped > 0  # the bound actually removes work on these graphs
    ra = engine.align(graphs, dense, ll_off, ll_cols, frame_off, beam=100.0, retry_beam=400.0, want_frame_likes=True)
    rb = engine.align(graphs, sparse, ll_off, ll_cols, frame_off, beam=100.0, retry_beam=400.0, want_frame_likes=True)
    for k in ("ali", "words", "n_words", "like", "status", "frame_like"):
        assert torch.equal(ra[k], rb[k]), k


def test_fmllr_statistics_match_oracle(engine, fx):
    """First-pass alignment → per-speaker fMLLR statistics on the device vs the oracle's accumulation, and the
    host solve on them vs the oracle's solve."""
    from montreal_forced_aligner_amd import fmllr as F
    from montreal_forced_aligner_amd.engine import fmllr_statistics

    am, tm = fx.g2p_am, fx.g2p_tm
    rng = np.random.default_rng(3)
    mf = [O.mfcc(s.astype(np.float32), O.default_mfcc_opts(snip_edges=1)) for s in _segments(fx)[:3]]
    feats = [O.affine(O.splice(O.cmvn_apply(O.cmvn_stats([m]), m)), fx.g2p_lda) for m in mf]
    frame_off = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int64)
    # a synthetic "alignment": random transition-ids of the model (what matters here is the statistics, not the path)
    alis = [rng.integers(1, tm.num_transition_ids + 1, size=f.shape[0]).astype(np.int32) for f in feats]
    alis[1][:5] = 0  # unaligned frames carry no weight
    utt2spk = np.array([7, 3, 7])
    sil_phones = [1, 2]
    engine.load_gmm(am)
    spk_ids, beta, K, G = fmllr_statistics(engine, _dev(engine, np.concatenate(feats)), frame_off, _dev(engine, np.concatenate(alis)),
                                           tm, utt2spk, sil_phones)
    assert spk_ids.tolist() == [3, 7]
    for k, spk in enumerate(spk_ids):
        stats = None
        for u in range(3):
            if utt2spk[u] != spk:
                continue
            ali = alis[u]
            w = np.where((ali == 0) | np.isin(tm.id2phone[ali], sil_phones), 0.0, 1.0).astype(np.float32)
            stats = O.fmllr_acc(feats[u], np.maximum(tm.id2pdf[ali], 0), w, am.gconsts, am.means_invvars, am.inv_vars,
                                am.pdf_offsets, stats)
        rb, rK, rG = stats[0][0], stats[1], stats[2]
        assert abs(beta[k] - rb) < 1e-3 * max(1.0, rb)
        assert np.allclose(K[k], rK, rtol=1e-4, atol=1e-2) and np.allclose(G[k], rG, rtol=1e-4, atol=1e-2)
        Wd, impr_d = F.compute_fmllr(beta[k], K[k], G[k], min_count=50.0)
        Wo, impr_o = O.fmllr_solve(rb, rK, rG, min_count=50.0)
        assert abs(impr_d - impr_o) < 1e-3 * max(1.0, abs(impr_o)) and np.abs(Wd - Wo).max() < 1e-3
