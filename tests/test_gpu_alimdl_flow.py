"""The reference's two-pass flow for acoustic models that ship an alignment model (final.alimdl), on the device:
first pass on final.alimdl (MFA/alignment/mixins.py:404-410), fMLLR statistics with posteriors from final.alimdl and
means/variances from final.mdl (FmllrComputer(ali_model_path, model_path, …), MFA/corpus/features.py:503-511), second pass on
final.mdl, first-pass results kept for utterances the second pass loses (MFA/alignment/multiprocessing.py:841-863,
:1784-1860); and boost_silence (MFA/alignment/multiprocessing.py:803-815)."""
import copy

import numpy as np
import pytest
import torch

from montreal_forced_aligner_amd import fmllr as F
from montreal_forced_aligner_amd import model as M
from montreal_forced_aligner_amd.aligner import AlignOptions, CorpusAligner, CorpusUtterance
from montreal_forced_aligner_amd.engine import fmllr_statistics
from oracle import oracle as O
from tests import synth
from tests.test_gpu_parity import _dev, _segments

pytestmark = pytest.mark.gpu


def test_two_model_fmllr_statistics_match_oracle(engine, fx):
    """acoustic_g2p_output_model.zip ships both final.mdl and final.alimdl (same Gaussian layout): device statistics with
    the alignment model loaded and final.mdl as the statistics model vs the oracle's two-model accumulation."""
    tm, am = fx.g2p_tm, fx.g2p_am
    tm_a, am_a = M.load_model_bytes(fx.g2p_archive["final.alimdl"])
    assert np.array_equal(am_a.pdf_offsets, am.pdf_offsets) and tm_a.num_transition_ids == tm.num_transition_ids
    rng = np.random.default_rng(4)
    if np.array_equal(am_a.means_invvars, am.means_invvars):     # the fixture's two models may coincide: force a difference
        am = copy.copy(am)
        am.means_invvars = (am.means_invvars * (1.0 + 0.05 * rng.normal(size=am.means_invvars.shape))).astype(np.float32)
        am.inv_vars = (am.inv_vars * rng.uniform(0.8, 1.25, size=am.inv_vars.shape)).astype(np.float32)
    mf = [O.mfcc(s.astype(np.float32), O.default_mfcc_opts(snip_edges=1)) for s in _segments(fx)[:3]]
    feats = [O.affine(O.splice(O.cmvn_apply(O.cmvn_stats([m]), m)), fx.g2p_lda) for m in mf]
    frame_off = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int64)
    alis = [rng.integers(1, tm.num_transition_ids + 1, size=f.shape[0]).astype(np.int32) for f in feats]
    alis[2][-7:] = 0
    utt2spk = np.array([1, 0, 1])
    sil_phones = [1, 2]
    engine.load_gmm(am_a)                       # posteriors: the alignment model
    ids, beta, K, G = fmllr_statistics(engine, _dev(engine, np.concatenate(feats)), frame_off, _dev(engine, np.concatenate(alis)),
                                       tm, utt2spk, sil_phones, stats_model=am)
    one, b1, K1, G1 = fmllr_statistics(engine, _dev(engine, np.concatenate(feats)), frame_off, _dev(engine, np.concatenate(alis)),
                                       tm, utt2spk, sil_phones)          # back to the single-model form
    assert not np.allclose(K, K1)               # the statistics model really is used
    for k, spk in enumerate(ids):
        st2, st1 = None, None
        for u in range(3):
            if utt2spk[u] != spk:
                continue
            ali = alis[u]
            w = np.where((ali == 0) | np.isin(tm.id2phone[ali], sil_phones), 0.0, 1.0).astype(np.float32)
            pdf = np.maximum(tm.id2pdf[ali], 0)
            st2 = O.fmllr_acc(feats[u], pdf, w, am_a.gconsts, am_a.means_invvars, am_a.inv_vars, am_a.pdf_offsets, st2,
                              stat_means_invvars=am.means_invvars, stat_inv_vars=am.inv_vars)
            st1 = O.fmllr_acc(feats[u], pdf, w, am_a.gconsts, am_a.means_invvars, am_a.inv_vars, am_a.pdf_offsets, st1)
        for (rb, rK, rG), (db, dK, dG) in (((st2[0][0], st2[1], st2[2]), (beta[k], K[k], G[k])),
                                           ((st1[0][0], st1[1], st1[2]), (b1[k], K1[k], G1[k]))):
            assert abs(db - rb) < 1e-3 * max(1.0, rb)
            assert np.allclose(dK, rK, rtol=1e-4, atol=1e-2) and np.allclose(dG, rG, rtol=1e-4, atol=1e-2)


@pytest.fixture(scope="module")
def sat(engine):
    world = synth.SynthWorld.build()
    engine.configure_mfcc()
    lda = synth.seeded_lda()
    d_lda = torch.from_numpy(lda).to(engine.device)

    def si_feats(pcm):
        so = np.array([0, len(pcm)], dtype=np.int64)
        mfcc, fo = engine.mfcc(torch.from_numpy(pcm).to(engine.device), so)
        own = np.zeros(1, dtype=np.int32)
        return engine.features(mfcc, fo, own, engine.cmvn_stats(mfcc, fo, own, 1), lda=d_lda).cpu().numpy()

    model = synth.train_triphone(world, lambda pcm, spk: si_feats(pcm), n_train=40, n_gauss=8, n_classes=2)
    # a speaker-independent "alignment model" with the same layout: broader variances, slightly shifted means
    rng = np.random.default_rng(21)
    ali_am = copy.copy(model.am)
    var = 1.0 / model.am.inv_vars
    mean = model.am.means_invvars * var
    var2 = var * 1.3
    mean2 = mean + 0.05 * np.sqrt(var) * rng.normal(size=mean.shape)
    w_log = model.am.gconsts + 0.5 * (model.am.dim * np.log(2 * np.pi) + np.log(var).sum(axis=1) + (mean * mean / var).sum(axis=1))
    ali_am.inv_vars = (1.0 / var2).astype(np.float32)
    ali_am.means_invvars = (mean2 / var2).astype(np.float32)
    ali_am.gconsts = (w_log - 0.5 * (model.am.dim * np.log(2 * np.pi) + np.log(var2).sum(axis=1) + (mean2 * mean2 / var2).sum(axis=1))).astype(np.float32)
    return world, model, ali_am, lda


def _utts(world, n=8):
    out = []
    for i in range(n):
        pcm, text, segs, spk = world.utterance(9100 + i, speaker=3 + (i % 2))
        out.append(CorpusUtterance(f"s{spk}-{i}", f"s{spk}", pcm, text))
    return out


def test_alimdl_first_pass_two_model_transforms_and_fallback(engine, sat):
    world, model, ali_am, lda = sat
    pt = world.lexicon.phone_table
    sil = [pt.find("sil"), pt.find("spn")]
    utts = _utts(world)
    opts = AlignOptions(beam=10.0, retry_beam=40.0)
    al = CorpusAligner(model.tm, model.am, model.tree, world.lexicon, lda=lda, engine=engine, options=opts, silence_phones=sil,
                       ali_am=ali_am)
    si_only = al.align(utts, speaker_adapted=False, make_ctm=False)           # one pass, on the alignment model
    plain = CorpusAligner(model.tm, ali_am, model.tree, world.lexicon, lda=lda, engine=engine, options=opts, silence_phones=sil)
    ref_first = plain.align(utts, speaker_adapted=False, make_ctm=False)
    assert all(a is not None and b is not None and np.array_equal(a.alignment, b.alignment) and a.likelihood == b.likelihood
               for a, b in zip(si_only, ref_first))
    second = al.align(utts, speaker_adapted=True)
    assert all(r is not None for r in second) and al.fallback_first_pass == [] and al.failed == []
    W = al.transforms
    # the same transforms by hand: first-pass alignments of the alignment model, two-model statistics, host solve
    spk_ids, cmvn = al.speaker_cmvn(utts)
    so = np.concatenate([[0], np.cumsum([len(u.pcm) for u in utts])]).astype(np.int64)
    mfcc, fo = engine.mfcc(torch.from_numpy(np.concatenate([u.pcm for u in utts])).to(engine.device), so)
    rows = np.array([spk_ids[u.speaker] for u in utts], dtype=np.int32)
    feats = engine.features(mfcc, fo, rows, cmvn, lda=torch.from_numpy(lda).to(engine.device))
    ali = torch.from_numpy(np.concatenate([r.alignment for r in ref_first]).astype(np.int32)).to(engine.device)
    engine.load_gmm(ali_am)
    ids, beta, K, G = fmllr_statistics(engine, feats, fo, ali, model.tm, rows, sil, 0.0, stats_model=model.am)
    ids1, beta1, K1, G1 = fmllr_statistics(engine, feats, fo, ali, model.tm, rows, sil, 0.0)
    for k, s in enumerate(ids):
        Wk, impr = F.compute_fmllr(beta[k], K[k], G[k], min_count=500.0)
        assert impr > 0 and np.array_equal(Wk, W[s])
        W1, _ = F.compute_fmllr(beta1[k], K1[k], G1[k], min_count=500.0)
        assert not np.array_equal(W1, W[s])                                    # not the single-model estimate
    for u, r in zip(utts, second):
        words = [w.label for w in r.ctm.word_intervals if w.label != world.lexicon.silence_word]
        assert words == u.text.split()
    # the model the caller handed in is untouched, and a second aligner on it sees the same Gaussians
    al._load(al.am)

    class LosesSecondPass(CorpusAligner):
        def _pass(self, utts_, spk_ids_, cmvn_, fmllr, want_feats=False, **kw):
            res, kept = super()._pass(utts_, spk_ids_, cmvn_, fmllr, want_feats, **kw)
            if fmllr is not None:
                res[2] = None
                res[5] = None
            return res, kept

    lossy = LosesSecondPass(model.tm, model.am, model.tree, world.lexicon, lda=lda, engine=engine, options=opts,
                            silence_phones=sil, ali_am=ali_am)
    out = lossy.align(utts, speaker_adapted=True, make_ctm=False)
    assert lossy.fallback_first_pass == [utts[2].utt_id, utts[5].utt_id] and lossy.failed == []
    for i, r in enumerate(out):
        want = ref_first[i] if i in (2, 5) else second[i]
        assert np.array_equal(r.alignment, want.alignment) and r.likelihood == want.likelihood
    # without an alignment model there is nothing to fall back to (the reference overwrites pass 1's files)
    lossy2 = LosesSecondPass(model.tm, model.am, model.tree, world.lexicon, lda=lda, engine=engine, options=opts,
                             silence_phones=sil)
    out2 = lossy2.align(utts, speaker_adapted=True, make_ctm=False)
    assert out2[2] is None and out2[5] is None and lossy2.failed == [utts[2].utt_id, utts[5].utt_id]


def test_boost_silence_matches_oracle_and_leaves_the_callers_model_alone(engine, sat):
    """GmmAligner.boost_silence(f, silence phones): the weights of the silence pdfs × f without renormalising, i.e.
    gconst += ln f (MFA/alignment/multiprocessing.py:803-815)."""
    world, model, ali_am, lda = sat
    pt = world.lexicon.phone_table
    sil = [pt.find("sil"), pt.find("spn")]
    before = model.am.gconsts.copy()
    al = CorpusAligner(model.tm, model.am, model.tree, world.lexicon, lda=lda, engine=engine,
                       options=AlignOptions(boost_silence=1.5), silence_phones=sil)
    assert np.array_equal(model.am.gconsts, before)                         # boosted on a copy
    sil_pdfs = M.pdfs_of_phones(model.tm, sil)
    is_sil = np.zeros(model.am.num_gauss, dtype=bool)
    for p in sil_pdfs:
        is_sil[model.am.pdf_offsets[p]: model.am.pdf_offsets[p + 1]] = True
    assert is_sil.any() and not is_sil.all()
    want = before.copy()
    want[is_sil] = (want[is_sil] + np.float32(np.log(np.float32(1.5)))).astype(np.float32)
    assert np.array_equal(al.am.gconsts, want)
    # device scores of the boosted model vs the oracle's on gconst + ln f
    utts = _utts(world, 2)
    so = np.concatenate([[0], np.cumsum([len(u.pcm) for u in utts])]).astype(np.int64)
    mfcc, fo = engine.mfcc(torch.from_numpy(np.concatenate([u.pcm for u in utts])).to(engine.device), so)
    own = np.arange(2, dtype=np.int32)
    feats = engine.features(mfcc, fo, own, engine.cmvn_stats(mfcc, fo, own, 2), lda=torch.from_numpy(lda).to(engine.device))
    pdfs = np.array(sorted(set(sil_pdfs) | set(range(0, model.am.num_pdfs, 37))), dtype=np.int32)
    pl, cc = engine.sort_pdf_list(pdfs)
    ll, ll_off, _ = engine.score(feats, fo, _dev(engine, np.tile(pl, 2)), np.array([0, len(pl), 2 * len(pl)], dtype=np.int64),
                                 _dev(engine, np.tile(cc, (2, 1)).astype(np.int32)))
    x = feats.cpu().numpy()
    for u in range(2):
        T = int(fo[u + 1] - fo[u])
        got = ll.cpu().numpy()[ll_off[u]: ll_off[u + 1]].reshape(T, len(pl))
        ref = O.gmm_loglikes(x[fo[u]: fo[u + 1]], want, model.am.means_invvars, model.am.inv_vars, model.am.pdf_offsets, pl)
        unb = O.gmm_loglikes(x[fo[u]: fo[u + 1]], before, model.am.means_invvars, model.am.inv_vars, model.am.pdf_offsets, pl)
        assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
        sil_cols = np.isin(pl, sil_pdfs)
        assert np.allclose((ref - unb)[:, sil_cols], np.log(1.5), atol=1e-4) and np.allclose((ref - unb)[:, ~sil_cols], 0.0, atol=1e-6)
    res = al.align(utts, make_ctm=False)
    assert all(r is not None for r in res)
