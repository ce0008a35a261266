"""The headline configuration end to end against the oracle: context-dependent 32-Gaussian model in LDA+fMLLR space
(BASELINE configs[2] shape, smaller pdf inventory so the CPU oracle finishes in seconds), 10 s synthetic utterances, whole
device path from PCM — MFCC → CMVN → splice+LDA+fMLLR → bf16×3 scoring with reachability → Viterbi with capacity tiers —
versus the oracle's whole path from the same PCM.  Also the ragged case of configs[4]: utterances of 1 – 30 s in one batch.

Bars (north_star): boundaries frame-identical, per-frame log-likelihood within 1e-3."""
import numpy as np
import pytest
import torch

from montreal_forced_aligner_amd import graph as G
from oracle import oracle as O
from tests import helpers, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(engine):
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


def _oracle_path(pcm, lda, fm_spk, model, fst, pl):
    mf = O.mfcc(pcm.astype(np.float32), O.default_mfcc_opts())
    x = O.affine(O.affine(O.splice(O.cmvn_apply(O.cmvn_stats([mf]), mf)), lda), fm_spk)
    am = model.am
    ll = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl)
    return helpers.oracle_align(model.tm, fst, ll, pl, beam=10.0, retry_beam=40.0), ll


def _device_path(engine, model, feats_of, utts, fsts):
    engine.load_gmm(model.am)
    feats, fo = feats_of([u[0] for u in utts], [u[3] for u in utts])
    graphs = engine.pack_graphs(fsts, model.tm)
    ll, ll_off, ll_cols = engine.score(feats, fo, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts,
                                       pdf_first_frame=graphs.pdf_first_frame)
    res = engine.align(graphs, ll, ll_off, ll_cols, fo, beam=10.0, retry_beam=40.0, max_tokens=1024, bp_tokens_per_frame=256)
    return {k: v.cpu().numpy() for k, v in res.items() if v is not None}, fo, graphs, ll.cpu().numpy(), ll_off


def test_triphone_fmllr_pipeline_matches_oracle_from_pcm(engine, setup):
    world, model, lda, fm, feats_of = setup
    assert int(np.diff(model.am.pdf_offsets).min()) == 32          # every pdf is a single 32-row block: bf16×3 kernel
    utts = [world.utterance(7000 + i) for i in range(6)]
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    res, fo, graphs, ll, ll_off = _device_path(engine, model, feats_of, utts, fsts)
    assert np.all(res["status"] == 0)
    worst_score = 0.0
    for u, (pcm, text, segs, spk) in enumerate(utts):
        pl = graphs.pdf_lists_host[u]
        ref, ref_ll = _oracle_path(pcm, lda, fm[spk % 16], model, fsts[u], pl)
        a, b = fo[u], fo[u + 1]
        assert ref["status"] == 0
        assert np.array_equal(res["ali"][a:b], ref["ali"]), f"utterance {u}: boundaries differ from the oracle"
        nw = int(res["n_words"][u])
        assert np.array_equal(res["words"][a: a + nw], ref["words"])
        assert abs(res["like"][u] - ref["like"]) / (b - a) < 1e-3
        # score matrix (where written) against the oracle's from ITS OWN features: MFCC FFT differences included
        T, P = b - a, len(pl)
        got = ll[ll_off[u]: ll_off[u + 1]].reshape(T, P)
        ff = graphs.pdf_first_frame_host[u]
        need = np.arange(T)[:, None] >= (ff[None, :] // 64 + 1) * 64   # cells every tile certainly wrote
        worst_score = max(worst_score, float(np.abs(got[need] - ref_ll[need]).max()))
    assert worst_score < 5e-2   # int16-scale audio through two different float32 FFTs moves a 40-dim score by ~1e-2


def test_ragged_batch_1_to_30_seconds(engine, setup):
    """configs[4] shape: one batch holding utterances from 1 s to 30 s (3 000 frames, 12 scoring tiles, long back-pointer
    trails) — every utterance must come out exactly as when it is aligned alone, and the long one as the oracle has it."""
    world, model, lda, fm, feats_of = setup
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    shapes = [(3, 16000), (8, 43200), (30, 160000), (90, 480000), (1, 8000), (30, 160001)]
    utts = [world.utterance(8000 + i, n_words=nw, samples=ns) for i, (nw, ns) in enumerate(shapes)]
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    res, fo, graphs, _ll, _off = _device_path(engine, model, feats_of, utts, fsts)
    assert np.all((res["status"] == 0) | (res["status"] == 1)), res["status"]
    assert list(np.diff(fo)) == [100, 270, 1000, 3000, 50, 1000]
    for u in range(len(utts)):
        solo, fo1, _g, _l, _o = _device_path(engine, model, feats_of, [utts[u]], [fsts[u]])
        assert solo["status"][0] == res["status"][u]
        assert np.array_equal(solo["ali"], res["ali"][fo[u]: fo[u + 1]]), u
        assert solo["like"][0] == res["like"][u]
    pcm, text, segs, spk = utts[3]
    ref, _ = _oracle_path(pcm, lda, fm[spk % 16], model, fsts[3], graphs.pdf_lists_host[3])
    assert ref["status"] == res["status"][3] and np.array_equal(ref["ali"], res["ali"][fo[3]: fo[4]])


def test_corpus_aligner_two_pass_flow(engine, setup):
    """CorpusAligner(speaker_adapted=True) = first pass → per-speaker fMLLR statistics on the device → host solve →
    second pass with the transforms (MFA/alignment/base.py:510-539).  The transforms it used must be exactly what the
    engine-level pieces give for the first-pass alignments, and the second pass must still spell every transcript."""
    from montreal_forced_aligner_amd import fmllr as F
    from montreal_forced_aligner_amd.aligner import AlignOptions, CorpusAligner, CorpusUtterance
    from montreal_forced_aligner_amd.engine import fmllr_statistics

    world, model, lda, fm, feats_of = setup
    pt = world.lexicon.phone_table
    sil = [pt.find("sil"), pt.find("spn")]
    utts = []
    for i in range(8):
        pcm, text, segs, spk = world.utterance(9000 + i, speaker=3 + (i % 2))
        utts.append(CorpusUtterance(f"s{spk}-{i}", f"s{spk}", pcm, text))
    al = CorpusAligner(model.tm, model.am, model.tree, world.lexicon, lda=lda, engine=engine,
                       options=AlignOptions(beam=10.0, retry_beam=40.0), silence_phones=sil)
    first = al.align(utts, speaker_adapted=False, make_ctm=False)
    second = al.align(utts, speaker_adapted=True)
    assert all(r is not None for r in first) and all(r is not None for r in second)
    # (the aligned log-likelihood omits fMLLR's log-determinant term, so it need not rise with the transform)
    W = al.transforms
    assert W.shape == (2, 40, 41) and np.isfinite(W).all()
    assert np.abs(W[:, :, :40] - np.eye(40)).max() > 0.05          # 4 000 frames per speaker: well above min_count
    # the same statistics and solve, by hand, from the first-pass alignments
    spk_ids, cmvn = al.speaker_cmvn(utts)
    so = np.concatenate([[0], np.cumsum([len(u.pcm) for u in utts])]).astype(np.int64)
    mfcc, fo = engine.mfcc(torch.from_numpy(np.concatenate([u.pcm for u in utts])).to(engine.device), so)
    rows = np.array([spk_ids[u.speaker] for u in utts], dtype=np.int32)
    feats = engine.features(mfcc, fo, rows, cmvn, lda=torch.from_numpy(lda).to(engine.device))
    ali = torch.from_numpy(np.concatenate([r.alignment for r in first]).astype(np.int32)).to(engine.device)
    ids, beta, K, G = fmllr_statistics(engine, feats, fo, ali, model.tm, rows, sil, 0.0)
    for k, s in enumerate(ids):
        Wk, impr = F.compute_fmllr(beta[k], K[k], G[k], min_count=500.0)
        assert impr > 0 and np.array_equal(Wk, W[s])
    for u, r in zip(utts, second):
        words = [w.label for w in r.ctm.word_intervals if w.label != world.lexicon.silence_word]
        assert words == u.text.split()
