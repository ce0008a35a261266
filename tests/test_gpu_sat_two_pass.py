"""Two-pass speaker-adapted alignment on the device (the reference's CorpusAligner.align flow for SAT models,
MFA/alignment/base.py:510-539): pass 1 → per-speaker fMLLR statistics (GPU) → host solve → pass 2 with the transforms.

Synthetic check of the defining property: speakers whose features were pushed through a known affine distortion get it
undone — the estimated transform composed with the distortion is close to identity and the aligned log-likelihood rises."""
import numpy as np
import pytest
import torch

from montreal_forced_aligner_amd import fmllr as F
from montreal_forced_aligner_amd import graph as G
from montreal_forced_aligner_amd.engine import fmllr_statistics
from tests import synth

pytestmark = pytest.mark.gpu


def _compose(W2, W1):
    """x → W2 [W1 [x;1]; 1]"""
    D = W1.shape[0]
    A = W2[:, :D] @ W1[:, :D]
    b = W2[:, :D] @ W1[:, D] + W2[:, D]
    return np.concatenate([A, b[:, None]], axis=1).astype(np.float32)


def test_two_pass_fmllr_recovers_speaker_distortion(engine):
    world = synth.SynthWorld.build()
    engine.configure_mfcc()
    lda = synth.seeded_lda()
    d_lda = torch.from_numpy(lda).to(engine.device)
    dev = engine.device

    def lda_feats(pcm_list, fm=None, u2s=None):
        sample_off = np.concatenate([[0], np.cumsum([len(p) for p in pcm_list])]).astype(np.int64)
        mfcc, frame_off = engine.mfcc(torch.from_numpy(np.concatenate(pcm_list)).to(dev), sample_off)
        own = np.arange(len(pcm_list), dtype=np.int32)
        stats = engine.cmvn_stats(mfcc, frame_off, own, len(pcm_list))
        if fm is None:
            return engine.features(mfcc, frame_off, own, stats, lda=d_lda), frame_off
        # CMVN per utterance, fMLLR per speaker: two calls share the CMVN by giving every utterance its own transform row
        per_utt = torch.from_numpy(fm[u2s]).to(dev)
        return engine.features(mfcc, frame_off, own, stats, lda=d_lda, fmllr=per_utt), frame_off

    model = synth.train_triphone(world, lambda pcm, spk: lda_feats([pcm])[0].cpu().numpy(), n_train=40, n_gauss=8)
    engine.load_gmm(model.am)
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)

    n_spk, per_spk = 3, 8
    rng = np.random.default_rng(5)
    distort = np.stack([np.concatenate([np.eye(40) + 0.06 * rng.normal(size=(40, 40)), 0.8 * rng.normal(size=(40, 1))], axis=1)
                        for _ in range(n_spk)]).astype(np.float32)
    utts = [world.utterance(5000 + i, speaker=i % n_spk) for i in range(n_spk * per_spk)]
    u2s = np.array([u[3] for u in utts], dtype=np.int32)
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    graphs = engine.pack_graphs(fsts, model.tm)

    def align_with(fm):
        feats, frame_off = lda_feats([u[0] for u in utts], fm, u2s)
        ll, ll_off, ll_cols = engine.score(feats, frame_off, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts)
        res = engine.align(graphs, ll, ll_off, ll_cols, frame_off, beam=10.0, retry_beam=40.0, max_tokens=512)
        return feats, frame_off, res

    sil = [world.lexicon.phone_table.find("sil"), world.lexicon.phone_table.find("spn")]
    identity = np.stack([np.concatenate([np.eye(40), np.zeros((40, 1))], axis=1)] * n_spk).astype(np.float32)

    def two_pass(pre):
        """pass 1 on features pre-transformed by `pre` → statistics → solve → pass 2 with (estimate ∘ pre)."""
        feats1, frame_off, res1 = align_with(pre)
        st1 = res1["status"].cpu().numpy()
        assert np.all((st1 == 0) | (st1 == 1)), st1
        spk_ids, beta, K, Gm = fmllr_statistics(engine, feats1, frame_off, res1["ali"], model.tm, u2s, sil)
        assert spk_ids.tolist() == list(range(n_spk)) and np.all(beta > 1500)  # silence frames carry no weight
        solved = [F.compute_fmllr(beta[s], K[s], Gm[s]) for s in range(n_spk)]
        assert all(impr > 0.0 for _, impr in solved)  # the auxiliary function (it carries the log-det term) rises
        total = np.stack([_compose(solved[s][0], pre[s]) for s in range(n_spk)])
        feats2, _fo, res2 = align_with(total)
        st2 = res2["status"].cpu().numpy()
        assert np.all((st2 == 0) | (st2 == 1))
        # (the decoder's likelihood carries no Jacobian term, so it may move either way: the objective that must rise
        #  is the auxiliary function checked above)
        return feats1.cpu().numpy(), feats2.cpu().numpy()

    # fMLLR moves every speaker towards the speaker-independent model's space, so the adapted features must come out
    # (nearly) the same whether or not the input was distorted: adapt(D·x) ≈ adapt(x), although D·x is far from x.
    clean1, clean2 = two_pass(identity)
    dist1, dist2 = two_pass(distort)
    var = clean2.var(axis=0)
    d_in = (((dist1 - clean1) ** 2) / var).mean()
    d_out = (((dist2 - clean2) ** 2) / var).mean()
    assert d_out < 0.5 * d_in, (d_in, d_out)
