"""The device path against the committed golden vectors (tests/golden/oracle_vectors.npz): PCM → MFCC → CMVN → Δ /
splice+LDA → scores → alignment on the reference's plumbing fixture, without calling the oracle at all."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu
GOLD = helpers.REF.parent / "oracle_vectors.npz"


def test_device_pipeline_reproduces_golden_vectors(engine, fx):
    import torch

    gold = np.load(GOLD)
    pcm = fx.pcm[: int(gold["pcm_samples"])]
    so = np.array([0, len(pcm)], dtype=np.int64)
    d_pcm = torch.from_numpy(pcm.astype(np.int16)).to(engine.device)
    for snip in (1, 0):
        engine.configure_mfcc(snip_edges=snip)
        mfcc, frame_off = engine.mfcc(d_pcm, so)
        assert np.abs(mfcc.cpu().numpy() - gold[f"mfcc_snip{snip}"]).max() < 2e-3   # different FFT radix, int16-scale audio
    spk = np.zeros(1, np.int32)
    stats = engine.cmvn_stats(mfcc, frame_off, spk, 1)
    assert np.allclose(stats.cpu().numpy()[0], gold["cmvn_stats"], rtol=1e-6, atol=2e-2)
    x = engine.features(mfcc, frame_off, spk, stats)
    assert np.abs(x.cpu().numpy() - gold["delta_feats"]).max() < 2e-3
    y = engine.features(mfcc, frame_off, spk, stats, lda=torch.from_numpy(fx.g2p_lda.astype(np.float32)).to(engine.device))
    assert np.abs(y.cpu().numpy() - gold["lda_feats"]).max() < 5e-3
    # scores for all 132 pdfs on the golden features themselves (isolates the scoring kernel): within 1e-3 as north_star asks
    tm, am = fx.mono_tm, fx.mono_am
    engine.load_gmm(am)
    gx = torch.from_numpy(gold["delta_feats"][:50].copy()).to(engine.device)
    pl, cc = engine.sort_pdf_list(np.arange(am.num_pdfs, dtype=np.int32))
    ll, ll_off, _ = engine.score(gx, np.array([0, 50], np.int64), engine._dev(pl), np.array([0, len(pl)], np.int64),
                                 engine._dev(cc[None, :].astype(np.int32)))
    got = ll.cpu().numpy().reshape(50, len(pl))
    assert np.abs(got - gold["loglikes_first50_allpdfs"][:, pl]).max() < 1e-3
    # alignment of the golden graph on device features: transition-ids, words and phone boundaries frame-identical
    fst = fx.mono_graph(str(gold["text"]))
    graphs = engine.pack_graphs([fst], tm)
    ll, ll_off, ll_cols = engine.score(x, frame_off, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts,
                                       pdf_first_frame=graphs.pdf_first_frame)
    res = engine.align(graphs, ll, ll_off, ll_cols, frame_off, beam=100.0, retry_beam=400.0)
    assert int(res["status"][0]) == int(gold["status"])
    assert np.array_equal(res["ali"].cpu().numpy(), gold["ali"])
    nw = int(res["n_words"][0])
    assert np.array_equal(res["words"].cpu().numpy()[:nw], gold["words"])
    assert abs(float(res["like"][0]) - float(gold["like"])) / len(gold["ali"]) < 1e-3
    from montreal_forced_aligner_amd import ctm as C
    assert [list(r) for r in C.split_to_phones(res["ali"].cpu().numpy(), tm)] == gold["phone_intervals"].tolist()
    engine.configure_mfcc()
