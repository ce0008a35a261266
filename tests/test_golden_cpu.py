"""The committed golden vectors (tests/golden/oracle_vectors.npz, made by tests/golden/make_golden.py from the reference's
own plumbing fixtures) against the oracle and against the independent numpy restatement.  Integer results are exact;
float32 results are allowed the last-place noise of a different libm (1e-5 relative), nothing more."""
import numpy as np
import pytest

from oracle import np_oracle as NP
from oracle import oracle as O
from tests import helpers

GOLD = helpers.REF.parent / "oracle_vectors.npz"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _close(a, b, rtol=1e-5, atol=1e-5):
    assert a.shape == b.shape
    assert np.allclose(a, b, rtol=rtol, atol=atol), float(np.abs(a - b).max())


def test_oracle_reproduces_golden_features(fx, gold):
    pcm = fx.pcm[: int(gold["pcm_samples"])]
    wave = pcm.astype(np.float32)
    for snip in (0, 1):
        _close(O.mfcc(wave, O.default_mfcc_opts(snip_edges=snip)), gold[f"mfcc_snip{snip}"], atol=2e-4)
    mf = gold["mfcc_snip0"]
    stats = O.cmvn_stats([mf])
    assert np.allclose(stats, gold["cmvn_stats"], rtol=1e-12, atol=1e-9)
    x = O.deltas(O.cmvn_apply(stats, mf))
    _close(x, gold["delta_feats"])
    _close(O.affine(O.splice(O.cmvn_apply(stats, mf)), fx.g2p_lda), gold["lda_feats"], atol=1e-4)
    # the independent numpy restatement lands on the same vectors (float64 internally → looser on the FFT stage)
    assert np.abs(NP.mfcc(wave) - gold["mfcc_snip0"]).max() < 2e-3
    assert np.abs(NP.deltas(NP.cmvn(mf)) - gold["delta_feats"]).max() < 1e-4


def test_oracle_reproduces_golden_scores_graph_and_alignment(fx, gold):
    tm, am = fx.mono_tm, fx.mono_am
    x = gold["delta_feats"]
    ll = O.gmm_loglikes(x[:50], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, np.arange(am.num_pdfs))
    _close(ll, gold["loglikes_first50_allpdfs"], atol=1e-4)
    fst = fx.mono_graph(str(gold["text"]))
    assert fst.start == int(gold["graph_start"])
    assert np.array_equal(fst.arc_offsets, gold["graph_arc_offsets"])
    for k in ("ilabel", "olabel", "nextstate"):
        assert np.array_equal(fst.arcs[k], gold[f"graph_{k}"]), k
    _close(fst.arcs["weight"], gold["graph_weight"], atol=1e-6)
    pl = np.unique(tm.id2pdf[fst.arcs["ilabel"]])
    lls = O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl)
    res = helpers.oracle_align(tm, fst, lls, pl, beam=100.0, retry_beam=400.0)
    assert res["status"] == int(gold["status"])
    assert np.array_equal(res["ali"], gold["ali"]) and np.array_equal(res["words"], gold["words"])
    assert abs(res["like"] - float(gold["like"])) < 1e-2
    ph, ok = O.split_to_phones(res["ali"], tm.id2state, tm.is_self_loop, tm.is_final, tm.tuples)
    assert ok and np.array_equal(ph, gold["phone_intervals"])
    # the product's host code splits the same alignment into the same phones
    from montreal_forced_aligner_amd import ctm as C
    assert [(a, n, p) for a, n, p in C.split_to_phones(gold["ali"], tm)] == [tuple(r) for r in gold["phone_intervals"].tolist()]


def test_oracle_against_reference_output_when_captured(fx):
    """tools/capture_kalpy_golden.py (run by hand where kalpy is installed) writes tests/golden/kalpy_vectors.npz: the same
    inputs through the reference's own dependency.  Absent here (kalpy cannot be installed: parity unpinned) → skipped;
    present → the oracle must reproduce the reference's numbers (north_star bars: alignment identical, log-likelihood
    within 1e-3 per frame; features / scores to float32 FFT and summation-order noise)."""
    import os

    path = os.environ.get("MFA_KALPY_GOLDEN", str(helpers.REF.parent / "kalpy_vectors.npz"))
    if not os.path.exists(path):
        pytest.skip("no reference capture (tools/capture_kalpy_golden.py has not been run where kalpy exists)")
    ref = np.load(path)
    wave = fx.pcm[: int(ref["pcm_samples"])].astype(np.float32)
    for snip in (0, 1):
        assert np.abs(O.mfcc(wave, O.default_mfcc_opts(snip_edges=snip)) - ref[f"mfcc_snip{snip}"]).max() < 2e-3
    mf = ref["mfcc_snip0"]
    stats = O.cmvn_stats([mf])
    assert np.allclose(stats, ref["cmvn_stats"], rtol=1e-9, atol=1e-6)
    x = O.deltas(O.cmvn_apply(stats, mf))
    assert np.abs(x - ref["delta_feats"]).max() < 1e-4
    assert np.abs(O.affine(O.splice(O.cmvn_apply(stats, mf)), fx.g2p_lda) - ref["lda_feats"]).max() < 1e-3
    tm, am = fx.mono_tm, fx.mono_am
    ll = O.gmm_loglikes(ref["delta_feats"][:50], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, np.arange(am.num_pdfs))
    assert np.abs(ll - ref["loglikes_first50_allpdfs"]).max() < 1e-3
    fst = fx.mono_graph(str(ref["text"]))
    pl = np.unique(tm.id2pdf[fst.arcs["ilabel"]])
    lls = O.gmm_loglikes(ref["delta_feats"], am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl)
    res = helpers.oracle_align(tm, fst, lls, pl, beam=100.0, retry_beam=400.0)
    assert np.array_equal(res["ali"], ref["ali"]) and np.array_equal(res["words"], ref["words"])
    assert abs(res["like"] - float(ref["like"])) / len(ref["ali"]) < 1e-3
