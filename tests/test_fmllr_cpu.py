"""fMLLR estimation: oracle accumulation vs an independent numpy accumulation, host solver vs oracle solver,
and the defining property — features distorted by a known affine map are pulled back towards the model."""
import numpy as np

from montreal_forced_aligner_amd import fmllr as F
from oracle import oracle as O
from tests import helpers


def _np_acc(feats, ali_pdf, weight, am):
    D = feats.shape[1]
    beta, K, G = 0.0, np.zeros((D, D + 1)), np.zeros((D, D + 1, D + 1))
    for t in range(feats.shape[0]):
        if weight[t] == 0:
            continue
        a0, a1 = am.pdf_offsets[ali_pdf[t]], am.pdf_offsets[ali_pdf[t] + 1]
        x = feats[t].astype(np.float64)
        ll = am.gconsts[a0:a1] + am.means_invvars[a0:a1] @ x - 0.5 * am.inv_vars[a0:a1] @ (x * x)
        post = np.exp(ll - ll.max())
        post = post / post.sum() * weight[t]
        xi = np.append(x, 1.0)
        beta += post.sum()
        K += np.outer(post @ am.means_invvars[a0:a1], xi)
        G += (post @ am.inv_vars[a0:a1])[:, None, None] * np.outer(xi, xi)[None]
    return beta, K, G


def _data(rng, am, n_frames, W_true=None):
    D = am.dim
    pdfs = rng.integers(0, am.num_pdfs, size=n_frames).astype(np.int32)
    feats = np.zeros((n_frames, D), np.float32)
    for t, p in enumerate(pdfs):
        g = rng.integers(am.pdf_offsets[p], am.pdf_offsets[p + 1])
        var = 1.0 / am.inv_vars[g]
        feats[t] = am.means_invvars[g] * var + np.sqrt(var) * rng.normal(size=D)
    if W_true is not None:  # speaker distortion: x_spk = A^-1 (x - b)  ⇒  the estimate should recover (A, b)
        A, b = W_true[:, :D], W_true[:, D]
        feats = ((feats - b) @ np.linalg.inv(A).T).astype(np.float32)
    return feats, pdfs


def test_accumulation_oracle_vs_numpy():
    rng = np.random.default_rng(0)
    am = helpers.random_gmm(rng, 12, [1, 3, 8, 5, 32, 2])
    feats, pdfs = _data(rng, am, 300)
    w = (rng.random(300) > 0.2).astype(np.float32)
    beta, K, G = O.fmllr_acc(feats, pdfs, w, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets)
    b2, K2, G2 = _np_acc(feats, pdfs, w, am)
    assert abs(beta[0] - b2) < 1e-3 and abs(beta[0] - w.sum()) < 1e-3
    assert np.allclose(K, K2, rtol=1e-4, atol=1e-3) and np.allclose(G, G2, rtol=1e-4, atol=1e-3)
    assert np.allclose(G, np.transpose(G, (0, 2, 1)))


def test_solver_host_vs_oracle_and_recovers_known_transform():
    rng = np.random.default_rng(1)
    D = 10
    am = helpers.random_gmm(rng, D, [4] * 30)
    W_true = np.concatenate([np.eye(D) + 0.08 * rng.normal(size=(D, D)), 0.5 * rng.normal(size=(D, 1))], axis=1)
    feats, pdfs = _data(rng, am, 4000, W_true)
    stats = O.fmllr_acc(feats, pdfs, np.ones(4000, np.float32), am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets)
    beta, K, G = stats[0][0], stats[1], stats[2]
    W_o, impr_o = O.fmllr_solve(beta, K, G)
    W_h, impr_h = F.compute_fmllr(beta, K, G)
    assert impr_o > 0 and abs(impr_o - impr_h) < 1e-6 * abs(impr_o) + 1e-6
    assert np.abs(W_o - W_h).max() < 1e-4
    assert np.abs(W_h - W_true).max() < 0.15  # sampling noise of 4000 frames
    # below min_count: identity, no improvement
    W_i, impr_i = F.compute_fmllr(100.0, K, G)
    assert impr_i == 0.0 and np.array_equal(W_i[:, :D], np.eye(D, dtype=np.float32))
    assert O.fmllr_solve(100.0, K, G)[1] == 0.0
