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


def test_two_model_accumulation_and_transform_composition():
    """Two-model form (posteriors from the alignment model, statistics from the final model — MFA/corpus/features.py:503-511)
    against a float64 numpy restatement, and compose_transforms (previous_transform_archive, :482-512) as plain algebra."""
    from montreal_forced_aligner_amd import fmllr as F

    rng = np.random.default_rng(12)
    D, T = 5, 60
    sizes = [3, 1, 4]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)

    def model(seed):
        r = np.random.default_rng(seed)
        n = int(offs[-1])
        mean = r.normal(0, 2, size=(n, D)); var = r.uniform(0.5, 2, size=(n, D))
        w = np.concatenate([r.dirichlet(np.ones(s)) for s in sizes])
        gc = np.log(w) - 0.5 * (D * np.log(2 * np.pi) + np.log(var).sum(1) + (mean * mean / var).sum(1))
        return gc.astype(np.float32), (mean / var).astype(np.float32), (1 / var).astype(np.float32)

    gc_a, mi_a, iv_a = model(1)
    gc_f, mi_f, iv_f = model(2)
    x = rng.normal(0, 2, size=(T, D)).astype(np.float32)
    pdf = rng.integers(0, len(sizes), size=T).astype(np.int32)
    w = (rng.random(T) > 0.2).astype(np.float32)
    beta, K, G = O.fmllr_acc(x, pdf, w, gc_a, mi_a, iv_a, offs, stat_means_invvars=mi_f, stat_inv_vars=iv_f)
    rb, rK, rG = 0.0, np.zeros((D, D + 1)), np.zeros((D, D + 1, D + 1))
    for t in range(T):
        if w[t] == 0:
            continue
        a, b = offs[pdf[t]], offs[pdf[t] + 1]
        xd = x[t].astype(np.float64)
        ll = gc_a[a:b] + mi_a[a:b].astype(np.float64) @ xd - 0.5 * iv_a[a:b].astype(np.float64) @ (xd * xd)
        post = np.exp(ll - ll.max()); post /= post.sum()
        xi = np.append(xd, 1.0)
        rb += 1.0
        rK += np.outer(post @ mi_f[a:b], xi)
        bb = post @ iv_f[a:b]
        rG += bb[:, None, None] * np.outer(xi, xi)[None]
    assert abs(beta[0] - rb) < 1e-4 and np.allclose(K, rK, rtol=1e-4, atol=1e-3) and np.allclose(G, rG, rtol=1e-4, atol=1e-3)
    W1 = np.concatenate([np.eye(D) + 0.1 * rng.normal(size=(D, D)), rng.normal(size=(D, 1))], axis=1).astype(np.float32)
    W2 = np.concatenate([np.eye(D) + 0.1 * rng.normal(size=(D, D)), rng.normal(size=(D, 1))], axis=1).astype(np.float32)
    Wc = F.compose_transforms(W2, W1)
    v = rng.normal(size=D)
    once = W1[:, :D] @ v + W1[:, D]
    assert np.allclose(Wc[:, :D] @ v + Wc[:, D], W2[:, :D] @ once + W2[:, D], atol=1e-5)
