"""Per-speaker fMLLR estimation between the two alignment passes (SURVEY "next" row N3).

Reference: ``CalcFmllrFunction`` (MFA/corpus/features.py:460-548; options :759-766 — update type "full", silence weight
0.0) driven by ``calc_fmllr`` (MFA/corpus/acoustic_corpus.py:1370-1419) from ``CorpusAligner.align``
(MFA/alignment/base.py:510-539).  Kaldi: transform/fmllr-diag-gmm.cc (FmllrDiagGmmAccs, ComputeFmllrMatrixDiagGmmFull).

Split of work: the per-frame statistics (Gaussian posteriors of the aligned pdf, β, K = Σ a ξᵀ, G_d = Σ b_d ξ ξᵀ) are
accumulated per speaker on the GPU (``mfa_fmllr_acc_batch``); the tiny per-speaker solve (D rows × 40 sweeps of
(D+1)×(D+1) systems) runs here on the host in float64, exactly following Kaldi's row-by-row update.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def fmllr_aux(W: np.ndarray, beta: float, K: np.ndarray, G: np.ndarray) -> float:
    """Kaldi FmllrAuxFuncDiagGmm: β log|det A| + tr(W Kᵀ) − ½ Σ_d w_d G_d w_dᵀ."""
    D = W.shape[0]
    sign, logdet = np.linalg.slogdet(W[:, :D])
    obj = beta * logdet + float((W * K).sum())
    obj -= 0.5 * float(np.einsum("de,def,df->", W, G, W))
    return obj


def compute_fmllr(beta: float, K: np.ndarray, G: np.ndarray, num_iters: int = 40, min_count: float = 500.0,
                  init: Optional[np.ndarray] = None) -> Tuple[np.ndarray, float]:
    """ComputeFmllrMatrixDiagGmmFull.  K: [D, D+1], G: [D, D+1, D+1] (float64).  Returns (W [D, D+1] float32,
    auxiliary-function improvement); identity and 0.0 when β < min_count or the objective would not increase."""
    D = K.shape[0]
    W0 = np.concatenate([np.eye(D), np.zeros((D, 1))], axis=1) if init is None else np.asarray(init, dtype=np.float64)
    if beta < min_count:
        return W0.astype(np.float32), 0.0
    K = np.asarray(K, dtype=np.float64)
    G = np.asarray(G, dtype=np.float64)
    inv_G = np.linalg.inv(G)
    W = W0.copy()
    old = fmllr_aux(W0, beta, K, G)
    for _ in range(num_iters):
        for d in range(D):
            cof = np.zeros(D + 1)
            cof[:D] = np.linalg.inv(W[:, :D].T)[d]          # row d of the cofactor matrix (up to the determinant)
            cg = inv_G[d] @ cof
            e1 = float(cg @ cof)
            e2 = float(cg @ K[d])
            discr = np.sqrt(e2 * e2 + 4.0 * e1 * beta)
            a1, a2 = (-e2 + discr) / (2 * e1), (-e2 - discr) / (2 * e1)
            f1 = beta * np.log(abs(a1 * e1 + e2)) - 0.5 * a1 * a1 * e1
            f2 = beta * np.log(abs(a2 * e1 + e2)) - 0.5 * a2 * a2 * e1
            alpha = a1 if f1 > f2 else a2
            W[d] = inv_G[d] @ (alpha * cof + K[d])
    new = fmllr_aux(W, beta, K, G)
    impr = new - old
    if impr < 0.0 and not abs(new - old) <= 0.001 * (abs(new) + abs(old)):
        return W0.astype(np.float32), 0.0
    return W.astype(np.float32), float(impr)


def compose_transforms(new: np.ndarray, previous: np.ndarray) -> np.ndarray:
    """Affine composition ``new ∘ previous`` of two fMLLR matrices [D, D+1] (Kaldi ComposeTransforms with b_is_affine):
    features that already carry ``previous`` were used to estimate ``new``; the product applies to the untransformed ones
    (the ``previous_transform_archive`` branch of CalcFmllrFunction, MFA/corpus/features.py:482-512)."""
    new = np.asarray(new, dtype=np.float64)
    prev = np.asarray(previous, dtype=np.float64)
    D = new.shape[0]
    A = new[:, :D] @ prev[:, :D]
    b = new[:, :D] @ prev[:, D] + new[:, D]
    return np.concatenate([A, b[:, None]], axis=1).astype(np.float32)
