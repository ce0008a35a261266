"""Second, independent restatement of the path in numpy/float64 (TEST INFRASTRUCTURE ONLY).

Used to cross-check oracle/mfa_oracle.cpp — the reference holds no golden vectors at the kalpy boundary
(SURVEY.md §8c), so two independently written implementations agreeing is the available evidence.
Written from the formulas in SURVEY.md Appendix A, vectorised, float64 throughout (so it approximates the
float32 oracle from the "exact arithmetic" side).  PARITY STATUS: parity unpinned.
"""
from __future__ import annotations

import numpy as np


def mfcc(wave, samp_freq=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0, preemph=0.97, low_freq=20.0,
         high_freq=7800.0, num_mel_bins=23, num_ceps=13, cepstral_lifter=22.0, snip_edges=False, remove_dc_offset=True):
    """SURVEY Appendix A.1 (Kaldi compute-mfcc-feats with MFA's options, dither 0, use_energy False)."""
    wave = np.asarray(wave, dtype=np.float64)
    n = wave.shape[0]
    win = int(samp_freq * 0.001 * frame_length_ms)
    shift = int(samp_freq * 0.001 * frame_shift_ms)
    nfft = 1
    while nfft < win:
        nfft *= 2
    if snip_edges:
        T = 0 if n < win else 1 + (n - win) // shift
        starts = np.arange(T) * shift
    else:
        T = (n + shift // 2) // shift
        starts = np.arange(T) * shift + shift // 2 - win // 2
    idx = starts[:, None] + np.arange(win)[None, :]
    for _ in range(4):  # reflect (repeatedly for pathological short inputs)
        idx = np.where(idx < 0, -idx - 1, idx)
        idx = np.where(idx >= n, 2 * n - 1 - idx, idx)
    fr = wave[idx]
    if remove_dc_offset:
        fr = fr - fr.mean(axis=1, keepdims=True)
    pre = fr.copy()
    pre[:, 1:] = fr[:, 1:] - preemph * fr[:, :-1]
    pre[:, 0] = fr[:, 0] - preemph * fr[:, 0]
    window = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win) / (win - 1))) ** 0.85
    spec = np.fft.rfft(pre * window, nfft, axis=1)
    power = spec.real ** 2 + spec.imag ** 2  # [T, nfft/2+1]

    def mel(f):
        return 1127.0 * np.log(1.0 + f / 700.0)

    nyq = 0.5 * samp_freq
    hf = high_freq if high_freq > 0 else nyq + high_freq
    ml, mh = mel(low_freq), mel(hf)
    delta = (mh - ml) / (num_mel_bins + 1)
    binf = mel(samp_freq / nfft * np.arange(nfft // 2))
    W = np.zeros((num_mel_bins, nfft // 2))
    for b in range(num_mel_bins):
        left, center, right = ml + b * delta, ml + (b + 1) * delta, ml + (b + 2) * delta
        up = (binf - left) / (center - left)
        down = (right - binf) / (right - center)
        w = np.where(binf <= center, up, down)
        W[b] = np.where((binf > left) & (binf < right), w, 0.0)
    melE = power[:, : nfft // 2] @ W.T
    melE = np.log(np.maximum(melE, np.finfo(np.float32).eps))
    k = np.arange(num_ceps)[:, None]
    nn = np.arange(num_mel_bins)[None, :]
    dct = np.sqrt(2.0 / num_mel_bins) * np.cos(np.pi / num_mel_bins * (nn + 0.5) * k)
    dct[0, :] = np.sqrt(1.0 / num_mel_bins)
    c = melE @ dct.T
    if cepstral_lifter != 0:
        c = c * (1.0 + 0.5 * cepstral_lifter * np.sin(np.pi * np.arange(num_ceps) / cepstral_lifter))
    return c


def cmvn(feats):
    return feats - feats.mean(axis=0, keepdims=True)


def deltas(feats):
    """Appendix A.3: Δ = [-2,-1,0,1,2]/10, ΔΔ = Δ∗Δ, edge frames clamped."""
    feats = np.asarray(feats, np.float64)
    T = feats.shape[0]
    d1 = np.array([-2, -1, 0, 1, 2], dtype=np.float64) / 10.0
    d2 = np.convolve(d1, d1)

    def conv(k):
        h = (len(k) - 1) // 2
        out = np.zeros_like(feats)
        for j in range(-h, h + 1):
            out += k[j + h] * feats[np.clip(np.arange(T) + j, 0, T - 1)]
        return out

    return np.concatenate([feats, conv(d1), conv(d2)], axis=1)


def splice(feats, left=3, right=3):
    T = feats.shape[0]
    return np.concatenate([feats[np.clip(np.arange(T) + j, 0, T - 1)] for j in range(-left, right + 1)], axis=1)


def affine(feats, M):
    d = feats.shape[1]
    y = feats @ M[:, :d].T
    if M.shape[1] == d + 1:
        y = y + M[:, d]
    return y


def gmm_loglikes(feats, gconsts, means_invvars, inv_vars, pdf_offsets, pdf_list):
    """Appendix A.6, float64, plain log-sum-exp (no cutoff: it only drops terms < eps relative)."""
    x = np.asarray(feats, np.float64)
    ll = gconsts[None, :].astype(np.float64) + x @ means_invvars.T.astype(np.float64) - 0.5 * (x * x) @ inv_vars.T.astype(np.float64)
    out = np.zeros((x.shape[0], len(pdf_list)))
    for j, p in enumerate(pdf_list):
        a, b = pdf_offsets[p], pdf_offsets[p + 1]
        m = ll[:, a:b].max(axis=1)
        out[:, j] = m + np.log(np.exp(ll[:, a:b] - m[:, None]).sum(axis=1))
    return out


def viterbi_exact(num_states, start, arc_offsets, arcs, final, neg_scaled_loglikes, tid2col):
    """Unpruned Viterbi over an epsilon-free graph: returns (best total cost, alignment).  Cost = Σ arc weight +
    Σ acoustic cost + final weight; used to check that the beam decoder with a wide beam finds the optimum."""
    T = neg_scaled_loglikes.shape[0]
    INF = np.inf
    cost = np.full(num_states, INF)
    cost[start] = 0.0
    src = np.repeat(np.arange(num_states), np.diff(arc_offsets))
    il = arcs["ilabel"]
    if np.any(il == 0):
        raise ValueError("viterbi_exact expects an epsilon-free graph")
    w = arcs["weight"].astype(np.float64)
    dst = arcs["nextstate"]
    cols = tid2col[il]
    back = np.zeros((T, num_states), dtype=np.int64)
    for t in range(T):
        cand = cost[src] + w + neg_scaled_loglikes[t, cols].astype(np.float64)
        new = np.full(num_states, INF)
        order = np.argsort(cand, kind="stable")
        # first occurrence per dst in cost order wins
        d_sorted = dst[order]
        first = np.unique(d_sorted, return_index=True)[1]
        win = order[first]
        new[dst[win]] = cand[win]
        back[t, dst[win]] = win
        cost = new
    tot = cost + np.where(np.isinf(final), INF, final.astype(np.float64))
    s = int(np.argmin(tot))
    best = tot[s]
    ali = np.zeros(T, dtype=np.int32)
    for t in range(T - 1, -1, -1):
        a = back[t, s]
        ali[t] = il[a]
        s = src[a]
    return best, ali
