"""ctypes face of the CPU oracle (oracle/mfa_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never from the product package.
PARITY STATUS: parity unpinned (see mfa_oracle.cpp header and DESIGN.md).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

ARC_DTYPE = np.dtype([("ilabel", "<i4"), ("olabel", "<i4"), ("weight", "<f4"), ("nextstate", "<i4")])


class MfccOpts(C.Structure):
    _fields_ = [
        ("samp_freq", C.c_float), ("frame_length_ms", C.c_float), ("frame_shift_ms", C.c_float),
        ("preemph", C.c_float), ("low_freq", C.c_float), ("high_freq", C.c_float),
        ("cepstral_lifter", C.c_float), ("energy_floor", C.c_float),
        ("num_mel_bins", C.c_int32), ("num_ceps", C.c_int32), ("snip_edges", C.c_int32),
        ("remove_dc_offset", C.c_int32), ("use_energy", C.c_int32), ("raw_energy", C.c_int32),
    ]


def default_mfcc_opts(**kw) -> MfccOpts:
    d = dict(samp_freq=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0, preemph=0.97, low_freq=20.0,
             high_freq=7800.0, cepstral_lifter=22.0, energy_floor=0.0, num_mel_bins=23, num_ceps=13,
             snip_edges=0, remove_dc_offset=1, use_energy=0, raw_energy=1)
    d.update(kw)
    return MfccOpts(**d)


def build(force: bool = False) -> Path:
    so = _HERE / "libmfa_oracle.so"
    src = _HERE / "mfa_oracle.cpp"
    if force or not so.exists() or (src.exists() and so.stat().st_mtime < src.stat().st_mtime):
        subprocess.check_call(["make", "-C", str(_HERE), "-B", "libmfa_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = _HERE / "libmfa_oracle.so"
        if not so.exists():
            build()
        _LIB = C.CDLL(str(so))
        _LIB.orc_mfcc_num_frames.restype = C.c_int32
        _LIB.orc_mfcc_num_frames.argtypes = [C.c_int64, C.POINTER(MfccOpts)]
    return _LIB


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


def mfcc_num_frames(n: int, opts: MfccOpts) -> int:
    return int(lib().orc_mfcc_num_frames(C.c_int64(n), C.byref(opts)))


def mfcc(wave: np.ndarray, opts: MfccOpts) -> np.ndarray:
    """wave: float32 on the int16 scale."""
    wave = np.ascontiguousarray(wave, dtype=np.float32)
    T = mfcc_num_frames(wave.shape[0], opts)
    out = np.zeros((T, opts.num_ceps), dtype=np.float32)
    if T > 0:
        lib().orc_mfcc(_p(wave), C.c_int64(wave.shape[0]), C.byref(opts), _p(out))
    return out


def mfcc_tables(opts: MfccOpts):
    win = int(opts.samp_freq * 0.001 * opts.frame_length_ms)
    padded = 1
    while padded < win:
        padded <<= 1
    window = np.zeros(win, np.float32)
    mel = np.zeros((opts.num_mel_bins, padded // 2), np.float32)
    dct = np.zeros((opts.num_ceps, opts.num_mel_bins), np.float32)
    lifter = np.zeros(opts.num_ceps, np.float32)
    lib().orc_mfcc_tables(C.byref(opts), _p(window), _p(mel), _p(dct), _p(lifter))
    return window, mel, dct, lifter


def cmvn_stats(feats_list) -> np.ndarray:
    dim = feats_list[0].shape[1]
    stats = np.zeros((2, dim + 1), dtype=np.float64)
    for f in feats_list:
        f = np.ascontiguousarray(f, dtype=np.float32)
        lib().orc_cmvn_acc(_p(f), C.c_int32(f.shape[0]), C.c_int32(dim), _p(stats))
    return stats


def cmvn_apply(stats: np.ndarray, feats: np.ndarray) -> np.ndarray:
    out = np.ascontiguousarray(feats, dtype=np.float32).copy()
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    lib().orc_cmvn_apply(_p(stats), _p(out), C.c_int32(out.shape[0]), C.c_int32(out.shape[1]))
    return out


def delta_scales(order=2, window=2) -> np.ndarray:
    out = np.zeros((order + 1, 2 * order * window + 1), np.float32)
    lib().orc_delta_scales(C.c_int32(order), C.c_int32(window), _p(out))
    return out


def deltas(feats: np.ndarray, order=2, window=2) -> np.ndarray:
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    T, d = feats.shape
    out = np.zeros((T, (order + 1) * d), np.float32)
    lib().orc_deltas(_p(feats), C.c_int32(T), C.c_int32(d), C.c_int32(order), C.c_int32(window), _p(out))
    return out


def splice(feats: np.ndarray, left=3, right=3) -> np.ndarray:
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    T, d = feats.shape
    out = np.zeros((T, (left + right + 1) * d), np.float32)
    lib().orc_splice(_p(feats), C.c_int32(T), C.c_int32(d), C.c_int32(left), C.c_int32(right), _p(out))
    return out


def affine(feats: np.ndarray, M: np.ndarray) -> np.ndarray:
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    M = np.ascontiguousarray(M, dtype=np.float32)
    T, d = feats.shape
    assert M.shape[1] in (d, d + 1)
    out = np.zeros((T, M.shape[0]), np.float32)
    lib().orc_affine(_p(feats), C.c_int32(T), C.c_int32(d), _p(M), C.c_int32(M.shape[0]), C.c_int32(M.shape[1]), _p(out))
    return out


def gmm_loglikes(feats, gconsts, means_invvars, inv_vars, pdf_offsets, pdf_list) -> np.ndarray:
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    gconsts = np.ascontiguousarray(gconsts, dtype=np.float32)
    means_invvars = np.ascontiguousarray(means_invvars, dtype=np.float32)
    inv_vars = np.ascontiguousarray(inv_vars, dtype=np.float32)
    pdf_offsets = np.ascontiguousarray(pdf_offsets, dtype=np.int32)
    pdf_list = np.ascontiguousarray(pdf_list, dtype=np.int32)
    T, D = feats.shape
    out = np.zeros((T, pdf_list.shape[0]), np.float32)
    lib().orc_gmm_loglikes(_p(feats), C.c_int32(T), C.c_int32(D), _p(gconsts), _p(means_invvars), _p(inv_vars),
                           _p(pdf_offsets), _p(pdf_list), C.c_int32(pdf_list.shape[0]), _p(out))
    return out


def tm_derive(phone2entry, entry_off, trans_off, trans_dst, tuples):
    tuples = np.ascontiguousarray(tuples, dtype=np.int32)
    n = tuples.shape[0]
    cap = 1 + sum(1 for _ in range(1)) + int(trans_dst.shape[0]) * n + 8  # generous
    cap = min(cap, 1 << 22)
    state2id = np.zeros(n + 2, np.int32)
    id2state = np.zeros(cap, np.int32)
    id2pdf = np.zeros(cap, np.int32)
    isl = np.zeros(cap, np.int32)
    isf = np.zeros(cap, np.int32)
    nt = lib().orc_tm_derive(_p(np.ascontiguousarray(phone2entry, np.int32)), _p(np.ascontiguousarray(entry_off, np.int32)),
                             _p(np.ascontiguousarray(trans_off, np.int32)), _p(np.ascontiguousarray(trans_dst, np.int32)),
                             _p(tuples), C.c_int32(n), _p(state2id), _p(id2state), _p(id2pdf), _p(isl), _p(isf), C.c_int32(cap))
    assert nt > 0
    return state2id, id2state[: nt + 1], id2pdf[: nt + 1], isl[: nt + 1], isf[: nt + 1]


def tm_scaled_logprobs(state2id, id2state, is_self_loop, log_probs, transition_scale, self_loop_scale) -> np.ndarray:
    n_ids = id2state.shape[0] - 1
    out = np.zeros(n_ids + 1, np.float32)
    lib().orc_tm_scaled_logprobs(_p(np.ascontiguousarray(state2id, np.int32)), _p(np.ascontiguousarray(id2state, np.int32)),
                                 _p(np.ascontiguousarray(is_self_loop, np.int32)),
                                 _p(np.ascontiguousarray(log_probs, np.float32)), C.c_int32(n_ids),
                                 C.c_float(transition_scale), C.c_float(self_loop_scale), _p(out))
    return out


def add_transition_probs(arcs: np.ndarray, scaled: np.ndarray) -> np.ndarray:
    arcs = np.ascontiguousarray(arcs.astype(ARC_DTYPE)).copy()
    scaled = np.ascontiguousarray(scaled, np.float32)
    rc = lib().orc_add_transition_probs(_p(arcs), C.c_int64(arcs.shape[0]), _p(scaled), C.c_int32(scaled.shape[0] - 1))
    if rc != 0:
        raise ValueError("AddTransitionProbs: invalid symbol on graph input side")
    return arcs


def align(num_states, start, arc_offsets, arcs, final, loglikes, tid2col, acoustic_scale, beam, retry_beam,
          want_stats=False):
    """Returns dict(status, ali, words, like, per_frame).  arcs must already carry transition probs."""
    arc_offsets = np.ascontiguousarray(arc_offsets, np.int64)
    arcs = np.ascontiguousarray(arcs.astype(ARC_DTYPE))
    final = np.ascontiguousarray(final, np.float32)
    loglikes = np.ascontiguousarray(loglikes, np.float32)
    tid2col = np.ascontiguousarray(tid2col, np.int32)
    T, ncols = loglikes.shape
    ali = np.zeros(T, np.int32)
    cap_words = T + 16
    words = np.zeros(cap_words, np.int32)
    n_words = C.c_int32(0)
    like = C.c_float(0)
    pf = np.zeros(T, np.float32)
    stats = np.zeros(2, np.int64)
    lib().orc_align.restype = C.c_int32
    st = lib().orc_align(C.c_int32(num_states), C.c_int32(start), _p(arc_offsets), _p(arcs), _p(final), _p(loglikes),
                         C.c_int32(T), C.c_int32(ncols), _p(tid2col), C.c_float(acoustic_scale), C.c_float(beam),
                         C.c_float(retry_beam), _p(ali), _p(words), C.c_int32(cap_words), C.byref(n_words), C.byref(like),
                         _p(pf), _p(stats))
    out = dict(status=int(st), ali=ali, words=words[: n_words.value].copy(), like=float(like.value), per_frame=pf)
    if want_stats:
        out["max_toks"] = int(stats[0])
        out["sum_toks"] = int(stats[1])
    return out


def align_feats(num_states, start, arc_offsets, arcs, final, feats, gconsts, means_invvars, inv_vars, pdf_offsets, tid2pdf,
                acoustic_scale, beam, retry_beam, state_depth=None):
    """The alignment with Kaldi's LAZY decodable (orc_align_feats): features + acoustic model in; a (frame, pdf) score is
    computed when a live token's arc first asks for it.  Same results as gmm_loglikes + align; also returns ``cells`` = the
    number of (frame, pdf) scores evaluated.  ``tid2pdf``: pdf id per transition-id (entry 0 unused)."""
    arc_offsets = np.ascontiguousarray(arc_offsets, np.int64)
    arcs = np.ascontiguousarray(arcs.astype(ARC_DTYPE))
    final = np.ascontiguousarray(final, np.float32)
    feats = np.ascontiguousarray(feats, np.float32)
    gconsts = np.ascontiguousarray(gconsts, np.float32)
    means_invvars = np.ascontiguousarray(means_invvars, np.float32)
    inv_vars = np.ascontiguousarray(inv_vars, np.float32)
    pdf_offsets = np.ascontiguousarray(pdf_offsets, np.int32)
    tid2pdf = np.ascontiguousarray(np.maximum(tid2pdf, 0), np.int32)
    T, D = feats.shape
    ali = np.zeros(T, np.int32)
    cap_words = T + 16
    words = np.zeros(cap_words, np.int32)
    n_words = C.c_int32(0)
    like = C.c_float(0)
    pf = np.zeros(T, np.float32)
    stats = np.zeros(3, np.int64)
    fn = lib().orc_align_feats
    fn.restype = C.c_int32
    sd = None if state_depth is None else np.ascontiguousarray(state_depth, np.int32)
    band = None if sd is None else np.zeros((T, 4), np.int32)
    st = fn(C.c_int32(num_states), C.c_int32(start), _p(arc_offsets), _p(arcs), _p(final), _p(feats), C.c_int32(T), C.c_int32(D),
            _p(gconsts), _p(means_invvars), _p(inv_vars), _p(pdf_offsets), C.c_int32(pdf_offsets.shape[0] - 1), _p(tid2pdf),
            C.c_float(acoustic_scale), C.c_float(beam), C.c_float(retry_beam), _p(ali), _p(words), C.c_int32(cap_words),
            C.byref(n_words), C.byref(like), _p(pf), _p(stats), None if sd is None else _p(sd), None if band is None else _p(band))
    out = dict(status=int(st), ali=ali, words=words[: n_words.value].copy(), like=float(like.value), per_frame=pf,
               max_toks=int(stats[0]), sum_toks=int(stats[1]), cells=int(stats[2]))
    if band is not None:
        out["frame_band"] = band      # per frame: min longest-reach depth, max BFS depth, tokens, BFS depth of the best token
    return out


def split_to_phones(ali, id2state, is_self_loop, is_final, tuples):
    ali = np.ascontiguousarray(ali, np.int32)
    T = ali.shape[0]
    out = np.zeros((T + 1, 3), np.int32)
    ok = C.c_int32(0)
    n = lib().orc_split_to_phones(_p(ali), C.c_int32(T), _p(np.ascontiguousarray(id2state, np.int32)),
                                  _p(np.ascontiguousarray(is_self_loop, np.int32)), _p(np.ascontiguousarray(is_final, np.int32)),
                                  _p(np.ascontiguousarray(tuples, np.int32)), _p(out), C.c_int32(T + 1), C.byref(ok))
    return out[:n].copy(), bool(ok.value)


def fmllr_acc(feats, ali_pdf, weight, gconsts, means_invvars, inv_vars, pdf_offsets, stats=None, stat_means_invvars=None,
              stat_inv_vars=None):
    """Accumulate fMLLR statistics of one utterance; stats = (beta[1], K[D,D+1], G[D,D+1,D+1]) float64, created if None.
    With ``stat_means_invvars`` / ``stat_inv_vars`` the posteriors come from (gconsts, means_invvars, inv_vars) — the
    alignment model — and the statistics from the stat_* arrays — the final model (MFA/corpus/features.py:503-511)."""
    feats = np.ascontiguousarray(feats, np.float32)
    T, D = feats.shape
    if stats is None:
        stats = (np.zeros(1), np.zeros((D, D + 1)), np.zeros((D, D + 1, D + 1)))
    beta, K, G = stats
    smi = None if stat_means_invvars is None else np.ascontiguousarray(stat_means_invvars, np.float32)
    siv = None if stat_inv_vars is None else np.ascontiguousarray(stat_inv_vars, np.float32)
    lib().orc_fmllr_acc2(_p(feats), C.c_int32(T), C.c_int32(D), _p(np.ascontiguousarray(ali_pdf, np.int32)),
                         _p(np.ascontiguousarray(weight, np.float32)), _p(np.ascontiguousarray(gconsts, np.float32)),
                         _p(np.ascontiguousarray(means_invvars, np.float32)), _p(np.ascontiguousarray(inv_vars, np.float32)),
                         None if smi is None else _p(smi), None if siv is None else _p(siv),
                         _p(np.ascontiguousarray(pdf_offsets, np.int32)), _p(beta), _p(K), _p(G))
    return stats


def fmllr_solve(beta, K, G, num_iters=40, min_count=500.0):
    D = K.shape[0]
    out = np.zeros((D, D + 1), np.float32)
    lib().orc_fmllr_solve.restype = C.c_double
    impr = lib().orc_fmllr_solve(C.c_int32(D), C.c_double(float(beta)), _p(np.ascontiguousarray(K, np.float64)),
                                 _p(np.ascontiguousarray(G, np.float64)), C.c_int32(num_iters), C.c_double(min_count), _p(out))
    return out, float(impr)
