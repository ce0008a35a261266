"""Host-side model tables: transition model (derived ids) and the diagonal-GMM acoustic model in SoA form.

Mirrors what the reference obtains from kalpy — ``read_gmm_model`` / ``read_transition_model``
(MFA/alignment/base.py:339, MFA/models.py:481-491) and the model half of
``GmmAligner.__init__`` / ``.boost_silence`` (MFA/alignment/multiprocessing.py:814-815).  The derived-table rules
are Kaldi's ``TransitionModel::ComputeDerived``; the scaling rule is ``AddTransitionProbs``
(Appendix A.8).  These run once per model on the host; the device receives flat arrays.
"""
from __future__ import annotations

import ctypes
import ctypes.util
from dataclasses import dataclass
from typing import Iterable, List

import numpy as np

from . import kaldi_io

# Kaldi's Exp()/Log() on BaseFloat are the C library's expf/logf; numpy's float32 exp/log are separate SIMD
# implementations that differ in the last bit now and then, so bind libm directly for the model tables.
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.expf.restype = ctypes.c_float
_libm.expf.argtypes = [ctypes.c_float]
_libm.logf.restype = ctypes.c_float
_libm.logf.argtypes = [ctypes.c_float]


def _expf(x) -> np.float32:
    return np.float32(_libm.expf(float(x)))


def _logf(x) -> np.float32:
    return np.float32(_libm.logf(float(x)))


class TransitionModel:
    """Transition-id tables.  transition-ids and transition-states are 1-based as in Kaldi."""

    def __init__(self, raw: kaldi_io.RawTransitionModel):
        self.raw = raw
        self.topo = raw.topo
        self.tuples = raw.tuples
        self.log_probs = raw.log_probs.astype(np.float32)
        n = self.tuples.shape[0]
        state2id = np.zeros(n + 2, dtype=np.int32)
        cur = 1
        for ts in range(1, n + 2):
            state2id[ts] = cur
            if ts <= n:
                phone, hs = int(self.tuples[ts - 1, 0]), int(self.tuples[ts - 1, 1])
                cur += len(self.topo.entry_for_phone(phone)[hs].transitions)
        self.state2id = state2id
        self.num_transition_ids = cur - 1
        nt = cur
        self.id2state = np.zeros(nt, dtype=np.int32)
        self.id2pdf = np.full(nt, -1, dtype=np.int32)
        self.is_self_loop = np.zeros(nt, dtype=np.int32)
        self.is_final = np.zeros(nt, dtype=np.int32)
        self.id2phone = np.zeros(nt, dtype=np.int32)
        self.self_loop_of = np.zeros(n + 1, dtype=np.int32)  # per transition-state, 0 = none
        for ts in range(1, n + 1):
            phone, hs, fwd, slf = (int(x) for x in self.tuples[ts - 1])
            entry = self.topo.entry_for_phone(phone)
            for k, (dst, _p) in enumerate(entry[hs].transitions):
                tid = state2id[ts] + k
                self.id2state[tid] = ts
                self.id2phone[tid] = phone
                self.is_self_loop[tid] = int(dst == hs)
                self.is_final[tid] = int(len(entry[dst].transitions) == 0)
                self.id2pdf[tid] = slf if dst == hs else fwd
                if dst == hs:
                    self.self_loop_of[ts] = tid
        self.num_pdfs = int(max(self.tuples[:, 2].max(), self.tuples[:, 3].max())) + 1
        if self.log_probs.shape[0] != nt:
            raise kaldi_io.KaldiFormatError(
                f"final.mdl: {self.log_probs.shape[0]} log-probs for {nt - 1} transition-ids"
            )

    # Kaldi TransitionModel::GetNonSelfLoopLogProb
    def non_self_loop_log_prob(self, ts: int) -> np.float32:
        sl = int(self.self_loop_of[ts])
        if sl == 0:
            return np.float32(0.0)
        p = np.float32(1.0) - _expf(self.log_probs[sl])
        if p <= 0:
            p = np.float32(1.0e-10)
        return _logf(p)

    def scaled_log_probs(self, transition_scale: float, self_loop_scale: float) -> np.ndarray:
        """``GetScaledTransitionLogProb`` for every transition-id (index 0 = 0)."""
        ts_, sl_ = np.float32(transition_scale), np.float32(self_loop_scale)
        out = np.zeros(self.num_transition_ids + 1, dtype=np.float32)
        for tid in range(1, self.num_transition_ids + 1):
            lp = self.log_probs[tid]
            if ts_ == sl_:
                out[tid] = lp * ts_
            elif self.is_self_loop[tid]:
                out[tid] = sl_ * lp
            else:
                nsl = self.non_self_loop_log_prob(int(self.id2state[tid]))
                out[tid] = np.float32(sl_ * nsl) + np.float32(ts_ * np.float32(lp - nsl))
        return out

    def transition_id(self, ts: int, index: int) -> int:
        return int(self.state2id[ts]) + index

    def flat_topology(self):
        """Flattened topology arrays in the layout the oracle's orc_tm_derive takes (tests use this)."""
        phone2entry = self.topo.phone2idx.astype(np.int32)
        entry_off = [0]
        trans_off = [0]
        trans_dst: List[int] = []
        for e in self.topo.entries:
            for s in e:
                for dst, _ in s.transitions:
                    trans_dst.append(dst)
                trans_off.append(len(trans_dst))
            entry_off.append(len(trans_off) - 1)
        return (
            phone2entry,
            np.asarray(entry_off, dtype=np.int32),
            np.asarray(trans_off, dtype=np.int32),
            np.asarray(trans_dst, dtype=np.int32),
        )


@dataclass
class DiagGmmModel:
    """All Gaussians of an AmDiagGmm, concatenated (struct of arrays).

    gconsts [G], means_invvars [G, D], inv_vars [G, D], pdf_offsets [P+1] (Gaussian range per pdf).
    """

    dim: int
    gconsts: np.ndarray
    means_invvars: np.ndarray
    inv_vars: np.ndarray
    pdf_offsets: np.ndarray

    @property
    def num_pdfs(self) -> int:
        return int(self.pdf_offsets.shape[0] - 1)

    @property
    def num_gauss(self) -> int:
        return int(self.gconsts.shape[0])

    @classmethod
    def from_raw(cls, am: kaldi_io.RawAmDiagGmm) -> "DiagGmmModel":
        offs = np.zeros(am.num_pdfs + 1, dtype=np.int32)
        for i, g in enumerate(am.gconsts):
            offs[i + 1] = offs[i] + g.shape[0]
        return cls(
            am.dim,
            np.concatenate(am.gconsts).astype(np.float32),
            np.concatenate(am.means_invvars, axis=0).astype(np.float32),
            np.concatenate(am.inv_vars, axis=0).astype(np.float32),
            offs,
        )

    def boost_silence(self, factor: float, pdf_ids: Iterable[int]) -> None:
        """Scale the weights of the given pdfs by ``factor`` without renormalising ⇒ gconst += ln(factor).

        Reference: ``aligner.boost_silence(boost, silence_phone_ids)`` (MFA/alignment/multiprocessing.py:803-815);
        Kaldi ``AmDiagGmm`` weight scaling + ``ComputeGconsts``.
        """
        if factor == 1.0:
            return
        lf = _logf(np.float32(factor))
        for p in sorted(set(int(x) for x in pdf_ids)):
            a, b = int(self.pdf_offsets[p]), int(self.pdf_offsets[p + 1])
            self.gconsts[a:b] += lf


def pdfs_of_phones(tm: TransitionModel, phones: Iterable[int]) -> List[int]:
    """Kaldi ``GetPdfsForPhones``: every pdf reachable from the given phone ids."""
    ph = set(int(p) for p in phones)
    out = set()
    for row in tm.tuples:
        if int(row[0]) in ph:
            out.add(int(row[2]))
            out.add(int(row[3]))
    return sorted(out)


def load_model_bytes(data: bytes):
    raw_tm, raw_am = kaldi_io.read_model(data)
    return TransitionModel(raw_tm), DiagGmmModel.from_raw(raw_am)
