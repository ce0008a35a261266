"""Batched device pipeline: PCM → MFCC → CMVN → Δ/LDA/fMLLR → GMM scores → Viterbi alignment.

Host orchestration over the C ABI (include/mfa_hip.h).  torch is used only for device memory and streams
("plumbing, not the product"): every kernel runs inside libmfa_hip.so.  The stages mirror the reference's job
functions — MfccFunction / calc_cmvn / FinalFeatureFunction / AlignFunction
(MFA/corpus/features.py:193-376, MFA/corpus/acoustic_corpus.py:1315-1367, MFA/alignment/multiprocessing.py:791-863) —
but operate on ragged batches resident in HBM instead of ark/scp files between processes.
"""
from __future__ import annotations

import os
import ctypes as C
from dataclasses import dataclass
from typing import Tuple, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import AlignOpts, GraphBatch, MfccOpts, ScorePlan, check
from .kaldi_io import Fst
from .model import DiagGmmModel, TransitionModel

STATUS_OK, STATUS_RETRIED, STATUS_FAILED, STATUS_TOKEN_OVERFLOW, STATUS_BP_OVERFLOW, STATUS_UNSUPPORTED = 0, 1, 2, 3, 4, 5

DEFAULT_MFCC = dict(sample_frequency=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0, preemphasis=0.97,
                    low_frequency=20.0, high_frequency=7800.0, cepstral_lifter=22.0, energy_floor=0.0,
                    num_mel_bins=23, num_coefficients=13, snip_edges=0, remove_dc_offset=1, use_energy=0, raw_energy=1)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _host_threads(share: float = 1.0, cap: int = 32) -> int:
    from . import hostcpu
    return hostcpu.threads(share, cap)


class RaggedView:
    """``np.split(flat, offsets[1:-1])`` without making the pieces up front: piece u is ``flat[offsets[u]: offsets[u + 1]]``."""

    def __init__(self, flat: np.ndarray, offsets: np.ndarray):
        self.flat, self.offsets = flat, offsets

    def __len__(self) -> int:
        return len(self.offsets) - 1

    def __getitem__(self, u):
        if isinstance(u, slice):
            return [self[k] for k in range(*u.indices(len(self)))]
        if u < 0:
            u += len(self)
        return self.flat[int(self.offsets[u]): int(self.offsets[u + 1])]

    def __iter__(self):
        for u in range(len(self)):
            yield self[u]


class StagingPool:
    """Named host staging buffers in pinned memory, reused from batch to batch (a fresh 200 MB numpy array costs more in page
    faults than the copy it is for).  ``get`` hands out a numpy view; ``to_device`` starts an asynchronous H2D copy of a view
    and remembers it: ``wait`` blocks until every copy started since the last ``wait`` has left the buffers — call it before
    writing into them again."""

    def __init__(self, device: torch.device, pinned: bool = True):
        self.device = device
        self.pinned = pinned
        self._buf: Dict[str, torch.Tensor] = {}
        self._event: Optional[torch.cuda.Event] = None

    def get(self, name: str, count: int, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        need = max(1, int(count)) * dtype.itemsize
        t = self._buf.get(name)
        if t is None or t.numel() < need:
            size = int(need * 1.25) + 64
            t = None
            if self.pinned:
                try:
                    t = torch.empty(size, dtype=torch.uint8, pin_memory=True)
                except RuntimeError:      # pinned allocation refused (constrained container): pageable still works
                    self.pinned = False
            if t is None:
                t = torch.empty(size, dtype=torch.uint8)
            self._buf[name] = t
        return t.numpy()[: int(count) * dtype.itemsize].view(dtype)

    def to_device(self, view: np.ndarray, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        src = torch.from_numpy(view)
        with torch.cuda.device(self.device):
            out = src.to(self.device, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record(torch.cuda.current_stream(self.device) if stream is None else stream)
        return out

    def wait(self) -> None:
        if self._event is not None:
            self._event.synchronize()
            self._event = None


@dataclass
class PackedGraphs:
    """Graphs of a batch in the device layout mfa_align_batch consumes (see mfa_graph_batch)."""

    n_utt: int
    max_states: int
    max_arcs: int
    total_arcs: int
    tensors: Dict[str, torch.Tensor]
    pdf_list: torch.Tensor      # int32 [ΣP_u] (slot-sorted per utterance)
    pdf_off: torch.Tensor       # int64 [n_utt+1]
    class_counts: torch.Tensor  # int32 [n_utt,6]
    pdf_off_host: np.ndarray
    pdf_lists_host: Sequence[np.ndarray]                  # per-utterance views of the host copy of pdf_list
    pdf_first_frame: Optional[torch.Tensor] = None        # int32 [ΣP_u]: first frame a pdf can be asked for
    pdf_first_frame_host: Optional[List[np.ndarray]] = None
    # lazy (windowed) scoring keys — mfa_score_plan: running max inside each class of the longest-path depth of a pdf's
    # source states, and {BFS depth, longest-path depth} of every graph state
    pdf_last_depth: Optional[torch.Tensor] = None         # int32 [ΣP_u]
    state_depth: Optional[torch.Tensor] = None            # int32 [ΣS_u, 2]
    # class-0 columns laid out in `groups` runs (pdf id mod groups), one per XCD of the scoring kernel: columns per run
    groups: int = 1
    group_counts: Optional[torch.Tensor] = None           # int32 [n_utt, groups]

    def plan(self) -> ScorePlan:
        if self.pdf_last_depth is None or self.state_depth is None or self.pdf_first_frame is None:
            raise _lib.MfaHipError("these graphs were packed without depth keys (lazy scoring needs them)")
        return ScorePlan(self.pdf_list.data_ptr(), self.pdf_off.data_ptr(), self.class_counts.data_ptr(),
                         self.pdf_first_frame.data_ptr(), self.pdf_last_depth.data_ptr(), self.state_depth.data_ptr(),
                         int(np.diff(self.pdf_off_host).max()) if self.n_utt else 0, int(self.groups),
                         self.group_counts.data_ptr() if self.group_counts is not None else None)

    @property
    def has_eps(self) -> bool:
        """Some graph of the batch has epsilon input arcs (the decoder then runs ProcessNonemitting after every frame)."""
        return self.tensors.get("state_nemit") is not None

    def hard_bounds(self) -> Tuple[int, int]:
        """(max_tokens, bp_tokens_per_frame) that the decoder's capacity statuses cannot occur with: one token per graph state
        — and, for graphs with epsilon arcs, room behind the back-pointer trail for the path the traceback writes there
        (T + E arc indices, E ≤ T·S in the worst case: half a record per frame and state)."""
        s = max(1, int(self.max_states))
        return s, s + (s // 2 + 2 if self.has_eps else 0)

    def struct(self) -> GraphBatch:
        t = self.tensors
        nemit = t.get("state_nemit")
        return GraphBatch(self.n_utt, t["state_off"].data_ptr(), t["arc_base"].data_ptr(), t["start"].data_ptr(),
                          t["arc_off"].data_ptr(), t["final"].data_ptr(), t["arc_next"].data_ptr(),
                          t["arc_weight"].data_ptr(), t["arc_col"].data_ptr(), t["arc_ilabel"].data_ptr(),
                          t["arc_olabel"].data_ptr(), nemit.data_ptr() if nemit is not None else None)


class AlignmentEngine:
    """One engine per (process, GPU).  All tensors handed in must live on ``device``."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise _lib.MfaHipError("no GPU visible: the alignment engine has no CPU fallback")
        self.lib = _lib.lib()
        self.device = torch.device("cuda", device)
        self.ctx = self.lib.mfa_create(device)
        if not self.ctx:
            raise _lib.MfaHipError(f"mfa_create({device}) failed")
        self.ctx = C.c_void_p(self.ctx)
        self.use_torch_stream()
        self.mfcc_opts: Optional[MfccOpts] = None
        self.num_ceps = 13
        self.gmm: Optional[DiagGmmModel] = None
        self.slot_class: Optional[np.ndarray] = None
        # three sets of pinned staging buffers: the host fills one (the graphs of batch b + 2 compile into it) while the copies
        # out of the second are in flight and the third still backs the host-side graphs of the batch being collected
        self._staging = [StagingPool(self.device), StagingPool(self.device), StagingPool(self.device)]
        self._staging_turn = 0
        self._pcm_staging = [StagingPool(self.device), StagingPool(self.device)]   # PCM has its own pair: gathered while graphs compile
        self._pcm_turn = 0

    def next_staging(self) -> StagingPool:
        """The staging pool to fill next (round robin; waits until the copies last started from it are done)."""
        self._staging_turn = (self._staging_turn + 1) % len(self._staging)
        pool = self._staging[self._staging_turn]
        pool.wait()
        return pool

    def gather_pcm(self, arrays: Sequence[np.ndarray], pool: Optional[StagingPool] = None):
        """int16 arrays (one per utterance, host) → one device tensor + sample offsets: threaded gather into pinned staging
        memory (mfa_gather_pcm), one asynchronous H2D copy."""
        n = len(arrays)
        lens = np.fromiter((a.shape[0] for a in arrays), dtype=np.int64, count=n)
        so = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=so[1:])
        if pool is None:
            self._pcm_turn ^= 1
            pool = self._pcm_staging[self._pcm_turn]
            pool.wait()
        stage = pool.get("pcm", int(so[-1]), np.int16)
        ptrs = (C.c_void_p * max(n, 1))()
        keep = []
        for k, a in enumerate(arrays):
            if a.dtype != np.int16 or not a.flags.c_contiguous:
                a = np.ascontiguousarray(a, dtype=np.int16)
                keep.append(a)
            ptrs[k] = a.__array_interface__["data"][0]
        if self.lib.mfa_gather_pcm(n, ptrs, so.ctypes.data, stage.ctypes.data, _host_threads(0.5, 8)) != 0:
            raise _lib.MfaHipError("mfa_gather_pcm: bad arguments")
        return pool.to_device(stage, self.stream), so

    def close(self) -> None:
        if self.ctx:
            self.lib.mfa_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_torch_stream(self) -> None:
        self.stream = torch.cuda.current_stream(self.device)
        check(self.ctx, self.lib.mfa_set_stream(self.ctx, C.c_void_p(self.stream.cuda_stream), 0), "mfa_set_stream")

    # ------------------------------------------------------------------ configuration
    def configure_mfcc(self, **kw) -> None:
        d = dict(DEFAULT_MFCC)
        for k, v in kw.items():
            if k not in d:
                raise KeyError(f"unknown MFCC option {k!r}")
            d[k] = v
        for k in ("num_mel_bins", "num_coefficients", "snip_edges", "remove_dc_offset", "use_energy", "raw_energy"):
            d[k] = int(d[k])
        self.mfcc_opts = MfccOpts(**d)
        self.num_ceps = d["num_coefficients"]
        check(self.ctx, self.lib.mfa_mfcc_configure(self.ctx, C.byref(self.mfcc_opts)), "mfa_mfcc_configure")

    def load_gmm(self, gmm: DiagGmmModel) -> None:
        po = np.ascontiguousarray(gmm.pdf_offsets, dtype=np.int32)
        gc = np.ascontiguousarray(gmm.gconsts, dtype=np.float32)
        mi = np.ascontiguousarray(gmm.means_invvars, dtype=np.float32)
        iv = np.ascontiguousarray(gmm.inv_vars, dtype=np.float32)
        check(self.ctx, self.lib.mfa_load_gmm(self.ctx, gmm.dim, gmm.num_pdfs, po.ctypes.data, gc.ctypes.data,
                                              mi.ctypes.data, iv.ctypes.data), "mfa_load_gmm")
        self.gmm = gmm
        slots = np.array([self.lib.mfa_gmm_slot(self.ctx, p) for p in range(gmm.num_pdfs)], dtype=np.int32)
        n_gauss = np.diff(gmm.pdf_offsets)
        # classes in kernel order: 32 rows single block, 32 rows multi-block (> 32 Gaussians), 16, 8, 4, 1
        cls = np.full(gmm.num_pdfs, 5, dtype=np.int32)
        for i, s in enumerate((32, 16, 8, 4, 1)):
            cls[slots == s] = i + 1
        cls[(slots == 32) & (n_gauss <= 32)] = 0
        self.slot_class = cls

    def num_frames(self, num_samples: int) -> int:
        return int(self.lib.mfa_mfcc_num_frames(self.ctx, num_samples))

    def num_frames_array(self, num_samples: np.ndarray) -> np.ndarray:
        """``num_frames`` for a whole batch (Kaldi NumFrames, both snip_edges settings), with the window and shift computed as
        the library computes them (float32 arithmetic, mfa_mfcc_configure); checked against the library on first use."""
        n = np.asarray(num_samples, dtype=np.int64)
        if self.mfcc_opts is None:
            self.configure_mfcc()
        o = self.mfcc_opts
        f32 = np.float32
        win = int(f32(f32(o.sample_frequency) * f32(0.001)) * f32(o.frame_length_ms))
        shift = int(f32(f32(o.sample_frequency) * f32(0.001)) * f32(o.frame_shift_ms))
        snip = int(o.snip_edges)
        out = np.where(n < win, 0, 1 + (n - win) // shift) if snip else (n + shift // 2) // shift
        key = (win, shift, snip)
        if getattr(self, "_nf_checked", None) != key:
            for probe in (0, 1, win - 1, win, win + shift - 1, win + shift, 159999, 160000, 160001, 1 << 20):
                ref = (0 if probe < win else 1 + (probe - win) // shift) if snip else (probe + shift // 2) // shift
                if self.num_frames(probe) != ref:
                    raise _lib.MfaHipError("num_frames_array disagrees with mfa_mfcc_num_frames")
            self._nf_checked = key
        return out.astype(np.int64)

    # ------------------------------------------------------------------ stages
    _SLAB_BYTES = 512 << 20

    def _dev(self, a: np.ndarray) -> torch.Tensor:
        """Host array → device, without making the host wait for the device: the array is copied into a slab of pinned
        memory (bump allocation, wrap-around after a device synchronisation — once every several batches) and sent
        asynchronously.  A pageable ``.to(device)`` would block until everything queued on the stream before it has finished,
        i.e. until the previous batch has been decoded; ``Tensor.pin_memory()`` per array costs a pinned allocation (3 ms on
        this runtime)."""
        a = np.ascontiguousarray(a)
        nbytes = a.nbytes
        if nbytes == 0 or nbytes > self._SLAB_BYTES // 4:
            return torch.from_numpy(a).to(self.device)
        slab = getattr(self, "_slab", None)
        if slab is None:
            try:
                slab = self._slab = torch.empty(self._SLAB_BYTES, dtype=torch.uint8, pin_memory=True)
            except RuntimeError:
                self._slab = False
                return torch.from_numpy(a).to(self.device)
            self._slab_off = 0
        elif slab is False:
            return torch.from_numpy(a).to(self.device)
        off = (self._slab_off + 255) & ~255
        if off + nbytes > self._SLAB_BYTES:
            torch.cuda.synchronize(self.device)      # every copy out of the slab has run, whichever stream it was queued on
            off = 0
        self._slab_off = off + nbytes
        view = slab.numpy()[off: off + nbytes].view(a.dtype).reshape(a.shape)
        view[...] = a
        with torch.cuda.device(self.device):
            return torch.from_numpy(view).to(self.device, non_blocking=True)

    def frame_offsets(self, sample_off: np.ndarray) -> np.ndarray:
        frames = self.num_frames_array(np.diff(sample_off))
        return np.concatenate([[0], np.cumsum(frames)]).astype(np.int64)

    def mfcc(self, pcm: torch.Tensor, sample_off: np.ndarray, frame_off: Optional[np.ndarray] = None):
        """pcm: int16 [ΣN] on device.  Returns (mfcc float32 [ΣT, num_ceps], frame_off host int64 [n+1])."""
        if self.mfcc_opts is None:
            self.configure_mfcc()
        assert pcm.dtype == torch.int16 and pcm.is_cuda
        if frame_off is None:
            frame_off = self.frame_offsets(sample_off)
        n_utt = len(sample_off) - 1
        total = int(frame_off[-1])
        out = torch.empty((total, self.num_ceps), dtype=torch.float32, device=self.device)
        d_so, d_fo = self._dev(sample_off.astype(np.int64)), self._dev(frame_off)
        max_frames = int(np.diff(frame_off).max()) if n_utt else 0
        check(self.ctx, self.lib.mfa_mfcc_batch(self.ctx, _ptr(pcm), _ptr(d_so), _ptr(d_fo), n_utt, max_frames, _ptr(out)),
              "mfa_mfcc_batch")
        return out, frame_off

    def cmvn_stats(self, feats: torch.Tensor, frame_off: np.ndarray, utt2spk: np.ndarray, n_spk: int) -> torch.Tensor:
        n_utt = len(frame_off) - 1
        order = np.argsort(utt2spk, kind="stable").astype(np.int32)
        counts = np.bincount(utt2spk, minlength=n_spk)
        spk_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        dim = feats.shape[1]
        stats = torch.empty((n_spk, 2, dim + 1), dtype=torch.float64, device=self.device)
        d_fo, d_so, d_su = self._dev(frame_off), self._dev(spk_off), self._dev(order)
        check(self.ctx, self.lib.mfa_cmvn_stats(self.ctx, _ptr(feats), _ptr(d_fo), n_utt, dim, _ptr(d_so), _ptr(d_su), n_spk,
                                                _ptr(stats)), "mfa_cmvn_stats")
        return stats

    def features(self, mfcc: torch.Tensor, frame_off: np.ndarray, utt2spk: Optional[np.ndarray] = None,
                 cmvn: Optional[torch.Tensor] = None, lda: Optional[torch.Tensor] = None,
                 fmllr: Optional[torch.Tensor] = None, splice_context: int = 3) -> torch.Tensor:
        """CMVN → Δ+ΔΔ (no ``lda``) or splice+LDA(+fMLLR).  Mirrors FeatureArchive's chain (MFA/db.py:2101-2136)."""
        n_utt = len(frame_off) - 1
        dim = mfcc.shape[1]
        total = int(frame_off[-1])
        d_fo = self._dev(frame_off)
        d_u2s = self._dev(np.asarray(utt2spk, dtype=np.int32)) if utt2spk is not None else None
        max_frames = int(np.diff(frame_off).max()) if n_utt else 0
        if lda is None:
            out = torch.empty((total, 3 * dim), dtype=torch.float32, device=self.device)
            rc = self.lib.mfa_feats_batch(self.ctx, _ptr(mfcc), _ptr(d_fo), n_utt, max_frames, dim, _ptr(d_u2s), _ptr(cmvn), 0, 0,
                                          None, 0, 0, None, _ptr(out))
        else:
            rows, cols = lda.shape
            out = torch.empty((total, rows), dtype=torch.float32, device=self.device)
            rc = self.lib.mfa_feats_batch(self.ctx, _ptr(mfcc), _ptr(d_fo), n_utt, max_frames, dim, _ptr(d_u2s), _ptr(cmvn), 1,
                                          splice_context, _ptr(lda), rows, cols, _ptr(fmllr), _ptr(out))
        check(self.ctx, rc, "mfa_feats_batch")
        return out

    def sort_pdf_list(self, pdfs: np.ndarray, first_frame: Optional[np.ndarray] = None):
        """Order a pdf list by slot class as the scoring kernels require; returns (sorted, counts[6]) or, with
        ``first_frame`` keys, (sorted, counts[6], sorted keys) with ascending keys inside each class
        (mfa_gmm_sort_pdf_list / mfa_gmm_sort_pdf_list_keyed)."""
        pdfs = np.ascontiguousarray(pdfs, dtype=np.int32).copy()
        counts = np.zeros(6, dtype=np.int32)
        if first_frame is None:
            check(self.ctx, self.lib.mfa_gmm_sort_pdf_list(self.ctx, pdfs.ctypes.data, len(pdfs), counts.ctypes.data),
                  "mfa_gmm_sort_pdf_list")
            return pdfs, counts
        keys = np.ascontiguousarray(first_frame, dtype=np.int32).copy()
        check(self.ctx, self.lib.mfa_gmm_sort_pdf_list_keyed(self.ctx, pdfs.ctypes.data, keys.ctypes.data, len(pdfs),
                                                             counts.ctypes.data), "mfa_gmm_sort_pdf_list_keyed")
        return pdfs, counts, keys

    def state_first_frames(self, fst: Fst) -> np.ndarray:
        """depth[s] = fewest arcs from the start state to s = first frame a token can sit on s (mfa_fst_first_frames)."""
        arc_off = np.ascontiguousarray(fst.arc_offsets, dtype=np.int32)
        nxt = np.ascontiguousarray(fst.arcs["nextstate"], dtype=np.int32)
        depth = np.empty(fst.num_states, dtype=np.int32)
        if self.lib.mfa_fst_first_frames(fst.num_states, arc_off.ctypes.data, nxt.ctypes.data, int(fst.start),
                                         depth.ctypes.data) != 0:
            raise _lib.MfaHipError("mfa_fst_first_frames: malformed graph")
        return depth

    def state_last_depths(self, fst: Fst, bfs_depth: Optional[np.ndarray] = None) -> np.ndarray:
        """m[s] = smallest BFS depth among the states reachable from s (mfa_fst_last_depths): non-decreasing along every
        arc — the lower bound of the lazy-scoring band."""
        arc_off = np.ascontiguousarray(fst.arc_offsets, dtype=np.int32)
        nxt = np.ascontiguousarray(fst.arcs["nextstate"], dtype=np.int32)
        bfs = np.ascontiguousarray(self.state_first_frames(fst) if bfs_depth is None else bfs_depth, dtype=np.int32)
        depth = np.zeros(fst.num_states, dtype=np.int32)
        if self.lib.mfa_fst_last_depths(fst.num_states, arc_off.ctypes.data, nxt.ctypes.data, int(fst.start), bfs.ctypes.data,
                                        depth.ctypes.data) < 0:
            raise _lib.MfaHipError("mfa_fst_last_depths: malformed graph")
        return depth

    def score(self, feats: torch.Tensor, frame_off: np.ndarray, pdf_list: torch.Tensor, pdf_off_host: np.ndarray,
              class_counts: torch.Tensor, pdf_first_frame: Optional[torch.Tensor] = None):
        """Returns (loglikes float32 flat, ll_off host int64 [n+1], ll_cols int32 tensor).  With ``pdf_first_frame``
        (PackedGraphs.pdf_first_frame) cells no decoder token can ask for are left unwritten (zero here)."""
        n_utt = len(frame_off) - 1
        T = np.diff(frame_off)
        P = np.diff(pdf_off_host)
        ll_off = np.concatenate([[0], np.cumsum(T * P)]).astype(np.int64)
        alloc = torch.empty if pdf_first_frame is None else torch.zeros
        out = alloc(int(ll_off[-1]), dtype=torch.float32, device=self.device)
        d_fo, d_po, d_lo = self._dev(frame_off), self._dev(pdf_off_host.astype(np.int64)), self._dev(ll_off)
        max_frames = int(T.max()) if n_utt else 0
        check(self.ctx, self.lib.mfa_gmm_score_batch(self.ctx, _ptr(feats), _ptr(d_fo), n_utt, max_frames, _ptr(pdf_list),
                                                     _ptr(d_po), _ptr(class_counts), _ptr(pdf_first_frame), _ptr(d_lo),
                                                     _ptr(out)), "mfa_gmm_score_batch")
        return out, ll_off, self._dev(P.astype(np.int32))

    def pack_graphs_general(self, fsts: Sequence[Fst], tm: TransitionModel) -> PackedGraphs:
        """Device layout for the general-graph decoder (mfa_align_general_batch): graphs may hold epsilon input arcs
        (ilabel 0) and states of any out-degree.  One score column per pdf, no depth keys (scores are computed densely)."""
        n = len(fsts)
        S = np.array([f.num_states for f in fsts], dtype=np.int64)
        A = np.array([f.num_arcs for f in fsts], dtype=np.int64)
        state_off = np.concatenate([[0], np.cumsum(S)]).astype(np.int64)
        arc_base = np.concatenate([[0], np.cumsum(A)]).astype(np.int64)
        arc_off = np.concatenate([f.arc_offsets.astype(np.int32) for f in fsts]) if n else np.zeros(0, np.int32)
        final = np.concatenate([f.final for f in fsts]).astype(np.float32)
        arcs = np.concatenate([f.arcs for f in fsts])
        if np.any(arcs["ilabel"] < 0) or np.any(arcs["ilabel"] > tm.num_transition_ids):
            raise _lib.MfaHipError("a graph arc carries an input label outside the model's transition-ids")
        pdf_of_arc = tm.id2pdf[arcs["ilabel"]]            # -1 for epsilon arcs
        cols = np.zeros(arcs.shape[0], dtype=np.int32)
        pdf_lists, counts = [], []
        lut = np.full(tm.num_pdfs, -1, dtype=np.int32)
        for u in range(n):
            a0, a1 = int(arc_base[u]), int(arc_base[u + 1])
            pa = pdf_of_arc[a0:a1]
            emit = pa >= 0
            pl, cc = self.sort_pdf_list(np.unique(pa[emit])) if emit.any() else (np.zeros(0, np.int32), np.zeros(6, np.int32))
            lut[pl] = np.arange(pl.shape[0], dtype=np.int32)
            cols[a0:a1] = np.where(emit, lut[np.maximum(pa, 0)], 0)
            pdf_lists.append(pl)
            counts.append(cc)
        pdf_off = np.concatenate([[0], np.cumsum([len(p) for p in pdf_lists])]).astype(np.int64)
        t = dict(
            state_off=self._dev(state_off), arc_base=self._dev(arc_base),
            start=self._dev(np.array([f.start for f in fsts], dtype=np.int32)),
            arc_off=self._dev(arc_off), final=self._dev(final),
            arc_next=self._dev(arcs["nextstate"].astype(np.int32)), arc_weight=self._dev(arcs["weight"].astype(np.float32)),
            arc_col=self._dev(cols), arc_ilabel=self._dev(arcs["ilabel"].astype(np.int32)),
            arc_olabel=self._dev(arcs["olabel"].astype(np.int32)),
        )
        return PackedGraphs(n, int(S.max()) if n else 0, int(A.max()) if n else 0, int(A.sum()), t,
                            self._dev(np.concatenate(pdf_lists).astype(np.int32) if n else np.zeros(0, np.int32)),
                            self._dev(pdf_off), self._dev(np.stack(counts).astype(np.int32)), pdf_off, pdf_lists)

    def align_general(self, graphs: PackedGraphs, feats: torch.Tensor, frame_off: np.ndarray, beam: float = 10.0,
                      retry_beam: float = 40.0, acoustic_scale: float = 0.1, bp_tokens_per_frame: int = 512,
                      want_frame_likes: bool = False):
        """Alignment over graphs with epsilon input arcs / wide states: dense scores, then FasterDecoder as Kaldi runs it
        (ProcessNonemitting included), one GPU thread per utterance — mfa_align_general_batch."""
        n = graphs.n_utt
        total = int(frame_off[-1])
        dev = self.device
        ll, ll_off, ll_cols = self.score(feats, frame_off, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts)
        ali = torch.zeros(total, dtype=torch.int32, device=dev)
        words = torch.zeros(total, dtype=torch.int32, device=dev)
        n_words = torch.zeros(n, dtype=torch.int32, device=dev)
        like = torch.zeros(n, dtype=torch.float32, device=dev)
        status = torch.full((n,), -1, dtype=torch.int32, device=dev)
        flike = torch.zeros(total, dtype=torch.float32, device=dev) if want_frame_likes else None
        opts = AlignOpts(beam, retry_beam, acoustic_scale, 0, bp_tokens_per_frame)
        gs = graphs.struct()
        d_lo, d_fo = self._dev(ll_off), self._dev(frame_off)
        h_fo = np.ascontiguousarray(frame_off, dtype=np.int64)
        check(self.ctx, self.lib.mfa_align_general_batch(self.ctx, C.byref(gs), _ptr(ll), _ptr(d_lo), _ptr(ll_cols), _ptr(d_fo),
                                                         h_fo.ctypes.data, graphs.max_states, graphs.max_arcs, C.byref(opts),
                                                         _ptr(ali), _ptr(words), _ptr(n_words), _ptr(like), _ptr(flike),
                                                         _ptr(status)), "mfa_align_general_batch")
        return dict(ali=ali, words=words, n_words=n_words, like=like, status=status, frame_like=flike)

    @staticmethod
    def needs_general_decoder(fst: Fst) -> bool:
        """True for graphs the wavefront-parallel decoder does not take: a state with more than 64 emitting (or more than 64
        epsilon) arcs.  Epsilon input arcs as such are fine since round 3 (``pack_graphs`` stores every state's arcs [emitting | epsilon] and the decoder runs
        ProcessNonemitting after every frame)."""
        if not fst.num_arcs:
            return False
        eps = fst.arcs["ilabel"] == 0
        deg = np.diff(fst.arc_offsets)
        if not eps.any():
            return bool(int(deg.max()) > 64)
        src = np.repeat(np.arange(fst.num_states), deg)
        n_eps = np.bincount(src[eps], minlength=fst.num_states)
        return bool(int(n_eps.max()) > 64 or int((deg - n_eps).max()) > 64)

    def pack_graphs(self, fsts: Sequence[Fst], tm: TransitionModel, cluster_gap: Optional[int] = 32,
                    groups: Optional[int] = None, pool: Optional[StagingPool] = None) -> PackedGraphs:
        """Concatenate per-utterance graphs (with transition probabilities already applied) into the device layout.

        Score columns: one per (pdf, depth cluster) of the utterance.  A pdf whose arcs leave states far apart in the graph
        (the same phone in two words) gets one column per cluster of occurrences — occurrences whose BFS depths differ by
        more than ``cluster_gap``, or lie in different ``cluster_gap``-wide depth ranges, start a new column — so that the
        lazy-scoring band, which is a range of graph depths, does not have to keep the pdf alive for everything in
        between (optional silence after every word would otherwise chain into one column spanning the utterance).
        ``cluster_gap=None``: one column per pdf.

        ``groups`` (default 8, env MFA_PLAN_GROUPS): the single-block pdfs' columns are laid out in that many runs — pdf id
        mod groups — so that the lazy scoring kernel can give each run to one XCD (mfa_build_score_plan_grouped)."""
        if groups is None:
            groups = int(os.environ.get("MFA_PLAN_GROUPS", "8"))
        if cluster_gap == 32 and os.environ.get("MFA_PLAN_SPAN"):          # (diagnostics: tools/band_study.py)
            cluster_gap = int(os.environ["MFA_PLAN_SPAN"])
        n = len(fsts)
        whole = getattr(fsts, "arcs", None) is not None and len(getattr(fsts, "state_off", ())) == n + 1
        columns = whole and getattr(fsts, "arc_pdf", None) is not None and getattr(fsts, "arc_next", None) is not None \
            and getattr(fsts, "arc_off32", None) is not None
        if whole:
            # a batch from the native compiler (graph_native.FstBatch): its elements are views of these arrays
            state_off, arc_base = np.ascontiguousarray(fsts.state_off, dtype=np.int64), np.ascontiguousarray(fsts.arc_base, dtype=np.int64)
            S, A = np.diff(state_off), np.diff(arc_base)
            arc_off = fsts.arc_off32 if columns else fsts.arc_off.astype(np.int32)
            final, arcs = np.ascontiguousarray(fsts.final, dtype=np.float32), fsts.arcs
        else:
            S = np.array([f.num_states for f in fsts], dtype=np.int64)
            A = np.array([f.num_arcs for f in fsts], dtype=np.int64)
            state_off = np.concatenate([[0], np.cumsum(S)]).astype(np.int64)
            arc_base = np.concatenate([[0], np.cumsum(A)]).astype(np.int64)
            arc_off = np.concatenate([f.arc_offsets.astype(np.int32) for f in fsts]) if n else np.zeros(0, np.int32)
            final = np.concatenate([f.final for f in fsts]).astype(np.float32)
            arcs = np.concatenate([f.arcs for f in fsts])
        if self.slot_class is None:
            raise _lib.MfaHipError("pack_graphs needs the acoustic model's slot classes: call load_gmm first")
        nemit = None          # emitting arcs per state, for graphs with epsilon input arcs
        if columns:
            # (the compiler produced next-state and pdf columns; an input label outside the model was refused there)
            nxt, pdf_of_arc = fsts.arc_next, fsts.arc_pdf
            if arcs.shape[0] and (fsts.min_ilabel <= 0 if getattr(fsts, "min_ilabel", None) is not None else int(pdf_of_arc.min()) < 0):
                raise _lib.MfaHipError("a natively compiled batch cannot hold epsilon input arcs")
        else:
            il = arcs["ilabel"]
            if np.any(il < 0) or np.any(il > tm.num_transition_ids):
                raise _lib.MfaHipError("a graph arc carries an input label outside the model's transition-ids")
            is_eps = il == 0
            if is_eps.any():
                # epsilon input arcs (kalpy / Kaldi-compiled graphs): every state's arcs stored [emitting | epsilon], each
                # kind in its original order — FasterDecoder's two loops never see the interleaving (include/mfa_hip.h)
                deg = np.concatenate([np.diff(f.arc_offsets) for f in fsts]).astype(np.int64)
                src = np.repeat(np.arange(deg.shape[0], dtype=np.int64), deg)
                perm = np.argsort(src * 2 + is_eps, kind="stable")
                arcs = arcs[perm]
                is_eps = is_eps[perm]
                n_eps = np.bincount(src[perm][is_eps], minlength=deg.shape[0])
                nemit = (deg - n_eps).astype(np.int32)
                if int(n_eps.max()) > 64 or int(nemit.max()) > 64:
                    raise _lib.MfaHipError("a graph state has more than 64 arcs of a kind; the device decoder supports at most 64 "
                                           "emitting and 64 epsilon arcs per state (needs_general_decoder)")
            pdf_of_arc = np.ascontiguousarray(np.where(il[perm] > 0, tm.id2pdf[np.maximum(il[perm], 0)], -1) if nemit is not None
                                              else tm.id2pdf[il], dtype=np.int32)
            nxt = np.ascontiguousarray(arcs["nextstate"], dtype=np.int32)
        # (the concatenated offsets restart at every utterance: those steps are <= 0 and do not disturb the maximum)
        if columns and getattr(fsts, "max_degree", None) is not None:
            max_deg = int(fsts.max_degree)
        else:
            max_deg = int(np.diff(arc_off).max()) if arc_off.shape[0] > 1 else 0
        if max_deg > 64 and nemit is None:
            raise _lib.MfaHipError(f"a graph state has {max_deg} arcs; the device decoder supports at most 64")
        span = 0 if cluster_gap is None else int(cluster_gap)
        pdf_class = np.ascontiguousarray(self.slot_class, dtype=np.int32)
        # columns (pdf, depth cluster) in kernel order, their depth keys, and every arc's column: one host call for the batch,
        # utterances spread over the host's threads (mfa_build_score_plans_batch); outputs land in reused pinned staging
        # buffers and travel to the device asynchronously
        total_s, total_a = int(state_off[-1]), int(arc_base[-1])
        starts = np.zeros(n, dtype=np.int32) if whole else np.array([f.start for f in fsts], dtype=np.int32)
        # ``pool``: the staging buffers of this batch (the ones a native-compiled batch already lives in, when the caller
        # compiled it with ``alloc=pool.get``); default: the engine's next pair
        in_pool = pool is not None and getattr(fsts, "staging", None) is pool
        pool = pool or self.next_staging()
        sd_all = pool.get("sd", max(total_s, 1) * 2, np.int32).reshape(-1, 2)
        cols = pool.get("cols", max(total_a, 1), np.int32)
        cp, cf, cl = (pool.get(k, max(total_a, 1), np.int32) for k in ("cp", "cf", "cl"))
        cc_all = np.zeros((max(n, 1), 6), dtype=np.int32)
        gc_all = np.zeros((max(n, 1), max(groups, 1)), dtype=np.int32)
        n_cols = np.zeros(max(n, 1), dtype=np.int32)
        bad = C.c_int32(-1)
        rc = self.lib.mfa_build_score_plans_batch(n, state_off.ctypes.data, arc_base.ctypes.data, arc_off.ctypes.data,
                                                  nxt.ctypes.data, pdf_of_arc.ctypes.data, starts.ctypes.data,
                                                  int(pdf_class.shape[0]), pdf_class.ctypes.data, span, groups,
                                                  _host_threads(), sd_all.ctypes.data, cols.ctypes.data, cp.ctypes.data,
                                                  cf.ctypes.data, cl.ctypes.data, cc_all.ctypes.data, gc_all.ctypes.data,
                                                  n_cols.ctypes.data, C.byref(bad)) if n else 0
        if rc != 0:
            raise _lib.MfaHipError(f"mfa_build_score_plan: utterance {int(bad.value)}: " +
                                   ("a pdf id outside the loaded model" if rc == -2 else
                                    "unsupported number of plan groups" if rc == -3 else "malformed graph"))
        cols = cols[:total_a]
        n_cols = n_cols[:n].astype(np.int64)
        pdf_off = np.concatenate([[0], np.cumsum(n_cols)]).astype(np.int64)
        # utterance u's columns sit at [arc_base[u], arc_base[u] + n_cols[u]) of the column arrays: compact them
        take = np.repeat(arc_base[:-1] - pdf_off[:-1], n_cols) + np.arange(int(pdf_off[-1]), dtype=np.int64)
        pdf_all, first_all, last_all = cp[take], cf[take], cl[take]
        pdf_lists = RaggedView(pdf_all, pdf_off)
        first_frames = RaggedView(first_all, pdf_off)
        counts, group_counts = cc_all[:n], gc_all[:n]
        up = lambda a: pool.to_device(a, self.stream)       # noqa: E731  (pinned staging → device, asynchronous)
        if columns:
            # the 16-byte arc records travel once; their four columns are split on the device (strided device copies)
            flat = arcs.view(np.int32).reshape(-1)
            if not in_pool:
                stage = pool.get("arcs_copy", max(total_a, 1) * 4, np.int32)[: total_a * 4]
                stage[:] = flat
                flat = stage
            rec = up(flat).view(-1, 4)
            t = dict(
                state_off=self._dev(state_off), arc_base=self._dev(arc_base), start=self._dev(starts),
                arc_off=up(arc_off) if in_pool else self._dev(arc_off), final=up(final) if in_pool else self._dev(final),
                arc_next=rec[:, 3].contiguous(), arc_weight=rec[:, 2].contiguous().view(torch.float32),
                arc_col=up(cols), arc_ilabel=rec[:, 0].contiguous(), arc_olabel=rec[:, 1].contiguous(),
            )
            del rec
        else:
            t = dict(
                state_off=self._dev(state_off), arc_base=self._dev(arc_base),
                start=self._dev(starts),
                arc_off=self._dev(arc_off), final=self._dev(final),
                arc_next=self._dev(arcs["nextstate"].astype(np.int32)), arc_weight=self._dev(arcs["weight"].astype(np.float32)),
                arc_col=up(cols), arc_ilabel=self._dev(arcs["ilabel"].astype(np.int32)),
                arc_olabel=self._dev(arcs["olabel"].astype(np.int32)),
            )
        if nemit is not None:
            t["state_nemit"] = self._dev(nemit)
        sd_dev = up(sd_all[:total_s].reshape(-1)).view(-1, 2)
        return PackedGraphs(n, int(S.max()) if n else 0, int(A.max()) if n else 0, int(A.sum()), t, self._dev(pdf_all),
                            self._dev(pdf_off), self._dev(counts), pdf_off, pdf_lists,
                            self._dev(first_all), first_frames, self._dev(last_all),
                            sd_dev, groups,
                            self._dev(group_counts) if groups > 1 else None)

    def align(self, graphs: PackedGraphs, loglikes: torch.Tensor, ll_off: np.ndarray, ll_cols: torch.Tensor,
              frame_off: np.ndarray, beam: float = 10.0, retry_beam: float = 40.0, acoustic_scale: float = 0.1,
              max_tokens: int = 1024, bp_tokens_per_frame: int = 512, want_frame_likes: bool = False):
        """Returns device tensors: ali [ΣT], words [ΣT] (+ n_words [n]), like [n], status [n], frame_like or None."""
        n = graphs.n_utt
        total = int(frame_off[-1])
        dev = self.device
        ali = torch.zeros(total, dtype=torch.int32, device=dev)
        words = torch.zeros(total, dtype=torch.int32, device=dev)
        n_words = torch.zeros(n, dtype=torch.int32, device=dev)
        like = torch.zeros(n, dtype=torch.float32, device=dev)
        status = torch.full((n,), -1, dtype=torch.int32, device=dev)
        flike = torch.zeros(total, dtype=torch.float32, device=dev) if want_frame_likes else None
        opts = AlignOpts(beam, retry_beam, acoustic_scale, max_tokens, bp_tokens_per_frame)
        gs = graphs.struct()
        d_lo, d_fo = self._dev(ll_off), self._dev(frame_off)
        check(self.ctx, self.lib.mfa_align_batch(self.ctx, C.byref(gs), _ptr(loglikes), _ptr(d_lo), _ptr(ll_cols), _ptr(d_fo),
                                                 total, graphs.total_arcs, graphs.max_states, graphs.max_arcs, C.byref(opts), _ptr(ali), _ptr(words),
                                                 _ptr(n_words), _ptr(like), _ptr(flike), _ptr(status)), "mfa_align_batch")
        return dict(ali=ali, words=words, n_words=n_words, like=like, status=status, frame_like=flike)

    def gather_rows(self, feats: torch.Tensor, rows: np.ndarray) -> torch.Tensor:
        """Rows ``rows`` of a device matrix as a new contiguous device matrix (device-to-device copies of the contiguous
        runs; no arithmetic)."""
        rows = np.asarray(rows, dtype=np.int64)
        out = torch.empty((rows.shape[0], feats.shape[1]), dtype=feats.dtype, device=feats.device)
        if rows.shape[0] == 0:
            return out
        cuts = np.nonzero(np.diff(rows) != 1)[0] + 1
        starts = np.concatenate([[0], cuts]); ends = np.concatenate([cuts, [rows.shape[0]]])
        for a, b in zip(starts, ends):
            out[a:b].copy_(feats[int(rows[a]): int(rows[a]) + int(b - a)])
        return out

    def align_features(self, graphs: PackedGraphs, feats: torch.Tensor, frame_off: np.ndarray, beam: float = 10.0,
                       retry_beam: float = 40.0, acoustic_scale: float = 0.1, max_tokens: int = 1024,
                       bp_tokens_per_frame: int = 512, want_frame_likes: bool = False, window: int = 64,
                       loglikes: Optional[torch.Tensor] = None):
        """features + graphs → alignments with acoustic scores evaluated lazily, window by window, for the pdfs live tokens
        can reach (mfa_align_features_batch) — the shape of GmmAligner.align_utterance(fst, feats)
        (MFA/alignment/multiprocessing.py:846-853).  Same results as ``score`` + ``align``.  ``loglikes``: optional scratch
        [Σ T·P] (zero-filled here when omitted, so tests can see which cells were written)."""
        n = graphs.n_utt
        total = int(frame_off[-1])
        dev = self.device
        T = np.diff(frame_off)
        P = np.diff(graphs.pdf_off_host)
        ll_off = np.concatenate([[0], np.cumsum(T * P)]).astype(np.int64)
        if loglikes is None:
            loglikes = torch.zeros(int(ll_off[-1]), dtype=torch.float32, device=dev)
        ali = torch.zeros(total, dtype=torch.int32, device=dev)
        words = torch.zeros(total, dtype=torch.int32, device=dev)
        n_words = torch.zeros(n, dtype=torch.int32, device=dev)
        like = torch.zeros(n, dtype=torch.float32, device=dev)
        status = torch.full((n,), -1, dtype=torch.int32, device=dev)
        flike = torch.zeros(total, dtype=torch.float32, device=dev) if want_frame_likes else None
        opts = AlignOpts(beam, retry_beam, acoustic_scale, max_tokens, bp_tokens_per_frame)
        gs, plan = graphs.struct(), graphs.plan()
        d_lo, d_fo, d_cols = self._dev(ll_off), self._dev(frame_off), self._dev(P.astype(np.int32))
        check(self.ctx, self.lib.mfa_align_features_batch(
            self.ctx, C.byref(gs), C.byref(plan), _ptr(feats), _ptr(d_fo), int(T.max()) if n else 0, total, graphs.total_arcs,
            graphs.max_states, graphs.max_arcs, C.byref(opts), int(window), _ptr(loglikes), _ptr(d_lo), _ptr(d_cols), _ptr(ali),
            _ptr(words), _ptr(n_words), _ptr(like), _ptr(flike), _ptr(status)), "mfa_align_features_batch")
        return dict(ali=ali, words=words, n_words=n_words, like=like, status=status, frame_like=flike, loglikes=loglikes,
                    ll_off=ll_off)

    # ------------------------------------------------------------------ timing helpers (bench.py)
    def kernel_timing(self, enable: bool) -> None:
        check(self.ctx, self.lib.mfa_kernel_timing(self.ctx, int(enable)), "mfa_kernel_timing")

    def kernel_times(self) -> Dict[str, Dict[str, float]]:
        out = {}
        for i, name in enumerate(("mfcc", "cmvn", "feats", "gmm", "viterbi")):
            ms, n = C.c_float(0), C.c_int(0)
            check(self.ctx, self.lib.mfa_kernel_time_ms(self.ctx, i, C.byref(ms), C.byref(n)), "mfa_kernel_time_ms")
            out[name] = dict(ms=float(ms.value), launches=int(n.value))
        return out

    def reset_kernel_times(self) -> None:
        check(self.ctx, self.lib.mfa_kernel_time_reset(self.ctx), "mfa_kernel_time_reset")


class Pipeline:
    """A fixed-shape batch whose inputs, offsets, graphs and output buffers all live in HBM, so that one ``step()`` is
    nothing but the five kernel launches of the hot path (MFCC → CMVN stats → features → GMM scores → Viterbi).

    This is the unit bench.py times and the shape a corpus aligner would feed: build once per batch shape, refill the
    PCM/graph tensors, call ``step()``.
    """

    def __init__(self, engine: AlignmentEngine, pcm: torch.Tensor, sample_off: np.ndarray, utt2spk: np.ndarray,
                 graphs: PackedGraphs, lda: Optional[torch.Tensor] = None, fmllr: Optional[torch.Tensor] = None,
                 splice_context: int = 3, beam: float = 10.0, retry_beam: float = 40.0, acoustic_scale: float = 0.1,
                 max_tokens: int = 1024, bp_tokens_per_frame: int = 256, reachability: bool = True, lazy: bool = True,
                 window: int = 64):
        e = self.e = engine
        # lazy: scores are evaluated window by window for the pdfs live decoder tokens can reach
        # (mfa_align_features_batch); otherwise the (reachability-bounded) matrix is scored first, then decoded
        self.lazy = bool(lazy) and graphs.pdf_last_depth is not None
        self.window = int(window)
        # reachability: score a pdf only from the first frame a decoder token can ask for it (Kaldi evaluates its
        # decodable lazily; the dense score matrix here skips the cells that provably are never read)
        self.reachability = bool(reachability) and graphs.pdf_first_frame is not None
        dev = e.device
        self.pcm = pcm
        self.n_utt = len(sample_off) - 1
        self.graphs = graphs
        self.lda, self.fmllr, self.ctx_frames = lda, fmllr, splice_context
        self.frame_off = e.frame_offsets(sample_off)
        T = np.diff(self.frame_off)
        self.total_frames = int(self.frame_off[-1])
        self.max_frames = int(T.max())
        P = np.diff(graphs.pdf_off_host)
        self.ll_off = np.concatenate([[0], np.cumsum(T * P)]).astype(np.int64)
        utt2spk = np.asarray(utt2spk, dtype=np.int32)
        spk_ids, inv = np.unique(utt2spk, return_inverse=True)
        self.n_spk = len(spk_ids)
        order = np.argsort(inv, kind="stable").astype(np.int32)
        spk_off = np.concatenate([[0], np.cumsum(np.bincount(inv, minlength=self.n_spk))]).astype(np.int32)
        d = e._dev
        self.d_sample_off, self.d_frame_off = d(sample_off.astype(np.int64)), d(self.frame_off)
        self.d_utt2spk, self.d_spk_off, self.d_spk_utt = d(inv.astype(np.int32)), d(spk_off), d(order)
        self.d_ll_off, self.d_ll_cols = d(self.ll_off), d(P.astype(np.int32))
        self.num_ceps = e.num_ceps
        self.feat_dim = 3 * self.num_ceps if lda is None else int(lda.shape[0])
        f32, i32 = torch.float32, torch.int32
        self.mfcc = torch.empty((self.total_frames, self.num_ceps), dtype=f32, device=dev)
        self.cmvn = torch.empty((self.n_spk, 2, self.num_ceps + 1), dtype=torch.float64, device=dev)
        self.feats = torch.empty((self.total_frames, self.feat_dim), dtype=f32, device=dev)
        self.loglikes = torch.empty(int(self.ll_off[-1]), dtype=f32, device=dev)
        self.ali = torch.zeros(self.total_frames, dtype=i32, device=dev)
        self.words = torch.zeros(self.total_frames, dtype=i32, device=dev)
        self.n_words = torch.zeros(self.n_utt, dtype=i32, device=dev)
        self.like = torch.zeros(self.n_utt, dtype=f32, device=dev)
        self.status = torch.full((self.n_utt,), -1, dtype=i32, device=dev)
        self.opts = AlignOpts(beam, retry_beam, acoustic_scale, max_tokens, bp_tokens_per_frame)
        self.gstruct = graphs.struct()
        # algorithmic work of one step: GMM flops 4·D·g·P·T summed over utterances
        # With reachability a (frame, pdf) cell is algorithmically needed only from the pdf's first possible frame on.
        g_of_pdf = np.diff(e.gmm.pdf_offsets).astype(np.float64)
        cells = 0.0
        for u, pl in enumerate(graphs.pdf_lists_host):
            if self.reachability and graphs.pdf_first_frame_host is not None:
                live = np.clip(float(T[u]) - graphs.pdf_first_frame_host[u].astype(np.float64), 0.0, None)
            else:
                live = np.full(len(pl), float(T[u]))
            cells += float((g_of_pdf[pl] * live).sum())
        self.gmm_flops = 4.0 * e.gmm.dim * cells
        self.audio_seconds = float(np.diff(sample_off).sum() / e.mfcc_opts.sample_frequency)

    def step(self) -> None:
        self.front()
        if self.lazy:
            self.score_and_decode()
        else:
            self.score()
            self.decode()

    def score_and_decode(self) -> None:
        """features + graphs → alignments, scores evaluated lazily per window (mfa_align_features_batch)."""
        L, c, g = self.e.lib, self.e.ctx, self.graphs
        if not hasattr(self, "_plan"):
            self._plan = g.plan()
        check(c, L.mfa_align_features_batch(
            c, C.byref(self.gstruct), C.byref(self._plan), _ptr(self.feats), _ptr(self.d_frame_off), self.max_frames,
            self.total_frames, g.total_arcs, g.max_states, g.max_arcs, C.byref(self.opts), self.window, _ptr(self.loglikes),
            _ptr(self.d_ll_off), _ptr(self.d_ll_cols), _ptr(self.ali), _ptr(self.words), _ptr(self.n_words), _ptr(self.like),
            None, _ptr(self.status)), "mfa_align_features_batch")

    # ---- host side of the boundary: alignments land in (pinned) host memory
    def host_output_buffers(self, pinned: bool = True) -> Dict[str, torch.Tensor]:
        out = {}
        for name in ("ali", "words", "n_words", "like", "status"):
            t = getattr(self, name)
            out[name] = torch.empty(t.shape, dtype=t.dtype, pin_memory=pinned)
        return out

    def outputs_to_host(self, host: Dict[str, torch.Tensor]) -> None:
        """Asynchronous D2H of ali / words / n_words / like / status on the engine's stream."""
        for name, dst in host.items():
            dst.copy_(getattr(self, name), non_blocking=True)

    # ---- bench.py reporting helpers
    def scores_string(self) -> str:
        if self.lazy:
            return (f"lazy: per {self.window}-frame window, only the pdfs arcs within {self.window} arcs of the live tokens "
                    "can emit (a superset of what Kaldi's lazy decodable evaluates)")
        return "reachable cells only (pdf j from its first possible frame on)" if self.reachability else "dense T x P"

    def dtype_string(self, mono: bool, gauss_per_pdf: int) -> str:
        import os
        split = os.environ.get("MFA_GMM_BF16", "1") != "0" and not mono and gauss_per_pdf != 1
        f16 = split and os.environ.get("MFA_GMM_F16", "1") != "0"
        if f16:
            return "f32 scores from 2-way f16-split products (3*2^-22 per term worst case), f64 path costs"
        if split:
            return "f32 scores from 3-way bf16-split products (2^-24 per term), f64 path costs"
        return "f32 (scores), f64 (path costs)"

    def algorithmic_bytes(self, score_cells_read: Optional[float] = None) -> Dict[str, float]:
        """SURVEY §8(d) staged-pipeline bytes of one step, per stage (each tensor written once and read once).  The score
        matrix counts for what is actually moved: the scoring stage writes the cells it computes, the decoder reads
        ``score_cells_read`` cells (counted by the oracle's lazy decodable on a sample — the device decoder asks for exactly
        the same cells) or, without that count, at most the cells written."""
        T = np.diff(self.frame_off).astype(np.float64)
        P = np.diff(self.graphs.pdf_off_host).astype(np.float64)
        n_samples = float(self.pcm.numel())
        A = float(self.graphs.total_arcs)
        S = float(self.graphs.tensors["final"].numel())
        D = float(self.feat_dim)
        tot = float(T.sum())
        written = getattr(self, "cells_scored", None) if self.lazy else None
        if written is None:
            written = float((P * T).sum())
        if score_cells_read is not None:
            read, how = float(score_cells_read), "4 B x cells the decoder read (oracle's lazy-decodable count on the CPU sample, scaled to the batch)"
        else:
            read, how = written, "4 B x cells the scoring stage wrote (upper bound of the cells the decoder read)"
        if not hasattr(self, "_states_times_frames"):
            self._states_times_frames = float((T * np.diff(self.graphs.tensors["state_off"].cpu().numpy())).sum())
        return {
            "mfcc": 2.0 * n_samples + 4.0 * self.num_ceps * tot,
            "feats": 4.0 * self.num_ceps * tot + 4.0 * D * tot,
            "gmm": 4.0 * D * tot + 4.0 * written,                  # model slice cache-resident (SURVEY's 6.0 MB variant)
            "viterbi": 4.0 * read + 16.0 * A + 2.0 * self._states_times_frames + 6.0 * tot,
            "viterbi_score_bytes_how": how,
            "states": S,
        }

    def measure_scored_cells(self) -> Dict[str, float]:
        """One step on a zero-filled score scratch: which cells did the scoring kernels write?  Returns the count, the
        fraction of the T x P matrix, and the flops those cells cost (4*D*g per cell, g = Gaussians of the cell's pdf) —
        the work the scoring stage EXECUTES, which is what bench.py's roofline prices."""
        self.loglikes.zero_()
        self.step()
        torch.cuda.synchronize(self.e.device)
        g_of_pdf = torch.from_numpy(np.diff(self.e.gmm.pdf_offsets).astype(np.float64)).to(self.e.device)
        pdf_list = self.graphs.pdf_list.long()
        T = np.diff(self.frame_off)
        cells = torch.zeros((), dtype=torch.float64, device=self.e.device)
        weighted = torch.zeros((), dtype=torch.float64, device=self.e.device)
        po = self.graphs.pdf_off_host
        for u in range(self.n_utt):
            a, b = int(self.ll_off[u]), int(self.ll_off[u + 1])
            if b == a:
                continue
            per_col = (self.loglikes[a:b].view(int(T[u]), -1) != 0).sum(dim=0).to(torch.float64)
            cells += per_col.sum()
            weighted += (per_col * g_of_pdf[pdf_list[int(po[u]): int(po[u + 1])]]).sum()
        self.cells_scored = float(cells.item())
        self.cells_scored_fraction = self.cells_scored / max(1, self.loglikes.numel())
        self.scored_flops = 4.0 * self.e.gmm.dim * float(weighted.item())
        return {"cells": self.cells_scored, "fraction": self.cells_scored_fraction, "flops": self.scored_flops}

    def roofline(self, dominant: str, ktimes: Dict[str, Dict[str, float]], steps: int, mono: bool, gauss_per_pdf: int,
                 score_cells_read: Optional[float] = None) -> Dict:
        """Roofline object for a stage of the step (bench.py): per-step stage time from the HIP-event timers.  Everything
        priced here is work the kernels EXECUTE: scoring = the cells the lazy path wrote (``measure_scored_cells``), the
        decoder's score bytes = the cells it read (``score_cells_read`` when the oracle counted them on a sample, else the
        cells written — an upper bound)."""
        import os
        ms = ktimes[dominant]["ms"] / max(1, steps)
        launches = ktimes[dominant]["launches"] / max(1, steps)
        lazy_measured = self.lazy and getattr(self, "scored_flops", None) is not None
        if dominant == "gmm":
            split = os.environ.get("MFA_GMM_BF16", "1") != "0" and not mono and gauss_per_pdf != 1
            f16 = split and os.environ.get("MFA_GMM_F16", "1") != "0"
            mult = 3.0 if f16 else (6.0 if split else 1.0)
            peak = 2500.0 if split else 157.3
            flops = self.scored_flops if lazy_measured else self.gmm_flops
            ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            equiv = self.gmm_flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"kernel": "diagonal-GMM scoring (" + ("f16x2" if f16 else "bf16x3" if split else "f32") + " MFMA)",
                    "kernel_key": "gmm", "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None,
                    "executed_flops_per_step": flops, "mfma_flops_per_algorithmic_flop": int(mult),
                    "executed_mfma_tflops": round(mult * ach, 3), "executed_mfma_frac_of_peak": round(mult * ach / peak, 4),
                    "cells_scored_fraction": getattr(self, "cells_scored_fraction", None),
                    "algorithmic_equivalent_tflops": round(equiv, 3), "ms_per_step": round(ms, 4), "launches_per_step": launches,
                    "note": "achieved = 4*D*g flops of the (frame, pdf) cells the scoring kernels actually computed / stage time; "
                            "executed_mfma_* = the same times the split-operand products per term; algorithmic_equivalent_tflops "
                            "prices every cell of the reachability-bounded matrix (what a dense scorer would do) and is NOT work done"}
        by = self.algorithmic_bytes(score_cells_read)
        b = by.get(dominant, 0.0)
        gbs = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        names = {"viterbi": "viterbi kernels (beam Viterbi, one wavefront per utterance)", "mfcc": "mfcc_kernel",
                 "feats": "feats_lda_kernel / feats_kernel", "cmvn": "cmvn kernels"}
        out = {"kernel": names.get(dominant, dominant), "kernel_key": dominant, "bound": "hbm", "achieved": round(gbs, 2),
               "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 5), "traffic": None,
               "algorithmic_bytes_per_step": b, "ms_per_step": round(ms, 4), "launches_per_step": launches,
               "note": "algorithmic bytes = SURVEY 8(d) staged-pipeline bytes of this stage (each tensor written once, read once)"}
        if dominant == "viterbi":
            out["score_bytes"] = by["viterbi_score_bytes_how"]
        return out

    def front(self) -> None:
        """PCM → MFCC → CMVN statistics → final features (three launches on the engine's stream)."""
        L, c = self.e.lib, self.e.ctx
        check(c, L.mfa_mfcc_batch(c, _ptr(self.pcm), _ptr(self.d_sample_off), _ptr(self.d_frame_off), self.n_utt,
                                  self.max_frames, _ptr(self.mfcc)), "mfa_mfcc_batch")
        check(c, L.mfa_cmvn_stats(c, _ptr(self.mfcc), _ptr(self.d_frame_off), self.n_utt, self.num_ceps, _ptr(self.d_spk_off),
                                  _ptr(self.d_spk_utt), self.n_spk, _ptr(self.cmvn)), "mfa_cmvn_stats")
        if self.lda is None:
            rc = L.mfa_feats_batch(c, _ptr(self.mfcc), _ptr(self.d_frame_off), self.n_utt, self.max_frames, self.num_ceps,
                                   _ptr(self.d_utt2spk), _ptr(self.cmvn), 0, 0, None, 0, 0, None, _ptr(self.feats))
        else:
            rc = L.mfa_feats_batch(c, _ptr(self.mfcc), _ptr(self.d_frame_off), self.n_utt, self.max_frames, self.num_ceps,
                                   _ptr(self.d_utt2spk), _ptr(self.cmvn), 1, self.ctx_frames, _ptr(self.lda),
                                   int(self.lda.shape[0]), int(self.lda.shape[1]), _ptr(self.fmllr), _ptr(self.feats))
        check(c, rc, "mfa_feats_batch")

    def score(self) -> None:
        """features → GMM log-likelihoods of every utterance's pdf list."""
        L, c, g = self.e.lib, self.e.ctx, self.graphs
        check(c, L.mfa_gmm_score_batch(c, _ptr(self.feats), _ptr(self.d_frame_off), self.n_utt, self.max_frames,
                                       _ptr(g.pdf_list), _ptr(g.pdf_off), _ptr(g.class_counts),
                                       _ptr(g.pdf_first_frame) if self.reachability else None, _ptr(self.d_ll_off),
                                       _ptr(self.loglikes)), "mfa_gmm_score_batch")

    def decode(self) -> None:
        """log-likelihoods + graphs → alignments (beam Viterbi, retry beam for the utterances that need it)."""
        L, c, g = self.e.lib, self.e.ctx, self.graphs
        check(c, L.mfa_align_batch(c, C.byref(self.gstruct), _ptr(self.loglikes), _ptr(self.d_ll_off), _ptr(self.d_ll_cols),
                                   _ptr(self.d_frame_off), self.total_frames, g.total_arcs, g.max_states, g.max_arcs, C.byref(self.opts), _ptr(self.ali),
                                   _ptr(self.words), _ptr(self.n_words), _ptr(self.like), None, _ptr(self.status)),
              "mfa_align_batch")


def fmllr_statistics(engine: "AlignmentEngine", feats: torch.Tensor, frame_off: np.ndarray, ali: torch.Tensor,
                     tm: TransitionModel, utt2spk: np.ndarray, silence_phones: Sequence[int], silence_weight: float = 0.0,
                     stats_model: Optional[DiagGmmModel] = None):
    """Per-speaker fMLLR statistics from first-pass alignments (mfa_fmllr_acc_batch).

    ``ali``: int32 [ΣT] transition-ids (0 where an utterance failed: those frames get weight 0).  Returns
    (speaker ids, beta [S], K [S,D,D+1], G [S,D,D+1,D+1]) as float64 numpy arrays.

    ``stats_model``: the two-model form the reference runs for models that ship ``final.alimdl``
    (FmllrComputer(ali_model_path, model_path, …), MFA/corpus/features.py:503-511): posteriors from the model loaded in the
    engine (the alignment model), statistics with ``stats_model``'s means and variances (``final.mdl``)."""
    dev = engine.device
    if stats_model is not None:
        po = np.ascontiguousarray(stats_model.pdf_offsets, dtype=np.int32)
        mi = np.ascontiguousarray(stats_model.means_invvars, dtype=np.float32)
        iv = np.ascontiguousarray(stats_model.inv_vars, dtype=np.float32)
        check(engine.ctx, engine.lib.mfa_fmllr_stats_model(engine.ctx, stats_model.dim, stats_model.num_pdfs, po.ctypes.data,
                                                           mi.ctypes.data, iv.ctypes.data), "mfa_fmllr_stats_model")
    else:
        check(engine.ctx, engine.lib.mfa_fmllr_stats_model(engine.ctx, 0, 0, None, None, None), "mfa_fmllr_stats_model")
    id2pdf = torch.from_numpy(np.maximum(tm.id2pdf, 0).astype(np.int32)).to(dev)
    sil = np.zeros(tm.id2phone.shape[0], dtype=np.float32)
    sil[np.isin(tm.id2phone, np.asarray(list(silence_phones), dtype=np.int64))] = 1.0
    w_host = np.where(sil > 0, np.float32(silence_weight), np.float32(1.0)).astype(np.float32)
    w_host[0] = 0.0
    w_of_tid = torch.from_numpy(w_host).to(dev)
    ali = ali.to(torch.int32).contiguous() if ali.dtype != torch.int32 else ali.contiguous()
    pdf = torch.empty(ali.shape[0], dtype=torch.int32, device=dev)       # scratch: the lookup runs inside the library
    weight = torch.empty(ali.shape[0], dtype=torch.float32, device=dev)
    spk_ids, inv = np.unique(np.asarray(utt2spk), return_inverse=True)
    n_spk = len(spk_ids)
    order = np.argsort(inv, kind="stable").astype(np.int32)
    spk_off = np.concatenate([[0], np.cumsum(np.bincount(inv, minlength=n_spk))]).astype(np.int32)
    D = feats.shape[1]
    beta = torch.zeros(n_spk, dtype=torch.float64, device=dev)
    K = torch.zeros((n_spk, D, D + 1), dtype=torch.float64, device=dev)
    G = torch.zeros((n_spk, D, D + 1, D + 1), dtype=torch.float64, device=dev)
    d_fo, d_so, d_su = engine._dev(frame_off), engine._dev(spk_off), engine._dev(order)
    check(engine.ctx, engine.lib.mfa_fmllr_acc_ali_batch(engine.ctx, _ptr(feats), _ptr(d_fo), len(frame_off) - 1,
                                                          int(frame_off[-1]), _ptr(ali), _ptr(id2pdf), _ptr(w_of_tid),
                                                          int(id2pdf.shape[0]), _ptr(pdf), _ptr(weight), _ptr(d_so), _ptr(d_su),
                                                          n_spk, _ptr(beta), _ptr(K), _ptr(G)), "mfa_fmllr_acc_ali_batch")
    return spk_ids, beta.cpu().numpy(), K.cpu().numpy(), G.cpu().numpy()
