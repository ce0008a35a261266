"""Corpus-level driver: the fan-out / fan-in of ``AlignMixin.align_utterances`` (MFA/alignment/mixins.py:282-380) and the
two-pass flow of ``CorpusAligner.align`` (MFA/alignment/base.py:510-539) over the device pipeline, ending in the interval
and TextGrid layer (MFA/alignment/multiprocessing.py:1733-1751, MFA/textgrid.py:463-572).

The reference fans utterances out to worker processes that each walk Kaldi table files one utterance at a time; here a
rank takes the speakers ``sharding.assign_speakers`` gives it, cuts its utterances into length-bucketed ragged batches and
pushes each batch through MFCC → CMVN → features → scores → Viterbi entirely in HBM.  Per-speaker CMVN needs every
utterance of a speaker, so statistics are accumulated over the whole shard first (a cheap pass: MFCC + statistics
kernels), exactly like ``calc_cmvn`` runs before alignment in the reference.

A failed utterance yields ``None`` and is counted, never raised (MFA/alignment/mixins.py:308-314); the reason is kept in
``CorpusAligner.failure_reasons``.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import ctm as _ctm
from . import _lib
from . import fmllr as _fmllr
from . import graph as _graph
from . import sharding
from .engine import AlignmentEngine, fmllr_statistics
from .model import DiagGmmModel, TransitionModel


@dataclass
class CorpusUtterance:
    utt_id: str
    speaker: str
    pcm: np.ndarray                 # int16 mono at the model's sample rate
    text: str
    begin: float = 0.0              # position inside its sound file (for TextGrid export)
    file_name: Optional[str] = None
    file_duration: Optional[float] = None


class UtteranceResult:
    """One aligned utterance.  ``alignment`` / ``words`` are views of the batch arrays the device handed back; ``ctm`` — the
    ``HierarchicalCtm`` of MFA/alignment/multiprocessing.py:1733-1751 — is built from the batch's interval arrays the first
    time it is asked for (``CorpusAligner.export_textgrids`` writes files from the arrays and never asks)."""

    __slots__ = ("utt_id", "speaker", "alignment", "words", "likelihood", "num_frames", "_ctm", "_lazy")

    def __init__(self, utt_id: str, speaker: str, alignment: np.ndarray, words: np.ndarray, likelihood: float, num_frames: int,
                 ctm: Optional[_ctm.HierarchicalCtm] = None):
        self.utt_id, self.speaker, self.alignment, self.words = utt_id, speaker, alignment, words
        self.likelihood, self.num_frames = likelihood, num_frames          # total; MFA reports likelihood / num_frames
        self._ctm = ctm
        self._lazy = None            # (IntervalBatch, index, text, begin, end) until the objects are wanted

    @property
    def per_frame_likelihood(self) -> float:
        return self.likelihood / max(1, self.num_frames)

    @property
    def ctm(self) -> Optional[_ctm.HierarchicalCtm]:
        if self._ctm is None and self._lazy is not None:
            batch, k, text, begin, end = self._lazy
            self._ctm = batch.ctm(k, text=text, begin=begin, end=end, likelihood=self.per_frame_likelihood)
            self._lazy = None
        return self._ctm

    @ctm.setter
    def ctm(self, value) -> None:
        self._ctm, self._lazy = value, None

    @property
    def has_intervals(self) -> bool:
        return self._ctm is not None or self._lazy is not None

    def __getstate__(self):          # (rank-to-rank gather: the objects travel, not the batch arrays they would be built from)
        return (self.utt_id, self.speaker, np.array(self.alignment), np.array(self.words), self.likelihood, self.num_frames, self.ctm)

    def __setstate__(self, st):
        self.utt_id, self.speaker, self.alignment, self.words, self.likelihood, self.num_frames, self._ctm = st
        self._lazy = None


@dataclass
class AlignOptions:
    beam: float = 10.0
    retry_beam: float = 40.0
    transition_scale: float = 1.0
    acoustic_scale: float = 0.1
    self_loop_scale: float = 0.1
    boost_silence: float = 1.0
    max_tokens: int = 1024
    bp_tokens_per_frame: int = 256
    batch_frames: int = 2_000_000   # frames per device batch (≈ 2 000 ten-second utterances)
    fmllr_min_count: float = 500.0
    silence_weight: float = 0.0
    # `mfa align` (the corpus path) stores raw MFCCs and the CMVN-applied features as 8-bit Kaldi CompressedMatrix tables
    # (MFA/corpus/features.py:235, :356-365) and therefore aligns features quantised twice; `align_one` / the online path
    # keep float32 (the default here).  True reproduces the corpus path's two quantisations (host-side codec).
    corpus_compression: bool = False
    # raw MFCCs of a run stay on the device between the CMVN pass and the alignment passes up to this many bytes
    mfcc_cache_bytes: int = 64 << 30


class _BatchOut:
    """What the device handed back for one batch, on the host: ragged arrays in the device layout (utterance k of the batch
    at frame_off[k]; its word ids packed at the same offset)."""

    __slots__ = ("idx", "frame_off", "ali", "words", "n_words", "like", "status", "intervals")

    def __init__(self, idx, frame_off, ali, words, n_words, like, status):
        self.idx, self.frame_off, self.ali, self.words = list(idx), np.asarray(frame_off, dtype=np.int64), ali, words
        self.n_words, self.like, self.status = n_words, like, status
        self.intervals = None          # intervals_native.IntervalBatch once extracted


class CorpusAligner:
    """``CorpusAligner(tm, am, tree, lexicon, lda=None, ali_am=None)`` then ``align(utterances)`` / ``export_textgrids(...)``.

    ``ali_am``: the speaker-independent alignment model of a SAT acoustic model (``final.alimdl``).  When given, the first
    pass runs on it (MFA/alignment/mixins.py:404-410), fMLLR statistics take their posteriors from it and their means and
    variances from ``am`` (MFA/corpus/features.py:503-511), the second pass runs on ``am`` with the transforms, and an
    utterance that fails the second pass keeps its first-pass alignment (MFA/alignment/multiprocessing.py:841-863,
    :1784-1860)."""

    def __init__(self, tm: TransitionModel, am: DiagGmmModel, tree, lexicon, lda: Optional[np.ndarray] = None,
                 options: Optional[AlignOptions] = None, device: int = 0, engine: Optional[AlignmentEngine] = None,
                 mfcc_options: Optional[dict] = None, silence_phones: Sequence[int] = (),
                 ali_am: Optional[DiagGmmModel] = None, lazy: bool = True):
        self.tm, self.tree, self.lexicon = tm, tree, lexicon
        self.lda = None if lda is None else np.asarray(lda, dtype=np.float32)
        self.opt = options or AlignOptions()
        self.engine = engine or AlignmentEngine(device)
        self.mfcc_options = dict(mfcc_options or {})
        self.engine.configure_mfcc(**self.mfcc_options)
        self.silence_phones = list(silence_phones)
        self.lazy = bool(lazy)
        # boost_silence scales the weights of the silence pdfs (GmmAligner.boost_silence, MFA/alignment/multiprocessing.py:
        # 803-815) of whichever model a pass aligns with — on COPIES: the caller's models are left as they were
        self.am = self._boosted(am)
        self.ali_am = None if ali_am is None else self._boosted(ali_am)
        if self.ali_am is not None and not np.array_equal(self.ali_am.pdf_offsets, self.am.pdf_offsets):
            raise ValueError("final.alimdl and final.mdl must have the same Gaussians per pdf")
        self._loaded = None
        self._load(self.am)
        self.compiler = _graph.TrainingGraphCompiler(tm, tree, lexicon)
        self.scaled = tm.scaled_log_probs(self.opt.transition_scale, self.opt.self_loop_scale)
        self.frame_shift = float(self.mfcc_options.get("frame_shift_ms", 10.0)) / 1000.0
        self.failed: List[str] = []
        self.failure_reasons: Dict[str, str] = {}
        self.fallback_first_pass: List[str] = []
        self.ctm_failed: List[str] = []          # aligned, but the interval stage raised (alignment kept, ctm None)
        self.transforms: Optional[np.ndarray] = None
        self._mfcc_cache: Dict[tuple, tuple] = {}
        self._mfcc_cache_bytes = 0
        self._mfcc_cache_on = False
        self._intervals = None          # intervals_native.IntervalExtractor, built on first use
        self._worker = None             # one worker thread: the next batch's graphs compile under the current batch
        self._out_pools = None          # pinned staging for the batches' outputs on their way back (two sets)
        self._out_turn = 0

    def _boosted(self, am: DiagGmmModel) -> DiagGmmModel:
        import copy

        if self.opt.boost_silence == 1.0 or not self.silence_phones:
            return am
        from .model import pdfs_of_phones
        am2 = copy.copy(am)
        am2.gconsts = np.array(am.gconsts, dtype=np.float32, copy=True)
        am2.boost_silence(self.opt.boost_silence, pdfs_of_phones(self.tm, self.silence_phones))
        return am2

    def _load(self, am: DiagGmmModel) -> None:
        if self._loaded is not am:
            self.engine.load_gmm(am)
            self._loaded = am

    # ------------------------------------------------------------------ helpers
    def _batches(self, utts: Sequence[CorpusUtterance]) -> List[List[int]]:
        """Length-bucketed batches (BASELINE configs[4]): sort by duration, cut at ``batch_frames``.  Computed once per run
        (``align`` asks three times: graph compilation ahead, the CMVN pass, the alignment passes)."""
        key = (id(utts), len(utts))
        if getattr(self, "_batches_key", None) == key:
            return self._batches_val
        lens = np.fromiter((len(u.pcm) for u in utts), dtype=np.int64, count=len(utts))
        order = np.argsort(lens, kind="stable")
        frames = self.engine.num_frames_array(lens) if hasattr(self.engine, "num_frames_array") else \
            np.array([self.engine.num_frames(int(n)) for n in lens], dtype=np.int64)
        out, cur, total = [], [], 0
        for i, t in zip(order.tolist(), frames[order].tolist()):
            if cur and total + t > self.opt.batch_frames:
                out.append(cur)
                cur, total = [], 0
            cur.append(i)
            total += t
        if cur:
            out.append(cur)
        self._batches_key, self._batches_val = key, out
        return out

    def _mfcc(self, utts: Sequence[CorpusUtterance], idx: Sequence[int]):
        import torch

        key = (id(utts), tuple(idx))          # (the list object of the run in progress; the cache only lives inside align())
        hit = self._mfcc_cache.get(key) if self._mfcc_cache_on else None
        if hit is not None:          # the CMVN pass computed them already (and a second alignment pass asks a third time)
            return hit
        # PCM gathered by host threads straight into pinned staging memory, then one asynchronous H2D copy (mfa_gather_pcm)
        pcm, so = self.engine.gather_pcm([utts[i].pcm for i in idx])
        mfcc, fo = self.engine.mfcc(pcm, so)
        if self.opt.corpus_compression:      # feats.*.ark of MfccFunction: compute_mfccs_for_export(seg, compress=True)
            from . import kaldi_io as _kio
            host = mfcc.cpu().numpy()
            for k in range(len(idx)):
                a, b = int(fo[k]), int(fo[k + 1])
                if b > a:
                    host[a:b] = _kio.compress_round_trip(host[a:b])
            mfcc = torch.from_numpy(host).to(self.engine.device)
        size = mfcc.numel() * mfcc.element_size()
        if self._mfcc_cache_on and self._mfcc_cache_bytes + size <= self.opt.mfcc_cache_bytes:   # 52 KB per 10 s utterance: HBM holds millions
            self._mfcc_cache[key] = (mfcc, fo)
            self._mfcc_cache_bytes += size
        return mfcc, fo

    def _final_features(self, mfcc, fo, rows, cmvn, d_lda, fmllr):
        """CMVN → Δ+ΔΔ | splice+LDA(+fMLLR).  With ``corpus_compression`` the CMVN-applied MFCCs take the second trip
        through the 8-bit codec first (FinalFeatureFunction, MFA/corpus/features.py:323-365), on the host."""
        import torch

        eng = self.engine
        if not self.opt.corpus_compression:
            return eng.features(mfcc, fo, rows, cmvn, lda=d_lda, fmllr=fmllr)
        from . import kaldi_io as _kio
        host = mfcc.cpu().numpy()
        stats = cmvn.cpu().numpy()
        dim = host.shape[1]
        for k in range(len(fo) - 1):
            a, b = int(fo[k]), int(fo[k + 1])
            if b > a:
                st = stats[rows[k]]
                mean = (st[0, :dim] / st[0, dim]).astype(np.float32)       # ApplyCmvn, no variance normalisation
                host[a:b] = _kio.compress_round_trip(host[a:b] - mean)
        return eng.features(torch.from_numpy(host).to(eng.device), fo, rows, None, lda=d_lda, fmllr=fmllr)

    def speaker_cmvn(self, utts: Sequence[CorpusUtterance]) -> Tuple[Dict[str, int], "object"]:
        """calc_cmvn: float64 [n_spk, 2, dim+1] on the device, and the speaker → row map."""
        import torch

        spk_ids = {s: k for k, s in enumerate(dict.fromkeys(u.speaker for u in utts))}
        # per-batch statistics come from the device kernel (float64, fixed order); batches are added up on the host, in
        # batch order — a few hundred bytes per speaker, and no framework arithmetic on the path
        total = np.zeros((len(spk_ids), 2, self.engine.num_ceps + 1), dtype=np.float64)
        pending = []
        for idx in self._batches(utts):
            mfcc, fo = self._mfcc(utts, idx)
            rows = np.array([spk_ids[utts[i].speaker] for i in idx], dtype=np.int32)
            local, inv = np.unique(rows, return_inverse=True)
            pending.append((local, self.engine.cmvn_stats(mfcc, fo, inv.astype(np.int32), len(local))))
        # read back after the last batch is queued: the next batch's PCM is gathered while this one's copy is on the bus
        for local, st in pending:
            total[local] += st.cpu().numpy()
        return spk_ids, torch.from_numpy(total).to(self.engine.device)

    def _decode(self, graphs, feats, fo, max_tokens, bp_tokens):
        """One device call: scores evaluated lazily for the cells live tokens can reach (default), or the dense matrix."""
        eng, o = self.engine, self.opt
        if self.lazy:
            return eng.align_features(graphs, feats, fo, beam=o.beam, retry_beam=o.retry_beam, acoustic_scale=o.acoustic_scale,
                                      max_tokens=max_tokens, bp_tokens_per_frame=bp_tokens)
        ll, ll_off, ll_cols = eng.score(feats, fo, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts,
                                        pdf_first_frame=graphs.pdf_first_frame)
        return eng.align(graphs, ll, ll_off, ll_cols, fo, beam=o.beam, retry_beam=o.retry_beam, acoustic_scale=o.acoustic_scale,
                         max_tokens=max_tokens, bp_tokens_per_frame=bp_tokens)

    # ------------------------------------------------------------------ one alignment pass, batch by batch
    def _compile(self, utts, idx_all, pool):
        """Pure host work of a batch, safe on a worker thread (no device call): transcripts → training graphs by the native
        compiler's threads, straight into the batch's pinned staging buffers.  Runs for batch b + 1 while the main thread packs,
        launches and collects batch b (and, for the first batch, while the CMVN pass gathers PCM)."""
        fsts_all = self.compiler.compile_fsts([utts[i].text for i in idx_all], self.scaled, columns=True, alloc=pool.get)
        return dict(idx_all=list(idx_all), fsts_all=fsts_all, pool=pool)

    def _submit_compile(self, utts, idx_all):
        from concurrent.futures import ThreadPoolExecutor

        if self._worker is None:
            self._worker = ThreadPoolExecutor(1, thread_name_prefix="mfa-graphs")
        pool = self.engine.next_staging()          # (main thread: waits for the copies last started from this pool)
        return self._worker.submit(self._compile, utts, idx_all, pool)

    def _prepare(self, utts, comp):
        """Device layout + score plan of a compiled batch (main thread: starts the batch's H2D copies)."""
        eng = self.engine
        idx_all, fsts_all, pool = comp["idx_all"], comp["fsts_all"], comp["pool"]
        prep = dict(idx_all=idx_all, idx=[], fsts=[], gidx=[], gfsts=[], graphs=None)
        if getattr(fsts_all, "arc_pdf", None) is not None and len(fsts_all) and getattr(fsts_all, "max_degree", None) is not None \
                and fsts_all.min_ilabel > 0 and fsts_all.max_degree <= 64 and fsts_all.min_arcs > 0:
            # a batch straight from the native compiler with nothing for the general decoder in it: taken as it is
            prep["idx"], prep["fsts"] = list(idx_all), fsts_all
        else:
            for i, f in zip(idx_all, fsts_all):
                if f.num_arcs == 0 or f.num_states == 0 or np.any(f.arcs["ilabel"] < 0):
                    self.failure_reasons[utts[i].utt_id] = "empty or malformed training graph"   # this utterance only
                elif eng.needs_general_decoder(f):    # epsilon input arcs / a state with more than 64 arcs
                    prep["gidx"].append(i); prep["gfsts"].append(f)
                else:
                    prep["idx"].append(i); prep["fsts"].append(f)
        if prep["idx"]:
            prep["graphs"] = eng.pack_graphs(prep["fsts"], self.tm, pool=pool if prep["fsts"] is fsts_all else None)
        return prep

    def _launch(self, utts, prep, spk_ids, cmvn, d_lda, fmllr):
        """Device side of a batch: features and the alignment call for the utterances the wavefront-parallel decoder takes —
        enqueued, not waited for."""
        if not prep["idx"]:
            return None
        idx = prep["idx"]
        mfcc, fo = self._mfcc(utts, idx)
        rows = np.array([spk_ids[utts[i].speaker] for i in idx], dtype=np.int32)
        feats = self._final_features(mfcc, fo, rows, cmvn, d_lda, fmllr)
        res = self._decode(prep["graphs"], feats, fo, self.opt.max_tokens, self.opt.bp_tokens_per_frame)
        # outputs start their way back right behind the kernels (asynchronous copies into pinned staging memory, one event):
        # _collect waits for THIS batch only, while the next batch's kernels are already queued behind it
        import torch
        pool = self._out_pools[self._out_turn]
        self._out_turn = (self._out_turn + 1) % len(self._out_pools)
        host = {}
        with torch.cuda.device(self.engine.device):
            for k in ("status", "ali", "words", "n_words", "like"):
                t = res[k]
                view = pool.get("out_" + k, t.numel(), {torch.int32: np.int32, torch.float32: np.float32}[t.dtype])
                torch.from_numpy(view).copy_(t, non_blocking=True)
                host[k] = view
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.engine.device))
        return dict(res=res, feats=feats, fo=fo, rows=rows, host=host, event=ev)

    def _general(self, utts, prep, spk_ids, cmvn, d_lda, fmllr, results, kept, want_feats):
        """FasterDecoder as Kaldi runs it, ProcessNonemitting included (mfa_align_general_batch): the slow, exact path for
        graphs with epsilon input arcs or very wide states."""
        eng, o = self.engine, self.opt
        gidx, gfsts = prep["gidx"], prep["gfsts"]
        mfcc, fo = self._mfcc(utts, gidx)
        rows = np.array([spk_ids[utts[i].speaker] for i in gidx], dtype=np.int32)
        feats = self._final_features(mfcc, fo, rows, cmvn, d_lda, fmllr)
        gg = eng.pack_graphs_general(gfsts, self.tm)
        rg = eng.align_general(gg, feats, fo, beam=o.beam, retry_beam=o.retry_beam, acoustic_scale=o.acoustic_scale,
                               bp_tokens_per_frame=max(o.bp_tokens_per_frame, 512))
        out = _BatchOut(gidx, fo, rg["ali"].cpu().numpy(), rg["words"].cpu().numpy(), rg["n_words"].cpu().numpy(),
                        rg["like"].cpu().numpy(), rg["status"].cpu().numpy())
        for k, i in enumerate(gidx):
            if out.status[k] in (0, 1):
                results[i] = (out, k)
            else:
                self.failure_reasons[utts[i].utt_id] = _lib.status_reason(out.status[k])
        if want_feats:
            import torch
            ali_dev = rg["ali"].clone()
            for k in range(len(gidx)):
                if out.status[k] not in (0, 1):
                    ali_dev[int(fo[k]): int(fo[k + 1])] = 0
            kept.append((gidx, feats, ali_dev, fo, rows))

    def _collect(self, utts, prep, launched, results, kept, want_feats):
        """Results of a launched batch back on the host (this is where the host waits for the device)."""
        import torch

        eng, o = self.engine, self.opt
        idx, fsts = prep["idx"], prep["fsts"]
        res, feats, fo, rows = launched["res"], launched["feats"], launched["fo"], launched["rows"]
        launched["event"].synchronize()
        h = launched["host"]          # (pinned staging views: copied out, the buffers are reused two batches later)
        status, ali, words = h["status"].copy(), h["ali"].copy(), h["words"].copy()
        n_words, like = h["n_words"].copy(), h["like"].copy()
        # Capacity overflows (status 3 tokens / 4 back-pointers) are not alignment failures: those utterances are decoded
        # again on their own with the hard upper bounds (one token per graph state), which cannot overflow — and what still
        # reports a capacity status then (the epsilon closure's pop budget on a pathological epsilon sub-graph) goes to the
        # general decoder, which runs Kaldi's loops as they are.
        over = np.flatnonzero((status == 3) | (status == 4)).tolist()
        if over:
            def take(ks):
                fo_s = np.concatenate([[0], np.cumsum([fo[k + 1] - fo[k] for k in ks])]).astype(np.int64)
                sel = np.concatenate([np.arange(fo[k], fo[k + 1]) for k in ks])
                return fo_s, eng.gather_rows(feats, sel)

            def merge(ks, fo_s, r):
                st_, ali_, w_ = r["status"].cpu().numpy(), r["ali"].cpu().numpy(), r["words"].cpu().numpy()
                nw_, like_ = r["n_words"].cpu().numpy(), r["like"].cpu().numpy()
                for j, k in enumerate(ks):
                    status[k] = st_[j]
                    a, b = int(fo[k]), int(fo[k + 1])
                    ali[a:b] = ali_[fo_s[j]: fo_s[j + 1]]
                    n_words[k] = nw_[j]
                    words[a: a + int(nw_[j])] = w_[fo_s[j]: fo_s[j] + int(nw_[j])]
                    like[k] = like_[j]

            sub = eng.pack_graphs([fsts[k] for k in over], self.tm)
            fo2, f2 = take(over)
            mt, bp = sub.hard_bounds()
            merge(over, fo2, self._decode(sub, f2, fo2, mt, bp))
            still = [k for k in over if status[k] in (3, 4)]
            if still:
                gg = eng.pack_graphs_general([fsts[k] for k in still], self.tm)
                fo3, f3 = take(still)
                merge(still, fo3, eng.align_general(gg, f3, fo3, beam=o.beam, retry_beam=o.retry_beam,
                                                    acoustic_scale=o.acoustic_scale, bp_tokens_per_frame=2 * gg.max_states + 64))
            if want_feats:
                res["ali"] = torch.from_numpy(ali).to(eng.device)
        out = _BatchOut(idx, fo, ali, words, n_words, like, status)
        ok = (status == 0) | (status == 1)
        for k in np.flatnonzero(ok).tolist():
            results[idx[k]] = (out, k)
        for k in np.flatnonzero(~ok).tolist():
            if status[k] == 2:
                self.failure_reasons.setdefault(utts[idx[k]].utt_id, _lib.status_reason(2))
            else:
                self.failure_reasons[utts[idx[k]].utt_id] = _lib.status_reason(status[k])
        if want_feats:
            bad = np.flatnonzero(~ok).tolist()
            if bad:   # frames of utterances that failed must not vote in the fMLLR statistics
                ali_dev = res["ali"].clone()
                for k in bad:
                    ali_dev[int(fo[k]): int(fo[k + 1])] = 0
                res["ali"] = ali_dev
            kept.append((idx, feats, res["ali"], fo, rows))

    def _pass(self, utts, spk_ids, cmvn, fmllr, want_feats=False, first_compile=None):
        """One alignment pass over all batches.  Returns per utterance ``(batch output, index)`` (None where alignment
        failed) and, when asked, what fMLLR estimation needs (features, alignments, frame offsets per batch).  Nothing an
        individual utterance does aborts the pass: unsupported graphs and decoder statuses beyond "failed" are recorded in
        ``failure_reasons`` (the reference logs and continues, MFA/alignment/mixins.py:308-314).

        Software pipeline over batches: the graphs of batch b + 1 compile on a worker thread while the main thread packs,
        launches and collects batch b.  ``first_compile``: the future of batch 0's compilation when the caller started it
        earlier (``_align`` does, under the CMVN pass)."""
        import torch

        eng = self.engine
        if self._out_pools is None:
            from .engine import StagingPool
            self._out_pools = [StagingPool(eng.device), StagingPool(eng.device), StagingPool(eng.device)]
        d_lda = None if self.lda is None else torch.from_numpy(self.lda).to(eng.device)
        results: List[Optional[tuple]] = [None] * len(utts)
        kept: List[tuple] = []
        batches = self._batches(utts)
        fut = first_compile if first_compile is not None else (self._submit_compile(utts, batches[0]) if batches else None)
        in_flight = None                      # (prep, launched) of the batch the device is working on
        for b in range(len(batches)):
            comp = fut.result()
            # the next batch's graphs compile on the worker thread from here on — under this batch's packing and launch and
            # the previous batch's collection (its staging pool was last read two batches ago)
            fut = self._submit_compile(utts, batches[b + 1]) if b + 1 < len(batches) else None
            prep = self._prepare(utts, comp)
            launched = self._launch(utts, prep, spk_ids, cmvn, d_lda, fmllr)
            if in_flight is not None:         # collected while the device runs the batch launched just now
                self._collect(utts, in_flight[0], in_flight[1], results, kept, want_feats)
                in_flight = None
            if prep["gidx"]:
                self._general(utts, prep, spk_ids, cmvn, d_lda, fmllr, results, kept, want_feats)
            if launched is not None:
                in_flight = (prep, launched)
        if in_flight is not None:
            self._collect(utts, in_flight[0], in_flight[1], results, kept, want_feats)
        return results, kept

    # ------------------------------------------------------------------ public
    def align(self, utterances: Sequence[CorpusUtterance], speaker_adapted: bool = False, make_ctm: bool = True,
              previous_transforms: Optional[np.ndarray] = None) -> List[Optional[UtteranceResult]]:
        """The reference's flow (MFA/alignment/base.py:510-539): first pass on speaker-independent features — with the
        alignment model when the acoustic model ships one —; with ``speaker_adapted`` (a SAT model, ``uses_speaker_adaptation``
        in MFA) per-speaker fMLLR is estimated from it and a second pass is run with the transforms on ``am``.
        ``previous_transforms`` [n_spk, D, D+1] (speaker rows in first-appearance order): transforms the features already
        carry in the first pass; the new estimate is composed onto them (CalcFmllrFunction's previous_transform_archive)."""
        import torch

        utts = list(utterances)
        self.failed, self.failure_reasons, self.fallback_first_pass, self.ctm_failed = [], {}, [], []
        self._mfcc_cache, self._mfcc_cache_bytes, self._mfcc_cache_on = {}, 0, True
        try:
            return self._align(utts, speaker_adapted, make_ctm, previous_transforms)
        finally:
            self._mfcc_cache, self._mfcc_cache_bytes, self._mfcc_cache_on = {}, 0, False
            self._batches_key = None

    def _align(self, utts, speaker_adapted, make_ctm, previous_transforms):
        import torch

        batches = self._batches(utts)
        first_compile = self._submit_compile(utts, batches[0]) if batches else None     # graphs compile under the CMVN pass
        spk_ids, cmvn = self.speaker_cmvn(utts)
        first_model = self.ali_am if self.ali_am is not None else self.am
        self._load(first_model)
        prev = None if previous_transforms is None else torch.from_numpy(np.asarray(previous_transforms, dtype=np.float32)).to(self.engine.device)
        results, kept = self._pass(utts, spk_ids, cmvn, prev, want_feats=speaker_adapted, first_compile=first_compile)
        self.transforms = None if previous_transforms is None else np.asarray(previous_transforms, dtype=np.float32)
        if speaker_adapted:
            if self.lda is None:
                raise NotImplementedError("speaker adaptation needs the LDA feature path")
            D = self.lda.shape[0]
            beta = np.zeros(len(spk_ids)); K = np.zeros((len(spk_ids), D, D + 1)); G = np.zeros((len(spk_ids), D, D + 1, D + 1))
            two_model = self.am if self.ali_am is not None else None
            for idx, feats, ali, fo, rows in kept:
                ids, b, k, g = fmllr_statistics(self.engine, feats, fo, ali, self.tm, rows, self.silence_phones,
                                                self.opt.silence_weight, stats_model=two_model)
                beta[ids] += b; K[ids] += k; G[ids] += g
            W = np.tile(np.eye(D, D + 1, dtype=np.float32), (len(spk_ids), 1, 1))
            for s in range(len(spk_ids)):
                W[s], _impr = _fmllr.compute_fmllr(beta[s], K[s], G[s], min_count=self.opt.fmllr_min_count)
                if previous_transforms is not None:
                    W[s] = _fmllr.compose_transforms(W[s], previous_transforms[s])
            self.transforms = W
            self._load(self.am)
            first = results
            self.failure_reasons = {}
            results, _ = self._pass(utts, spk_ids, cmvn, torch.from_numpy(W).to(self.engine.device))
            if self.ali_am is not None:
                # an utterance the second pass lost keeps its first-pass (alignment-model) result:
                # ali_first_pass / words_first_pass / likelihoods_first_pass, MFA/alignment/multiprocessing.py:1784-1860
                for i, (r2, r1) in enumerate(zip(results, first)):
                    if r2 is None and r1 is not None:
                        results[i] = r1
                        self.fallback_first_pass.append(utts[i].utt_id)
                        self.failure_reasons.pop(utts[i].utt_id, None)
        return self._results(utts, results, make_ctm)

    def _extractor(self):
        if self._intervals is None:
            from . import intervals_native
            self._intervals = intervals_native.IntervalExtractor(self.tm, self.lexicon, self.frame_shift)
        return self._intervals

    def _results(self, utts, results, make_ctm) -> List[Optional[UtteranceResult]]:
        """Per-utterance results from the batch outputs.  With ``make_ctm`` the interval stage — generate_ctm →
        phones_to_pronunciations → update_utterance_boundaries → fix_unk_words, MFA/alignment/multiprocessing.py:1733-1751 —
        runs once per batch over the arrays (intervals_native); the ``HierarchicalCtm`` objects are built when a caller asks
        an ``UtteranceResult`` for its ``ctm``.  Per utterance, as the reference's extraction loop: an alignment the stage
        cannot take concerns this utterance only (:1739-1770 catches, logs and continues)."""
        sr = float(self.mfcc_options.get("sample_frequency", 16000.0))
        if make_ctm:
            ex = self._extractor()
            for bo in {id(r[0]): r[0] for r in results if r is not None}.values():
                if bo.intervals is None:
                    bo.intervals = ex.extract(bo.frame_off, bo.ali, bo.words, bo.n_words, bo.status)
        out: List[Optional[UtteranceResult]] = []
        for u, r in zip(utts, results):
            if r is None:
                self.failed.append(u.utt_id)
                self.failure_reasons.setdefault(u.utt_id, "no alignment")
                out.append(None)
                continue
            bo, k = r
            a, b = int(bo.frame_off[k]), int(bo.frame_off[k + 1])
            ur = UtteranceResult(u.utt_id, u.speaker, bo.ali[a:b], bo.words[a: a + int(bo.n_words[k])], float(bo.like[k]), b - a)
            if make_ctm:
                end = u.begin + len(u.pcm) / sr
                if bo.intervals.err[k] == 0:
                    ur._lazy = (bo.intervals, k, u.text, u.begin, end)
                else:
                    try:                    # the Python specification on this utterance: it raises the specific error
                        ur.ctm = bo.intervals.ctm(k, text=u.text, begin=u.begin, end=end, likelihood=ur.per_frame_likelihood)
                    except Exception as e:  # noqa: BLE001
                        self.failure_reasons[u.utt_id] = f"interval extraction failed: {e}"
                        self.ctm_failed.append(u.utt_id)
            out.append(ur)
        return out

    def export_textgrids(self, utterances: Sequence[CorpusUtterance], results: Sequence[Optional[UtteranceResult]], output_directory,
                         output_format: str = "long_textgrid", cleanup_silence: bool = True) -> List[Path]:
        """One file per sound file, one (words, phones) tier pair per speaker (export_textgrid, MFA/textgrid.py:463-572).

        Files whose utterances still carry their interval arrays (the results of ``align``) are written from those arrays
        by the native writer — byte for byte what ``ctm.export_textgrid`` writes (tests/test_intervals_native_cpu.py);
        anything else (results whose ``ctm`` a caller set or changed, a file the native writer declines) goes through the
        Python objects."""
        out_dir = Path(output_directory)
        out_dir.mkdir(parents=True, exist_ok=True)
        sr = float(self.mfcc_options.get("sample_frequency", 16000.0))
        ext = {"long_textgrid": ".TextGrid", "short_textgrid": ".TextGrid", "json": ".json", "csv": ".csv"}[output_format]
        per_file: Dict[str, dict] = {}
        for n, (u, r) in enumerate(zip(utterances, results)):
            if r is None or not r.has_intervals:
                continue
            name = u.file_name or u.utt_id
            f = per_file.get(name)
            if f is None:
                f = per_file[name] = dict(name=name, duration=0.0, speakers={}, native=True)
            end = u.begin + len(u.pcm) / sr
            f["duration"] = max(f["duration"], u.file_duration or end)
            f["speakers"].setdefault(u.speaker, []).append(n)
            if r._lazy is None:
                f["native"] = False
        written: List[Path] = []
        slow = [f for f in per_file.values() if not f["native"]]
        fast = [f for f in per_file.values() if f["native"]]
        if fast:
            from . import intervals_native
            # the interval arrays of the batches involved, laid end to end: utterance (batch, k) becomes base[batch] + k
            batches, base = {}, {}
            for f in fast:
                for ns in f["speakers"].values():
                    for n in ns:
                        b = results[n]._lazy[0]
                        if id(b) not in batches:
                            base[id(b)] = sum(x.n_utt for x in batches.values())
                            batches[id(b)] = b
            merged = intervals_native.IntervalBatch.concat(list(batches.values()))
            total = merged.n_utt
            ub_l, ue_l = [0.0] * total, [0.0] * total          # (plain lists: per-element numpy stores cost more than the loop)
            texts: List[Optional[str]] = [None] * total
            files = []
            for f in fast:
                spk = []
                for name, ns in f["speakers"].items():
                    ids = []
                    for n in ns:
                        b, k, text, begin, end = results[n]._lazy
                        m = base[id(b)] + k
                        ub_l[m] = begin; ue_l[m] = end; texts[m] = text
                        ids.append(m)
                    spk.append((name, ids))
                files.append(dict(duration=f["duration"], speakers=spk))
            ub, ue = np.asarray(ub_l, dtype=np.float64), np.asarray(ue_l, dtype=np.float64)
            paths = [str(out_dir / (f["name"] + ext)) for f in fast]
            _none, codes = self._extractor().write_files(merged, files, ub, ue, merged.relabels(texts), output_format, cleanup_silence,
                                                          paths=paths)          # text AND files by the library's threads
            for f, path, code in zip(fast, paths, codes):
                if code == 0:
                    written.append(Path(path))
                elif code == 1:
                    slow.append(f)          # the Python writer raises the reference's error for this file
        sil = self.lexicon.silence_word
        for f in slow:
            speakers = {}
            for name, ns in f["speakers"].items():
                tiers = speakers.setdefault(name, {"words": [], "phones": []})
                for n in ns:
                    for w in results[n].ctm.word_intervals:
                        if cleanup_silence and w.label == sil:
                            continue
                        tiers["words"].append(_ctm.CtmInterval(w.begin, w.end, w.label))
                        tiers["phones"].extend(w.phones)
                tiers["words"].sort(); tiers["phones"].sort()
            path = out_dir / (f["name"] + ext)
            _ctm.export_textgrid(speakers, path, f["duration"], self.frame_shift, output_format)
            if path.exists():
                written.append(path)
        order = {name: k for k, name in enumerate(per_file)}
        written.sort(key=lambda p_: order.get(p_.name[: -len(ext)], 0))
        return written


def align_sharded(aligner_factory, utterances: Sequence[CorpusUtterance], rank: int, world_size: int, **kw):
    """One process per GPU: this rank aligns the speakers ``sharding.assign_speakers`` gives it (weights = audio seconds)
    and every rank receives all results (host-side object gather; no collective on the data path)."""
    spk_index = {s: k for k, s in enumerate(dict.fromkeys(u.speaker for u in utterances))}
    rank_of = sharding.assign_speakers([spk_index[u.speaker] for u in utterances], world_size,
                                       weights=[len(u.pcm) for u in utterances])
    mine = sharding.local_indices(rank_of, rank)
    aligner = aligner_factory()
    local = aligner.align([utterances[i] for i in mine], **kw)
    gathered = sharding.gather_results({int(i): r for i, r in zip(mine, local)}, world_size)
    return [gathered.get(i) for i in range(len(utterances))]
