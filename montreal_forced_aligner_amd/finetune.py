"""Boundary fine-tuning and phone-confidence passes: re-uses of the MFCC / feature / scoring / Viterbi
kernels on tiny windows and on all-pdf score matrices.

* ``fine_tune_boundaries`` mirrors FineTuneFunction._run (MFA/alignment/multiprocessing.py:1127-1350) and the interval
  repair loop at its end: every phone boundary is re-aligned inside a ±15 ms window of features computed with a 1 ms frame
  shift, against the two-phone graph "previous phone group → phone group" (the phone-group lexicon of
  MFA/dictionary/multispeaker.py:2078-2111 is a single-state transducer phone → group, i.e. a free choice among the
  group's phones, no silence insertion).  The reference does this one boundary at a time in a worker process; here all
  boundaries of a batch of utterances are one ragged device batch (≈120 windows per 10 s utterance).
* ``phone_confidence`` mirrors PhoneConfidenceFunction._run (:1353-1447): all-pdf log-likelihoods
  (``gmm_compute_likes``), phone scores as count-weighted sums over each phone's pdfs, and per interval the mean margin
  by which the best phone beats the aligned phone ("phone_goodness").
"""
from __future__ import annotations

import collections
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import ctm as _ctm
from . import graph as _graph

NEW_FRAME_SHIFT = 0.001          # FineTuneFunction.new_frame_shift_seconds
FEATURE_PADDING_FACTOR = 3       # FineTuneFunction.feature_padding_factor


@dataclass
class _Window:
    utt: int
    index: int                   # interval index inside the utterance (>= 1)
    feature_begin: float         # start of the audio cut the 1 ms features are computed on
    feature_end: float
    begin_offset: float          # decoded rows start this far into the cut
    end_offset: float
    prev_phone: int
    phone: int


def plan_windows(intervals: Sequence[Sequence[_ctm.CtmInterval]], utt_ends: Sequence[float], frame_shift: float = 0.01):
    """The time arithmetic of FineTuneFunction._run (:1254-1275), per boundary between consecutive phone intervals."""
    out: List[_Window] = []
    pad = round(frame_shift * 1.5, 3)
    for u, ivs in enumerate(intervals):
        for i in range(1, len(ivs)):
            iv = ivs[i]
            segment_begin = max(round(iv.begin - pad, 4), 0)
            feature_begin = max(round(iv.begin - pad * FEATURE_PADDING_FACTOR, 4), 0)
            segment_end = round(min(iv.begin + pad, iv.end), 3)
            feature_end = min(round(iv.begin + pad * FEATURE_PADDING_FACTOR, 4), utt_ends[u])
            out.append(_Window(u, i, feature_begin, feature_end, round(segment_begin - feature_begin, 4),
                               round(segment_end - feature_begin, 4), int(ivs[i - 1].symbol), int(iv.symbol)))
    return out


def two_phone_graph(compiler: _graph.TrainingGraphCompiler, first: Sequence[int], second: Sequence[int]):
    """Training graph of "any phone of ``first``, then any phone of ``second``" (phone ids), word labels 1 and 2."""
    pg = _graph.PhoneGraph()
    n0, n1, n2 = pg.add_node(), pg.add_node(), pg.add_node()
    pg.start = n0
    for p in first:
        pg.add_arc(n0, n1, int(p), 1, 0.0)
    for p in second:
        pg.add_arc(n1, n2, int(p), 2, 0.0)
    pg.final[n2] = 0.0
    return compiler.compile_phone_graph(pg)


def repair_intervals(mapping: List[dict]) -> Tuple[List[dict], List[object]]:
    """The loop at the end of FineTuneFunction._run (:1327-1349): close gaps (end := next begin), delete intervals that
    became empty, repeat until contiguous."""
    deletions: List[object] = []
    while True:
        for i in range(len(mapping) - 1):
            if mapping[i]["end"] != mapping[i + 1]["begin"]:
                mapping[i]["end"] = mapping[i + 1]["begin"]
        new_del = [x["id"] for x in mapping if x["begin"] >= x["end"]]
        mapping = [x for x in mapping if x["id"] not in new_del]
        deletions.extend(new_del)
        if not new_del and all(mapping[i]["end"] == mapping[i + 1]["begin"] for i in range(len(mapping) - 1)):
            break
    return mapping, deletions


def fine_tune_boundaries(aligner, compiler: _graph.TrainingGraphCompiler, pcm: Sequence[np.ndarray],
                         intervals: Sequence[Sequence[_ctm.CtmInterval]], utt2spk: Optional[Sequence[int]] = None,
                         cmvn=None, lda=None, fmllr=None, phone_group: Optional[Callable[[int], Sequence[int]]] = None,
                         frame_shift: float = 0.01, sample_rate: int = 16000, mfcc_options: Optional[dict] = None,
                         splice_context: int = 3):
    """Returns (new interval lists, deletions per utterance).

    ``aligner``: a kalpy_api.GmmAligner (its beams and transition scales are used; the acoustic scale is 1.0, falling
    back to 0.1 for a window that fails, as the reference does).  ``pcm``: one int16 array per utterance (time 0 = the
    utterance's begin).  ``intervals``: per utterance, phone CtmIntervals with ``symbol`` = phone id, sorted by begin.
    ``cmvn``: device tensor [n_spk, 2, dim+1] (the speakers' 10 ms statistics) or None; ``utt2spk`` indexes it."""
    import torch

    eng = aligner._engine()
    tm = aligner.transition_model
    group_of = phone_group or (lambda p: [p])
    utt_ends = [len(x) / sample_rate for x in pcm]
    windows = plan_windows(intervals, utt_ends, frame_shift)
    result = [[{"id": (u, 0), "begin": ivs[0].begin, "end": ivs[0].end, "label": int(ivs[0].symbol)}] if ivs else []
              for u, ivs in enumerate(intervals)]
    if not windows:
        return [list(ivs) for ivs in intervals], [[] for _ in intervals]
    # ---- 1 ms features of every window (one batch)
    cuts = []
    for w in windows:
        a, b = int(round(w.feature_begin * sample_rate)), int(round(w.feature_end * sample_rate))
        cuts.append(np.ascontiguousarray(pcm[w.utt][a:b], dtype=np.int16))
    sample_off = np.concatenate([[0], np.cumsum([len(c) for c in cuts])]).astype(np.int64)
    saved = dict(mfcc_options or {})
    eng.configure_mfcc(**{**saved, "frame_shift_ms": NEW_FRAME_SHIFT * 1000.0})
    try:
        d_pcm = torch.from_numpy(np.concatenate(cuts)).to(eng.device)
        mfcc, frame_off = eng.mfcc(d_pcm, sample_off)
    finally:
        eng.configure_mfcc(**saved)
    spk = np.asarray([0 if utt2spk is None else utt2spk[w.utt] for w in windows], dtype=np.int32)
    feats = eng.features(mfcc, frame_off, spk if cmvn is not None or fmllr is not None else None, cmvn, lda, fmllr, splice_context)
    # ---- rows [begin_offset, end_offset) of every window (FloatSubMatrix in the reference, :1300-1304)
    rows, new_off = [], [0]
    for k, w in enumerate(windows):
        T = int(frame_off[k + 1] - frame_off[k])
        a = min(max(int(round(w.begin_offset * 1000)), 0), T)
        b = min(max(int(round(w.end_offset * 1000)), a), T)
        rows.append(np.arange(frame_off[k] + a, frame_off[k] + b, dtype=np.int64))
        new_off.append(new_off[-1] + (b - a))
    new_off = np.asarray(new_off, dtype=np.int64)
    sub = eng.gather_rows(feats, np.concatenate(rows)) if new_off[-1] else feats[:0]   # device-to-device copies of the windows
    # ---- two-phone graphs (cached per group pair)
    cache: Dict[Tuple[Tuple[int, ...], Tuple[int, ...]], object] = {}
    fsts = []
    for w in windows:
        key = (tuple(group_of(w.prev_phone)), tuple(group_of(w.phone)))
        if key not in cache:
            cache[key] = _graph.add_transition_probs(two_phone_graph(compiler, *key), aligner._scaled)
        fsts.append(cache[key])
    ok = [k for k in range(len(windows)) if new_off[k + 1] > new_off[k]]

    def run(sel: List[int], acoustic_scale: float):
        if not sel:
            return {}
        graphs = eng.pack_graphs([fsts[k] for k in sel], tm)
        fo = np.concatenate([[0], np.cumsum([new_off[k + 1] - new_off[k] for k in sel])]).astype(np.int64)
        idx = np.concatenate([np.arange(new_off[k], new_off[k + 1]) for k in sel])
        x = eng.gather_rows(sub, idx)
        ll, ll_off, ll_cols = eng.score(x, fo, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts,
                                        pdf_first_frame=graphs.pdf_first_frame)
        res = eng.align(graphs, ll, ll_off, ll_cols, fo, beam=aligner.beam, retry_beam=aligner.retry_beam,
                        acoustic_scale=acoustic_scale)
        status, ali = res["status"].cpu().numpy(), res["ali"].cpu().numpy()
        return {k: ali[fo[i]: fo[i + 1]] for i, k in enumerate(sel) if status[i] in (0, 1)}

    done = run(ok, 1.0)
    done.update(run([k for k in ok if k not in done], 0.1))   # :1306-1309
    for k, w in enumerate(windows):
        iv = intervals[w.utt][w.index]
        begin, label = iv.begin, int(iv.symbol)
        if k in done:
            ctm = _ctm.generate_ctm(done[k], tm, None, NEW_FRAME_SHIFT)
            if len(ctm) > 1:
                begin = round(ctm[1].begin + w.feature_begin + w.begin_offset, 4)
                label = int(ctm[1].symbol)
        result[w.utt].append({"id": (w.utt, w.index), "begin": begin, "end": iv.end, "label": label})
    out_iv, out_del = [], []
    for u, mapping in enumerate(result):
        mapping, deleted = repair_intervals(mapping) if mapping else ([], [])
        labels = {int(iv.symbol): iv.label for iv in intervals[u]}
        out_iv.append([_ctm.CtmInterval(m["begin"], m["end"], labels.get(m["label"], m["label"]), m["label"]) for m in mapping])
        out_del.append([i for (_u, i) in deleted])
    return out_iv, out_del


def phone_pdf_weights(phone_pdf_counts: Dict[str, Dict[str, float]], strip_position: Callable[[str], str] = lambda p: p):
    """phone_pdf_counts.json → (sorted phone names, {phone: (pdf ids, weights)}) with weights = count / phone total
    (PhoneConfidenceFunction._run :1380-1393)."""
    acc: Dict[str, collections.Counter] = collections.defaultdict(collections.Counter)
    for phone, pdf_counts in phone_pdf_counts.items():
        base = strip_position(phone)
        for pdf, count in pdf_counts.items():
            acc[base][int(pdf)] += count
    names = sorted(acc)
    table = {}
    for p in names:
        total = sum(acc[p].values())
        table[p] = (np.fromiter(acc[p].keys(), dtype=np.int64), np.asarray([c / total for c in acc[p].values()], dtype=np.float64))
    return names, table


def phone_confidence(engine, feats, frame_off: np.ndarray, phone_pdf_counts: Dict[str, Dict[str, float]],
                     intervals: Sequence[Sequence[_ctm.CtmInterval]], utt_begins: Optional[Sequence[float]] = None,
                     strip_position: Callable[[str], str] = lambda p: p, silence_label: str = "sil"):
    """Per utterance, a list of (interval index, phone_goodness) for the non-silence intervals.

    ``feats``: device tensor [ΣT, D] of final features; the engine's loaded model is scored on ALL its pdfs
    (gmm_compute_likes, MFA/alignment/multiprocessing.py:1415)."""
    import torch

    names, table = phone_pdf_weights(phone_pdf_counts, strip_position)
    index = {p: i for i, p in enumerate(names)}
    num_pdfs = engine.gmm.num_pdfs
    pl, counts = engine.sort_pdf_list(np.arange(num_pdfs, dtype=np.int32))
    n_utt = len(frame_off) - 1
    pdf_off = (np.arange(n_utt + 1, dtype=np.int64) * num_pdfs)
    ll, ll_off, _ = engine.score(feats, frame_off, engine._dev(np.tile(pl, n_utt)), pdf_off,
                                 engine._dev(np.tile(counts, (n_utt, 1)).astype(np.int32)))
    # phone_likes = likes[:, pdfs(p)] · weights(p), as one [P, n_phones] matrix in list-column order (float64 like numpy's dot)
    col_of = np.empty(num_pdfs, dtype=np.int64)
    col_of[pl] = np.arange(num_pdfs)
    W = np.zeros((num_pdfs, len(names)), dtype=np.float64)
    for p, (pdfs, w) in table.items():
        np.add.at(W[:, index[p]], col_of[pdfs], w)
    ll_host = ll.cpu().numpy()     # the phone scores are host arithmetic, as in the reference (numpy, :1415-1440)
    out = []
    for u in range(n_utt):
        T = int(frame_off[u + 1] - frame_off[u])
        likes = ll_host[ll_off[u]: ll_off[u + 1]].reshape(T, num_pdfs).astype(np.float64)
        phone_likes = likes @ W
        top = phone_likes.argmax(axis=1)
        begin0 = 0.0 if utt_begins is None else utt_begins[u]
        res = []
        for i, iv in enumerate(intervals[u]):
            name = strip_position(str(iv.label))
            if name == silence_label or name not in index:
                continue
            fb = int(((iv.begin - begin0) * 1000) / 10)
            fe = int(((iv.end - begin0) * 1000) / 10)
            if fb == fe:
                fe += 1
            fe = min(fe, T)
            scores = []
            for t in range(fb, fe):
                scores.append(0.0 if names[top[t]] == name else float(phone_likes[t, top[t]] - phone_likes[t, index[name]]))
            if scores:
                res.append((i, float(np.mean(scores))))
        out.append(res)
    return out
