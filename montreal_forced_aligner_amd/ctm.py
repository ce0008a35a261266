"""Alignment → phone intervals → word intervals → TextGrid ("next" N2).

Reference boundary:
  * ``Alignment.generate_ctm(transition_model, phone_table, frame_shift)`` (MFA/alignment/multiprocessing.py:1734;
    MFA/online/alignment.py:113-117) = Kaldi SplitToPhones with reordered graphs;
  * ``LexiconCompiler.phones_to_pronunciations(words, intervals, transcription=False, text=…)`` → ``HierarchicalCtm``
    (MFA/alignment/multiprocessing.py:1741-1751; MFA/online/alignment.py:118-122);
  * ``export_textgrid`` / ``Textgrid.save`` (MFA/textgrid.py:463-572, :50-161) and ``CtmInterval.to_tg_interval``
    (MFA/data.py:2062-2080): rounding to 6 decimals, last interval snapped to the file end when closer than two frames,
    empty intervals inserted for gaps > 1 ms.
Host-side (integer/string work, ≈120 intervals per utterance); nothing here touches the GPU.
"""
from __future__ import annotations

import csv
import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np


class CtmError(ValueError):
    pass


@dataclass(slots=True)
class CtmInterval:
    """Mirror of MFA/data.py:2017-2080."""

    begin: float
    end: float
    label: object
    symbol: int = 0
    confidence: Optional[float] = None

    def __lt__(self, other):
        return self.begin < other.begin

    def to_tg_interval(self, file_duration: Optional[float] = None):
        if self.end < -1 or self.begin == 1000000:
            raise CtmError(self)
        end = round(self.end, 6)
        begin = round(self.begin, 6)
        if file_duration is not None and end > file_duration:
            end = round(file_duration, 6)
        if begin >= end:
            raise CtmError(self)
        return (round(self.begin, 6), end, self.label)


@dataclass
class WordCtmInterval:
    label: str
    word_id: int
    pronunciation: str
    phones: List[CtmInterval] = field(default_factory=list)

    @property
    def begin(self) -> float:
        return self.phones[0].begin if self.phones else -1.0

    @property
    def end(self) -> float:
        return self.phones[-1].end if self.phones else -1.0


@dataclass
class HierarchicalCtm:
    word_intervals: List[WordCtmInterval]
    text: Optional[str] = None
    likelihood: Optional[float] = None

    @property
    def phone_intervals(self) -> List[CtmInterval]:
        return [p for w in self.word_intervals for p in w.phones]

    def update_utterance_boundaries(self, begin: float, end: Optional[float] = None) -> None:
        """Shift by the utterance's begin time; clip the last phone to the utterance end."""
        if begin:
            for w in self.word_intervals:
                for p in w.phones:
                    p.begin += begin
                    p.end += begin
        if end is not None and self.word_intervals and self.word_intervals[-1].phones:
            last = self.word_intervals[-1].phones[-1]
            if last.end > end:
                last.end = end

    def export_textgrid(self, file_name, file_duration: Optional[float] = None, output_format: str = "long_textgrid",
                        frame_shift: float = 0.01, cleanup_silence: bool = True, silence_words=("<eps>",)) -> None:
        words = [CtmInterval(w.begin, w.end, w.label) for w in self.word_intervals
                 if not (cleanup_silence and w.label in silence_words)]
        phones = [p for w in self.word_intervals for p in w.phones if not (cleanup_silence and w.label in silence_words)]
        duration = file_duration if file_duration is not None else (phones[-1].end if phones else 0.0)
        export_textgrid({"speaker": {"words": words, "phones": phones}}, Path(file_name), duration, frame_shift, output_format)


def split_to_phones(alignment: Sequence[int], tm) -> List[tuple]:
    """Kaldi SplitToPhones (reordered): [(first frame, n frames, phone id)].  Raises if the alignment is not a sequence
    of complete phones.

    Array form of ``_split_to_phones_loop`` (the rule as Kaldi writes it, one frame at a time): a phone ends with its final
    transition plus the self-loops that follow it, i.e. right before the next forward transition.  Anything irregular is
    handed to the loop, which raises the specific error."""
    ali = np.asarray(alignment, dtype=np.int64)
    T = ali.shape[0]
    if T == 0:
        return []
    if ali.min() < 1 or ali.max() >= tm.is_final.shape[0]:
        return _split_to_phones_loop(ali, tm)
    fin = np.asarray(tm.is_final)[ali].astype(bool)
    loop = np.asarray(tm.is_self_loop)[ali].astype(bool)
    st = np.asarray(tm.id2state)[ali]
    ph = np.asarray(tm.id2phone)[ali]
    fwd = np.flatnonzero(~loop)                        # forward transitions
    fpos = np.flatnonzero(fin)                         # final transitions (forward ones, by construction of the model)
    if fpos.shape[0] == 0 or np.any(loop[fpos]):
        return _split_to_phones_loop(ali, tm)
    # end of every phone: the frame before the forward transition that follows its final transition
    nxt = np.searchsorted(fwd, fpos, side="right")
    ends = np.where(nxt < fwd.shape[0], fwd[np.minimum(nxt, fwd.shape[0] - 1)] - 1, T - 1)
    starts = np.concatenate([[0], ends[:-1] + 1])
    ok = ends[-1] == T - 1
    # frames inside a final run (final transition + its self-loops) — the loop form skips its checks there
    mark = np.zeros(T + 1, dtype=np.int32)
    np.add.at(mark, fpos, 1)
    np.add.at(mark, ends + 1, -1)
    in_run = np.cumsum(mark[:T]) > 0
    same_state = st[1:] == st[:-1]
    # (1) self-loops after a final transition stay in its transition-state
    ok = ok and not np.any(in_run[1:] & in_run[:-1] & ~fin[1:] & ~same_state)
    # (3) outside those runs the phone only changes across a final transition
    ok = ok and not np.any(~in_run[:-1] & ~same_state & (ph[1:] != ph[:-1]))
    if not ok:
        return _split_to_phones_loop(ali, tm)
    phones = ph[starts]
    return list(zip(starts.tolist(), (ends + 1 - starts).tolist(), phones.tolist()))


def _split_to_phones_loop(alignment: Sequence[int], tm) -> List[tuple]:
    """SplitToPhones frame by frame (the specification of ``split_to_phones``; raises CtmError on irregular input)."""
    ali = np.asarray(alignment, dtype=np.int64)
    T = ali.shape[0]
    out = []
    cur = 0
    i = 0
    while i < T:
        tid = ali[i]
        if tm.is_final[tid]:
            while i + 1 < T and tm.is_self_loop[ali[i + 1]]:
                if tm.id2state[ali[i]] != tm.id2state[ali[i + 1]]:
                    raise CtmError("self-loop after a final transition belongs to another transition-state")
                i += 1
            out.append((cur, i + 1 - cur, int(tm.id2phone[ali[cur]])))
            cur = i + 1
        elif i + 1 == T:
            raise CtmError("alignment ends inside a phone")
        else:
            a, b = tm.id2state[ali[i]], tm.id2state[ali[i + 1]]
            if a != b and tm.id2phone[ali[i]] != tm.id2phone[ali[i + 1]]:
                raise CtmError("phone changed without a final transition")
        i += 1
    return out


def _frame_times(frames: np.ndarray, frame_shift: float) -> List[float]:
    """``[round(f * frame_shift, 6) for f in frames]``.  When the shift is a whole number of microseconds (it is: 10 ms) the
    rounded value is the decimal f·µs/10^6 exactly, and the double nearest to it is one correctly rounded division — the
    same double Python's round returns — computed for all frames at once."""
    us = round(frame_shift * 1e6)
    if us > 0 and abs(frame_shift * 1e6 - us) < 1e-6 and (frames.size == 0 or int(frames.max()) * us < 2 ** 52):
        return ((frames.astype(np.int64) * us).astype(np.float64) / 1e6).tolist()
    return [round(int(f) * frame_shift, 6) for f in frames]


def generate_ctm(alignment: Sequence[int], tm, phone_table, frame_shift: float = 0.01) -> List[CtmInterval]:
    segs = split_to_phones(alignment, tm)
    if not segs:
        return []
    arr = np.asarray(segs, dtype=np.int64)
    begins, ends = _frame_times(arr[:, 0], frame_shift), _frame_times(arr[:, 0] + arr[:, 1], frame_shift)
    phones = arr[:, 2].tolist()
    find = phone_table.find if phone_table is not None else (lambda p: p)
    names = {p: find(p) for p in set(phones)}
    return [CtmInterval(b, e, names[p], p) for b, e, p in zip(begins, ends, phones)]


def _position_labels(phones: Sequence[str]) -> List[str]:
    """Word-position suffixes as MFA's position-dependent phone sets carry them: a one-phone word is ``_S``; otherwise the
    first phone ``_B``, the last ``_E``, the rest ``_I`` (MFA/dictionary/mixins.py position handling; cf. the expected
    lexicon in tests/data/dictionaries/expected/lexicon.text.fst)."""
    if len(phones) == 1:
        return [phones[0] + "_S"]
    return [phones[0] + "_B"] + [p + "_I" for p in phones[1:-1]] + [phones[-1] + "_E"]


def phones_to_pronunciations(lexicon, word_ids: Sequence[int], intervals: Sequence[CtmInterval], transcription: bool = False,
                             text: Optional[str] = None) -> HierarchicalCtm:
    """Group phone intervals into the aligned word sequence (MFA/alignment/multiprocessing.py:1741-1747).

    The reference composes the phone string with the align lexicon FST (kalpy LexiconCompiler.phones_to_pronunciations);
    here the same constraint — the word-id sequence the decoder emitted, each word spelt by one of its pronunciation
    variants, optional-silence intervals in between — is solved as a search over (word index, interval index) with
    backtracking, so a variant that is a prefix of another cannot steal the following word's phones.  With
    position-dependent phones the aligned labels carry ``_B/_I/_E/_S`` and a variant must match them exactly (the word-end
    evidence decides between "a b" + "c" and "a" + "b c").  Optional-silence
    intervals between words become ``silence_word`` entries, as MFA stores them."""
    sil = lexicon.silence_phone
    pos_dep = bool(lexicon.position_dependent_phones)
    labels = [str(iv.label) for iv in intervals]
    n = len(intervals)
    sil_id = lexicon.word_table.find(lexicon.silence_word)
    words = [lexicon.word_table.find(int(w)) for w in word_ids]

    def variants(word: str) -> List[List[str]]:
        prons = lexicon.word_pronunciations(word) if word != lexicon.oov_word else lexicon.word_pronunciations("\0oov\0")
        seen, out = set(), []
        for p in sorted(prons, key=lambda p: -len(p.pronunciation.split())):   # longest first: the greedy choice when unambiguous
            ph = tuple(p.pronunciation.split())
            if ph not in seen:
                seen.add(ph)
                out.append(list(ph))
        return out

    def expected(ph: List[str]) -> List[str]:
        # every phone of a WORD carries its position, the oov word's spn included (spn_S — LexiconCompiler.phone_ids and
        # tests/data/dictionaries/expected/lexicon.text.fst); only the optional silence between words is a bare phone
        return _position_labels(ph) if pos_dep else ph

    var_cache = [[(ph, expected(ph)) for ph in variants(w)] for w in words]
    dead = set()          # (word index, interval index) states known to have no completion

    def solve(i: int, k: int):
        """Segmentation of intervals[k:] into words[i:] (+ inter-word silence), or None."""
        if (i, k) in dead:
            return None
        if i == len(words):
            if all(labels[j] == sil for j in range(k, n)):
                return [("sil", j) for j in range(k, n)]
            dead.add((i, k))
            return None
        # a word first (its variants may themselves start with the silence phone), then "this interval is inter-word silence"
        for ph, exp in var_cache[i]:
            m = len(exp)
            if k + m <= n and labels[k: k + m] == exp:
                rest = solve(i + 1, k + m)
                if rest is not None:
                    return [("word", i, k, ph)] + rest
        if k < n and labels[k] == sil:
            rest = solve(i, k + 1)
            if rest is not None:
                return [("sil", k)] + rest
        dead.add((i, k))
        return None

    import sys
    limit = sys.getrecursionlimit()
    if limit < 4 * (n + len(words)) + 100:
        sys.setrecursionlimit(4 * (n + len(words)) + 100)
    try:
        plan = solve(0, 0)
    finally:
        sys.setrecursionlimit(limit)
    if plan is None:
        raise CtmError("the aligned phones cannot be spelt by any combination of the aligned words' pronunciations")
    out: List[WordCtmInterval] = []
    for step in plan:
        if step[0] == "sil":
            out.append(WordCtmInterval(lexicon.silence_word, sil_id, sil, [intervals[step[1]]]))
        else:
            _, i, k, ph = step
            out.append(WordCtmInterval(words[i], int(word_ids[i]), " ".join(ph), list(intervals[k: k + len(ph)])))
    return HierarchicalCtm(out, text=text)


def fix_unk_words(ref: Sequence[str], test: Sequence[WordCtmInterval], lexicon) -> List[WordCtmInterval]:
    """Give out-of-vocabulary word intervals the spelling they have in the transcript (MFA/helper.py:772-833, called at
    MFA/alignment/multiprocessing.py:1749-1751): global alignment of the transcript's words against the aligned word
    intervals with the reference's scores — identical labels or identical word ids 0, anything against the silence word
    -10, other mismatches -2, gaps -2 per position — then every aligned ``<unk>`` interval takes the transcript word it
    is paired with; intervals paired with a gap (silences) are kept, transcript words paired with a gap are dropped.
    (The reference delegates to Bio.pairwise2 with one_alignment_only; among equally scored alignments this takes the one
    that pairs items as late as possible — tie order of pairwise2 itself is unpinned here.)"""
    sil_word, oov_word = lexicon.silence_word, lexicon.oov_word

    def score(r: str, t: WordCtmInterval) -> float:
        if r == t.label:
            return 0.0
        if r == sil_word or t.label == sil_word:
            return -10.0
        if lexicon.to_int(r) == lexicon.to_int(t.label):
            return 0.0
        return -2.0

    # Every aligned interval comes back, in order, whatever the alignment pairs it with (both traceback branches that
    # consume an interval keep it); only intervals labelled with the OOV word can change.  Without one there is nothing to do.
    if not any(t.label == oov_word for t in test):
        return list(test)
    n, m = len(ref), len(test)
    gap = -2.0
    sc = [[score(r, t) for t in test] for r in ref]
    S = [[gap * j for j in range(m + 1)]] + [[gap * i] + [0.0] * m for i in range(1, n + 1)]     # (lists: 3x numpy scalars)
    for i in range(1, n + 1):
        up, cur, row = S[i - 1], S[i], sc[i - 1]
        for j in range(1, m + 1):
            cur[j] = max(up[j - 1] + row[j - 1], up[j] + gap, cur[j - 1] + gap)
    out: List[WordCtmInterval] = []
    i, j = n, m
    while i > 0 or j > 0:
        if i > 0 and j > 0 and S[i][j] == S[i - 1][j - 1] + sc[i - 1][j - 1]:
            t = test[j - 1]
            if ref[i - 1] != t.label and t.label == oov_word:
                t.label = ref[i - 1]
            out.append(t)
            i, j = i - 1, j - 1
        elif j > 0 and S[i][j] == S[i][j - 1] + gap:
            out.append(test[j - 1])      # aligned interval without a transcript word (silence): kept
            j -= 1
        else:
            i -= 1                        # transcript word without an interval: dropped
    out.reverse()
    return out


# --------------------------------------------------------------------------------------------------- TextGrid
def _escape(label) -> str:
    return str(label).replace('"', '""')


def _fill_blanks(entries: List[tuple], max_t: float) -> List[tuple]:
    """MFA/textgrid.py:115-131: blank intervals for gaps larger than 1 ms (start, between, end)."""
    if not entries:
        return entries
    e = list(entries)
    if e[0][0] > 0.001:
        e.insert(0, (0.0, e[0][0], ""))
    i = 1
    while i < len(e):
        start = e[i][0]
        prev_end = e[i - 1][1]
        if start - prev_end > 0.001:
            e.insert(i, (prev_end, start, ""))
            i += 1
        i += 1
    if max_t - e[-1][1] > 0.001:
        e.append((e[-1][1], max_t, ""))
    return e


def export_textgrid(speaker_data: Dict[str, Dict[str, List[CtmInterval]]], output_path: Path, duration: float,
                    frame_shift: float, output_format: str = "long_textgrid") -> None:
    """Same rules, same text as MFA/textgrid.py:463-572 (+ Textgrid.save :50-161)."""
    has_data = False
    duration = round(duration, 6)
    output_path = Path(output_path)
    if output_format == "csv":
        rows = []
        for speaker, data in speaker_data.items():
            for annotation_type, intervals in data.items():
                has_data = has_data or bool(intervals)
                for a in intervals:
                    if duration - a.end < frame_shift * 2:
                        a.end = duration
                    rows.append({"Begin": a.begin, "End": a.end, "Label": a.label, "Type": annotation_type, "Speaker": speaker})
        if has_data:
            with open(output_path, "w", encoding="utf8", newline="") as f:
                w = csv.DictWriter(f, fieldnames=["Begin", "End", "Label", "Type", "Speaker"])
                w.writeheader()
                w.writerows(rows)
        return
    if output_format == "json":
        js = {"start": 0, "end": duration, "tiers": {}}
        for speaker, data in speaker_data.items():
            for annotation_type, intervals in data.items():
                tier_name = f"{speaker} - {annotation_type}" if len(speaker_data) > 1 else annotation_type
                tier = js["tiers"].setdefault(tier_name, {"type": "interval", "entries": []})
                has_data = has_data or bool(intervals)
                for a in intervals:
                    if duration - a.end < frame_shift * 2:
                        a.end = duration
                    tier["entries"].append([a.begin, a.end, a.label])
        if has_data:
            with open(output_path, "w", encoding="utf8") as f:
                json.dump(js, f, indent=4, ensure_ascii=False)
        return
    tiers: Dict[str, List[tuple]] = {}
    for speaker, data in speaker_data.items():
        for annotation_type, intervals in data.items():
            has_data = has_data or bool(intervals)
            tier_name = f"{speaker} - {annotation_type}" if len(speaker_data) > 1 else annotation_type
            entries = tiers.setdefault(tier_name, [])
            for i, a in enumerate(sorted(intervals, key=lambda x: x.begin)):
                if i == len(intervals) - 1 and duration - a.end < frame_shift * 2:
                    a.end = duration
                iv = a.to_tg_interval()
                if i > 0 and entries[-1][1] > iv[0]:
                    a.begin = entries[-1][1]
                    iv = a.to_tg_interval()
                entries.append(iv)
    if not has_data:
        return
    for entries in tiers.values():
        if entries and entries[-1][1] > duration:
            entries[-1] = (entries[-1][0], duration, entries[-1][2])
    tab = " " * 4
    with open(output_path, "w", encoding="utf8") as fd:
        long_fmt = output_format != "short_textgrid"
        fd.write('File type = "ooTextFile"\nObject class = "TextGrid"\n\n')
        if long_fmt:
            fd.write(f"xmin = 0 \nxmax = {duration} \ntiers? <exists> \nsize = {len(tiers)} \nitem []: \n")
        else:
            fd.write(f"0\n{duration}\n<exists>\n{len(tiers)}\n")
        for num, (name, entries) in enumerate(tiers.items()):
            entries = _fill_blanks(entries, duration)
            tname = _escape(name)
            if long_fmt:
                fd.write(tab + f"item [{num + 1}]:\n")
                fd.write(tab * 2 + 'class = "IntervalTier" \n')
                fd.write(tab * 2 + f'name = "{tname}" \n')
                fd.write(tab * 2 + "xmin = 0 \n")
                fd.write(tab * 2 + f"xmax = {duration} \n")
                fd.write(tab * 2 + f"intervals: size = {len(entries)} \n")
            else:
                fd.write(f'"IntervalTier"\n"{tname}"\n0\n{duration}\n{len(entries)}\n')
            for i, (start, end, label) in enumerate(entries):
                label = _escape(label)
                if long_fmt:
                    fd.write(f"{tab * 2}intervals [{i + 1}]:\n{tab * 3}xmin = {start} \n{tab * 3}xmax = {end} \n"
                             f'{tab * 3}text = "{label}" \n')
                else:
                    fd.write(f'{start}\n{end}\n"{label}"\n')


def read_short_textgrid(path) -> Dict[str, List[tuple]]:
    """Minimal reader of Praat short-format TextGrids (the form of the reference's fixtures), for round-trip tests."""
    lines = Path(path).read_text(encoding="utf8").split("\n")
    i = 3
    _xmin, _xmax = float(lines[i]), float(lines[i + 1])
    n_tiers = int(lines[i + 3])
    i += 4
    tiers: Dict[str, List[tuple]] = {}
    for _ in range(n_tiers):
        name = lines[i + 1].strip('"')
        n = int(lines[i + 4])
        i += 5
        ent = []
        for _ in range(n):
            ent.append((float(lines[i]), float(lines[i + 1]), lines[i + 2][1:-1].replace('""', '"')))
            i += 3
        tiers[name] = ent
    return tiers
