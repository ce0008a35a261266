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


@dataclass
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
    of complete phones."""
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


def generate_ctm(alignment: Sequence[int], tm, phone_table, frame_shift: float = 0.01) -> List[CtmInterval]:
    out = []
    for first, n, phone in split_to_phones(alignment, tm):
        label = phone_table.find(phone) if phone_table is not None else phone
        out.append(CtmInterval(round(first * frame_shift, 6), round((first + n) * frame_shift, 6), label, phone))
    return out


def phones_to_pronunciations(lexicon, word_ids: Sequence[int], intervals: Sequence[CtmInterval], transcription: bool = False,
                             text: Optional[str] = None) -> HierarchicalCtm:
    """Group phone intervals into the aligned word sequence.

    The reference composes the phone string with the align lexicon FST; here the same constraint is solved directly:
    walk the word-id sequence the decoder emitted, and for each word take the pronunciation variant whose phones match the
    next non-silence intervals (position-independent comparison); optional-silence intervals between words become
    ``silence_word`` entries, as MFA stores them."""
    sil = lexicon.silence_phone
    strip = (lambda s: s.rsplit("_", 1)[0] if lexicon.position_dependent_phones and s[-2:] in ("_B", "_E", "_I", "_S") else s)
    labels = [strip(str(iv.label)) for iv in intervals]
    out: List[WordCtmInterval] = []
    k = 0
    n = len(intervals)
    sil_id = lexicon.word_table.find(lexicon.silence_word)

    def take_silence():
        nonlocal k
        while k < n and labels[k] == sil:
            out.append(WordCtmInterval(lexicon.silence_word, sil_id, sil, [intervals[k]]))
            k += 1

    for wid in word_ids:
        take_silence()
        word = lexicon.word_table.find(int(wid))
        prons = lexicon.word_pronunciations(word) if word != lexicon.oov_word else lexicon.word_pronunciations("\0oov\0")
        chosen = None
        for p in sorted(prons, key=lambda p: -len(p.pronunciation.split())):
            ph = p.pronunciation.split()
            if labels[k: k + len(ph)] == ph:
                chosen = ph
                break
        if chosen is None:
            raise CtmError(f"no pronunciation of {word!r} matches the aligned phones at interval {k}")
        out.append(WordCtmInterval(word, int(wid), " ".join(chosen), list(intervals[k: k + len(chosen)])))
        k += len(chosen)
    take_silence()
    if k != n:
        raise CtmError(f"{n - k} aligned phones are not covered by the word sequence")
    return HierarchicalCtm(out, text=text)


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
