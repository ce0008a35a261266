"""The native interval / TextGrid layer (include/mfa_intervals.h, csrc/intervals.cpp) against its specification, ctm.py —
which restates AlignmentExtractionFunction (MFA/alignment/multiprocessing.py:1733-1751) and export_textgrid
(MFA/textgrid.py:463-572): same phone intervals, same word items, the same ``CtmInterval`` objects when asked for, and
byte-identical long / short TextGrid, json and csv files — multi-speaker files, utterance offsets, the end-of-file snap,
overlap clipping, quotes in labels, out-of-vocabulary words restored from the transcript."""
import json
import re
from pathlib import Path

import numpy as np
import pytest

from montreal_forced_aligner_amd import ctm as C
from montreal_forced_aligner_amd import intervals_native as N
from tests import helpers
from tests.test_ctm_cpu import _toy_lexicon

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    N.build_native()
    header = (ROOT / "include" / "mfa_intervals.h").read_text()
    declared = set(re.findall(r"MFA_IV_API\s+[\w\s\*]+?\b(mfa_iv_\w+)\s*\(", header))
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    lib = N.load()
    assert all(hasattr(lib, name) for name in declared) and lib.mfa_iv_version() >= 1


def _oracle_alignments(fx, texts, frames, seed):
    tm = fx.mono_tm
    rng = np.random.default_rng(seed)
    pdfs = np.arange(tm.num_pdfs, dtype=np.int32)
    out = []
    for text, T in zip(texts, frames):
        ll = rng.normal(-60.0, 10.0, size=(T, tm.num_pdfs)).astype(np.float32)
        r = helpers.oracle_align(tm, fx.mono_graph(text), ll, pdfs, beam=1e4, retry_beam=0.0)
        assert r["status"] == 0
        out.append(r)
    return out


def _pack(results):
    fo = np.concatenate([[0], np.cumsum([len(r["ali"]) for r in results])]).astype(np.int64)
    ali = np.concatenate([r["ali"] for r in results]).astype(np.int32)
    words = np.zeros_like(ali)
    for u, r in enumerate(results):
        words[fo[u]: fo[u] + len(r["words"])] = r["words"]
    return fo, ali, words, np.array([len(r["words"]) for r in results], dtype=np.int32)


TEXTS = ["this is the acoustic corpus", "i'm talking pretty fast here", "there's nothing going else going on zzz",
         "um and that should be all thanks", "qqq", "we're just saying some words"]
FRAMES = [200, 230, 420, 310, 60, 250]


def _same_ctm(a: C.HierarchicalCtm, b: C.HierarchicalCtm):
    assert len(a.word_intervals) == len(b.word_intervals)
    for x, y in zip(a.word_intervals, b.word_intervals):
        assert (x.label, x.word_id, x.pronunciation) == (y.label, y.word_id, y.pronunciation)
        assert [(p.begin, p.end, p.label, p.symbol) for p in x.phones] == [(p.begin, p.end, p.label, p.symbol) for p in y.phones]


def test_extraction_equals_the_python_specification(fx):
    res = _oracle_alignments(fx, TEXTS, FRAMES, seed=5)
    fo, ali, words, nw = _pack(res)
    ex = N.IntervalExtractor(fx.mono_tm, fx.mono_lex, 0.01)
    b = ex.extract(fo, ali, words, nw)
    assert np.all(b.err[: len(res)] == N.OK)
    for u, (r, text) in enumerate(zip(res, TEXTS)):
        segs = C.split_to_phones(r["ali"], fx.mono_tm)
        p0, p1 = int(b.ph_off[u]), int(b.ph_off[u + 1])
        got = list(zip(b.ph_first[p0:p1].tolist(), b.ph_len[p0:p1].tolist(), b.ph_id[p0:p1].tolist()))
        assert got == segs
        for begin, end in ((0.0, None), (3.25, 3.25 + len(r["ali"]) * 0.01 - 0.004), (0.000031, None)):
            ivs = C.generate_ctm(r["ali"], fx.mono_tm, fx.mono_lex.phone_table, 0.01)
            ref = C.phones_to_pronunciations(fx.mono_lex, r["words"], ivs, text=text)
            ref.update_utterance_boundaries(begin, end)
            ref.word_intervals = C.fix_unk_words(text.split(), ref.word_intervals, fx.mono_lex)
            _same_ctm(b.ctm(u, text=text, begin=begin, end=end), ref)
    assert set(b.oov_items().tolist()) == {2, 4}                      # "zzz" and "qqq" are not in the dictionary
    labels = [w.label for w in b.ctm(2, text=TEXTS[2]).word_intervals if w.label != fx.mono_lex.silence_word]
    assert labels == TEXTS[2].split()


def test_irregular_and_unspellable_alignments_are_reported_per_utterance(fx):
    res = _oracle_alignments(fx, TEXTS[:3], FRAMES[:3], seed=6)
    fo, ali, words, nw = _pack(res)
    ali = ali.copy(); words = words.copy()
    ali[fo[1] + 40] = ali[fo[1] + 120]                                # a jump inside a phone
    words[fo[2]: fo[2] + nw[2]] = words[fo[2]: fo[2] + nw[2]][::-1].copy()   # phones the words no longer spell
    status = np.array([0, 1, 0], dtype=np.int32)
    ex = N.IntervalExtractor(fx.mono_tm, fx.mono_lex, 0.01)
    b = ex.extract(fo, ali, words, nw, status)
    assert b.err[:3].tolist() == [N.OK, N.IRREGULAR, N.UNSPELLABLE]
    with pytest.raises(C.CtmError):
        b.ctm(1)
    with pytest.raises(C.CtmError):
        b.ctm(2)
    b2 = ex.extract(fo, ali, words, nw, np.array([0, 2, 5], dtype=np.int32))
    assert b2.err[:3].tolist() == [N.OK, N.SKIPPED, N.SKIPPED]


class _ToyTm:
    """One transition-state per phone, a forward-final and a self-loop transition-id each: enough for SplitToPhones."""

    def __init__(self, n_phones):
        self.num_transition_ids = 2 * n_phones
        self.id2state = np.zeros(2 * n_phones + 1, np.int32)
        self.id2phone = np.zeros(2 * n_phones + 1, np.int32)
        self.is_self_loop = np.zeros(2 * n_phones + 1, np.int32)
        self.is_final = np.zeros(2 * n_phones + 1, np.int32)
        for p in range(1, n_phones + 1):
            for k, tid in enumerate((2 * p - 1, 2 * p)):
                self.id2state[tid], self.id2phone[tid] = p, p
                self.is_final[tid], self.is_self_loop[tid] = int(k == 0), int(k == 1)


@pytest.mark.parametrize("position_dependent", [False, True])
def test_word_grouping_search_order_matches_python(position_dependent):
    lex = _toy_lexicon(position_dependent)
    pt = lex.phone_table
    tm = _ToyTm(max(k for k, _ in pt))
    ex = N.IntervalExtractor(tm, lex, 0.01)

    def ali_of(labels):
        out = []
        for lab in labels:
            p = pt.find(lab)
            out += [2 * p - 1, 2 * p]
        return np.asarray(out, dtype=np.int32)

    pos = (lambda ws: [lab for w in ws for lab in C._position_labels(w.split())]) if position_dependent else \
        (lambda ws: [p for w in ws for p in w.split()])
    cases = [(["ab", "c"], pos(["a b", "c"])), (["ab", "bc"], pos(["a", "b c"])), (["ab", "c"], pos(["a b"]) + ["sil"] + pos(["c"])),
             (["d", "d"], pos(["d d", "d"])), (["d", "d"], pos(["d", "d d"])), (["abc"], ["sil"] + pos(["a b c"]) + ["sil", "sil"]),
             (["ab", "c"], pos(["a b"])), (["c"], pos(["a"]))]
    for word_list, labels in cases:
        ids = np.asarray([lex.word_table.find(w) for w in word_list], dtype=np.int32)
        ali = ali_of(labels)
        fo = np.array([0, len(ali)], dtype=np.int64)
        words = np.zeros(len(ali), np.int32); words[: len(ids)] = ids
        b = ex.extract(fo, ali, words, np.array([len(ids)], np.int32))
        ivs = C.generate_ctm(ali, tm, pt, 0.01)
        try:
            ref = C.phones_to_pronunciations(lex, ids, ivs)
        except C.CtmError:
            assert b.err[0] == N.UNSPELLABLE, (word_list, labels)
            continue
        assert b.err[0] == N.OK, (word_list, labels)
        _same_ctm(b.ctm(0), ref)


def _python_files(fx, res, texts, utts, fmt, tmp, cleanup=True):
    """The specification: objects per utterance, CorpusAligner.export_textgrids' grouping, ctm.export_textgrid."""
    per_file = {}
    sil = fx.mono_lex.silence_word
    for k, (r, text) in enumerate(zip(res, texts)):
        name, spk, begin, end, fdur = utts[k]
        ivs = C.generate_ctm(r["ali"], fx.mono_tm, fx.mono_lex.phone_table, 0.01)
        h = C.phones_to_pronunciations(fx.mono_lex, r["words"], ivs, text=text)
        h.update_utterance_boundaries(begin, end)
        h.word_intervals = C.fix_unk_words(text.split(), h.word_intervals, fx.mono_lex)
        f = per_file.setdefault(name, dict(duration=0.0, speakers={}))
        f["duration"] = max(f["duration"], fdur or end)
        tiers = f["speakers"].setdefault(spk, {"words": [], "phones": []})
        for w in h.word_intervals:
            if cleanup and w.label == sil:
                continue
            tiers["words"].append(C.CtmInterval(w.begin, w.end, w.label))
            tiers["phones"].extend(w.phones)
    out = {}
    for name, f in per_file.items():
        for tiers in f["speakers"].values():
            tiers["words"].sort(); tiers["phones"].sort()
        path = tmp / f"{name}.{fmt}"
        C.export_textgrid(f["speakers"], path, f["duration"], 0.01, fmt)
        out[name] = path.read_bytes() if path.exists() else None
    return out


@pytest.mark.parametrize("fmt", ["long_textgrid", "short_textgrid", "json", "csv"])
@pytest.mark.parametrize("cleanup", [True, False])
def test_file_bytes_equal_the_python_writer(fx, tmp_path, fmt, cleanup):
    res = _oracle_alignments(fx, TEXTS, FRAMES, seed=7)
    fo, ali, words, nw = _pack(res)
    dur = [len(r["ali"]) * 0.01 for r in res]
    # (file, speaker, begin, end, file duration): a two-speaker file with offsets and a near-overlap, a file whose last
    # interval snaps to the end, a speaker name with quotes and a comma, a single-utterance file without a duration
    utts = [("f0", "anna", 0.0, dur[0], 9.137), ("f0", 'bob "b", jr', 2.004999, 2.004999 + dur[1] - 0.003, 9.137),
            ("f0", "anna", 1.99, 1.99 + dur[2], 9.137), ("f1", "anna", 0.25, 0.25 + dur[3], 4.0 + dur[5] + 0.015),
            ("f2", "carl", 0.0, dur[4], None), ("f1", "anna", 4.0, 4.0 + dur[5], 4.0 + dur[5] + 0.015)]
    ref = _python_files(fx, res, TEXTS, utts, fmt, tmp_path, cleanup)
    ex = N.IntervalExtractor(fx.mono_tm, fx.mono_lex, 0.01)
    b = ex.extract(fo, ali, words, nw)
    files, order = [], {}
    for k, (name, spk, begin, end, fdur) in enumerate(utts):
        if name not in order:
            order[name] = len(files)
            files.append(dict(name=name, duration=0.0, speakers=[]))
        f = files[order[name]]
        f["duration"] = max(f["duration"], fdur or end)
        for s in f["speakers"]:
            if s[0] == spk:
                s[1].append(k)
                break
        else:
            f["speakers"].append((spk, [k]))
    relabel = b.relabels(TEXTS)
    assert sorted(r[2] for r in relabel) == ["qqq", "zzz"]
    texts, codes = ex.write_files(b, files, np.array([u[2] for u in utts]), np.array([u[3] for u in utts]), relabel, fmt, cleanup)
    assert codes == [0, 0, 0]
    for f, t in zip(files, texts):
        assert t == ref[f["name"]], f["name"]
    if fmt == "json":
        js = json.loads(texts[0].decode("utf8"))
        assert list(js["tiers"])[:2] == ["anna - words", "anna - phones"] and js["end"] == 9.137
    if fmt == "long_textgrid":
        assert b'text = "zzz"' in texts[0] and b'name = "bob ""b"", jr - words"' in texts[0]


def test_collapsed_interval_and_empty_file_codes(fx):
    res = _oracle_alignments(fx, TEXTS[:2], FRAMES[:2], seed=8)
    fo, ali, words, nw = _pack(res)
    ex = N.IntervalExtractor(fx.mono_tm, fx.mono_lex, 0.01)
    b = ex.extract(fo, ali, words, nw, np.array([0, 2], dtype=np.int32))
    files = [dict(duration=1.0, speakers=[("a", [0])]), dict(duration=3.0, speakers=[("b", [1])])]
    # utterance 0 ends long after its file does: the clipped last interval is empty after rounding — the Python writer raises
    texts, codes = ex.write_files(b, files, np.zeros(2), np.array([0.5, 3.0]))
    assert codes == [1, 2] and texts == [None, None]


def test_python_float_repr_is_reproduced(fx, tmp_path):
    """Offsets that make boundaries awkward doubles: every number in the csv (unrounded begin/end) must be Python's repr."""
    res = _oracle_alignments(fx, TEXTS[:1], FRAMES[:1], seed=9)
    fo, ali, words, nw = _pack(res)
    ex = N.IntervalExtractor(fx.mono_tm, fx.mono_lex, 0.01)
    b = ex.extract(fo, ali, words, nw)
    for begin in (0.1 + 0.2, 1e-5, 12345.678901234, 1.0 / 3.0, 2.0 ** -20):
        end = begin + 2.0
        utts = [("f", "s", begin, end, begin + 2.5)]
        ref = _python_files(fx, res, TEXTS[:1], utts, "csv", tmp_path)
        texts, codes = ex.write_files(b, [dict(duration=begin + 2.5, speakers=[("s", [0])])], np.array([begin]), np.array([end]), (),
                                      "csv")
        assert codes == [0] and texts[0] == ref["f"]
        ref = _python_files(fx, res, TEXTS[:1], utts, "long_textgrid", tmp_path)
        texts, codes = ex.write_files(b, [dict(duration=begin + 2.5, speakers=[("s", [0])])], np.array([begin]), np.array([end]))
        assert codes == [0] and texts[0] == ref["f"]


def test_corpus_aligner_writes_the_same_files_from_arrays_and_from_objects(fx, tmp_path, monkeypatch):
    """CorpusAligner.export_textgrids: results that still carry their interval arrays go through the native writer, results
    whose objects a caller touched through ctm.export_textgrid — same bytes; utterances of one file aligned in different
    batches are merged."""
    from montreal_forced_aligner_amd.aligner import CorpusAligner, CorpusUtterance, _BatchOut

    class StubEngine:
        device = "cpu"

        def configure_mfcc(self, **kw):
            pass

        def load_gmm(self, am):
            pass

        def num_frames(self, n_samples):
            return n_samples // 160

    res = _oracle_alignments(fx, TEXTS, FRAMES, seed=11)
    outs = []
    for part in ([0, 2, 4], [1, 3, 5]):                       # two "batches"
        fo, ali, words, nw = _pack([res[k] for k in part])
        outs.append(_BatchOut(part, fo, ali, words, nw, np.array([res[k]["like"] for k in part], dtype=np.float32),
                              np.zeros(len(part), dtype=np.int32)))
    where = {k: (outs[j], i) for j, part in enumerate(([0, 2, 4], [1, 3, 5])) for i, k in enumerate(part)}
    sr = 16000
    meta = [("f0", "anna", 0.0), ("f0", "bob", 2.5), ("f0", "anna", 2.01), ("f1", "anna", 0.25), ("f2", "carl", 0.0), ("f1", "anna", 4.0)]
    utts = [CorpusUtterance(f"u{k}", spk, np.zeros(FRAMES[k] * 160, dtype=np.int16), TEXTS[k], begin=b, file_name=name)
            for k, (name, spk, b) in enumerate(meta)]
    al = CorpusAligner(fx.mono_tm, fx.mono_am, fx.mono_tree, fx.mono_lex, engine=StubEngine())
    monkeypatch.setattr(al, "speaker_cmvn", lambda u: ({"s": 0}, None))
    monkeypatch.setattr(al, "_submit_compile", lambda u, idx: None)
    monkeypatch.setattr(al, "_pass", lambda u, s, c, f, **kw: ([where[k] for k in range(6)], []))
    for fmt in ("long_textgrid", "json"):
        results = al.align(utts)
        assert all(r is not None and r._lazy is not None for r in results)
        native = al.export_textgrids(utts, results, tmp_path / f"native_{fmt}", fmt)
        assert all(r._lazy is not None for r in results)                  # no objects were built
        results2 = al.align(utts)
        for r in results2:
            r.ctm = r.ctm                                                 # objects built: the Python writer's turn
        python = al.export_textgrids(utts, results2, tmp_path / f"python_{fmt}", fmt)
        assert [p.name for p in native] == [p.name for p in python] and len(native) == 3
        for a, b in zip(native, python):
            assert a.read_bytes() == b.read_bytes(), a.name
    words = [w.label for w in results[2].ctm.word_intervals if w.label != fx.mono_lex.silence_word]
    assert words == TEXTS[2].split()
