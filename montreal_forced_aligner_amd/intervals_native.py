"""Batched native interval extraction and TextGrid text (libmfa_intervals.so, include/mfa_intervals.h) behind ``ctm.py``.

``ctm.py`` is the specification — one ``CtmInterval`` object per phone, in Python: 0.2–0.6 ms per utterance and core, i.e. two
orders of magnitude slower than the device aligns.  The reference does this step per utterance too
(AlignmentExtractionFunction, MFA/alignment/multiprocessing.py:1733-1751; export_textgrid, MFA/textgrid.py:463-572); here a
whole batch of alignments — the arrays the device hands back — goes through ``csrc/intervals.cpp`` in one threaded call:
SplitToPhones, the word-grouping search, ``update_utterance_boundaries``, the transcript spelling of ``<unk>`` words, and on
request the bytes of the TextGrid / json / csv files.  ``CtmInterval`` objects are built only for callers that ask for them
(``IntervalBatch.ctm``), and they are the very objects ``ctm.py`` would have built (tests/test_intervals_native_cpu.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import ctm as _ctm

_PKG = Path(__file__).resolve().parent
_SO = _PKG / "libmfa_intervals.so"
_SRC = _PKG / "csrc" / "intervals.cpp"
_HDR = _PKG.parent / "include" / "mfa_intervals.h"

FORMATS = {"long_textgrid": 0, "short_textgrid": 1, "json": 2, "csv": 3}
OK, SKIPPED, IRREGULAR, UNSPELLABLE = 0, 1, 2, 3


def build_native(force: bool = False, verbose: bool = False) -> Path:
    """g++ -O2 -shared of csrc/intervals.cpp next to this file (host code only: no hipcc, no GPU)."""
    if not force and _SO.exists() and all(_SO.stat().st_mtime >= d.stat().st_mtime for d in (_SRC, _HDR) if d.exists()):
        return _SO
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
           "-fvisibility=hidden", "-o", str(_SO), str(_SRC)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _SO


class _Config(C.Structure):
    _fields_ = [("n_tids", C.c_int32), ("id2state", C.c_void_p), ("id2phone", C.c_void_p), ("is_self_loop", C.c_void_p),
                ("is_final", C.c_void_p), ("n_words", C.c_int32), ("word_var_off", C.c_void_p), ("var_off", C.c_void_p),
                ("var_phones", C.c_void_p), ("sil_phone", C.c_int32), ("sil_word", C.c_int32), ("oov_word", C.c_int32),
                ("frame_shift", C.c_double), ("n_phone_names", C.c_int32), ("phone_name_off", C.c_void_p),
                ("phone_names", C.c_void_p), ("word_name_off", C.c_void_p), ("word_names", C.c_void_p)]


_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
SIGNATURES = {
    "mfa_iv_create": (_vp, [C.POINTER(_Config)]),
    "mfa_iv_destroy": (None, [_vp]),
    "mfa_iv_last_error": (C.c_char_p, [_vp]),
    "mfa_iv_version": (C.c_int, []),
    "mfa_iv_extract_batch": (C.c_int, [_vp, _i32] + [_vp] * 5 + [_i32] + [_vp] * 3),
    "mfa_iv_fetch": (C.c_int, [_vp] * 9),
    "mfa_iv_save_files": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "mfa_iv_write_files": (C.c_int, [_vp, _i32, _i32, _i32] + [_vp] * 18 + [_i32] + [_vp] * 4 + [_i32, _vp, _i64, _vp, _vp, _vp]),
}
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _SO.exists():
            build_native()
        lib = C.CDLL(str(_SO))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class IntervalError(RuntimeError):
    pass


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def _strings(names: Sequence[str]):
    enc = [s.encode("utf8") for s in names]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in enc], out=off[1:])
    blob = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy()
    return off, blob


def _threads() -> int:
    from . import hostcpu
    return hostcpu.threads()


class IntervalBatch:
    """Phone intervals and word items of a batch of alignments, as arrays (see include/mfa_intervals.h), plus what is needed
    to turn one utterance into the ``HierarchicalCtm`` ``ctm.py`` would have built."""

    def __init__(self, ex: "IntervalExtractor", frame_off, ali, words, n_words, arrays: Dict[str, np.ndarray]):
        self.ex = ex
        self.frame_off, self.ali, self.words, self.n_words = frame_off, ali, words, n_words
        # ph_off, it_off (prefix sums per utterance), ph_first, ph_len, ph_id, it_word, it_var, it_first, it_count, it_ref, err
        self.__dict__.update(arrays)
        self.n_phones = np.diff(self.ph_off).astype(np.int32)
        self.n_items = np.diff(self.it_off).astype(np.int32)

    @property
    def n_utt(self) -> int:
        return len(self.ph_off) - 1

    @classmethod
    def concat(cls, batches: Sequence["IntervalBatch"]) -> "IntervalBatch":
        """The interval arrays of several batches laid end to end (utterance k of batch j becomes
        ``sum(n_utt of batches before j) + k``) — what ``write_files`` needs when a sound file's utterances were aligned in
        different batches.  The result carries no alignments (``ctm`` is asked of the original batches)."""
        if len(batches) == 1:
            return batches[0]
        cat = lambda key: np.concatenate([getattr(b, key) for b in batches])     # noqa: E731
        arrays = {k: cat(k) for k in ("ph_first", "ph_len", "ph_id", "it_word", "it_var", "it_first", "it_count", "it_ref")}
        arrays["err"] = np.concatenate([b.err[: b.n_utt] for b in batches])
        for key in ("ph_off", "it_off"):
            parts, at = [np.zeros(1, dtype=np.int64)], 0
            for b in batches:
                off = getattr(b, key)
                parts.append(off[1:] + at)
                at += int(off[-1])
            arrays[key] = np.concatenate(parts)
        return cls(batches[0].ex, None, None, None, None, arrays)

    def oov_items(self) -> np.ndarray:
        """Utterances that hold an out-of-vocabulary item (candidates for the transcript's spelling)."""
        hit = np.flatnonzero(self.it_word == self.ex.oov_word)
        if hit.shape[0] == 0:
            return hit
        return np.unique(np.searchsorted(self.it_off, hit, side="right") - 1)

    def relabels(self, texts: Sequence[Optional[str]]) -> List[tuple]:
        """(utterance, transcript position, spelling) for every out-of-vocabulary item: what ``fix_unk_words`` does when the
        aligned words are the transcript's (they are: the graph was compiled from it) — the k-th non-silence item is the k-th
        transcript word."""
        out = []
        for u in self.oov_items().tolist():
            if texts[u] is None:
                continue
            toks = texts[u].split()
            q0, q1 = int(self.it_off[u]), int(self.it_off[u + 1])
            if int((self.it_ref[q0:q1] >= 0).sum()) != len(toks):
                continue          # word sequence differs from the transcript: left to the Python specification
            for q in np.flatnonzero(self.it_word[q0:q1] == self.ex.oov_word).tolist():
                r = int(self.it_ref[q0 + q])
                if r >= 0:
                    out.append((u, r, toks[r]))
        return out

    def ctm(self, u: int, text: Optional[str] = None, begin: float = 0.0, end: Optional[float] = None,
            likelihood: Optional[float] = None) -> _ctm.HierarchicalCtm:
        """The objects of ``generate_ctm`` → ``phones_to_pronunciations`` → ``update_utterance_boundaries`` →
        ``fix_unk_words`` for utterance ``u``; raises what ``ctm.py`` raises when the native side reported a code."""
        ex = self.ex
        a, b = int(self.frame_off[u]), int(self.frame_off[u + 1])
        if self.err[u] != OK:
            if self.err[u] == SKIPPED:
                raise _ctm.CtmError("the utterance has no alignment")
            ivs = _ctm.generate_ctm(self.ali[a:b], ex.tm, ex.lexicon.phone_table, ex.frame_shift)          # raises if irregular
            return _ctm.phones_to_pronunciations(ex.lexicon, self.words[a: a + int(self.n_words[u])], ivs, text=text)  # raises
        p0, p1, q0, q1 = int(self.ph_off[u]), int(self.ph_off[u + 1]), int(self.it_off[u]), int(self.it_off[u + 1])
        first, length, pid = self.ph_first[p0:p1], self.ph_len[p0:p1], self.ph_id[p0:p1]
        begins = _ctm._frame_times(first, ex.frame_shift)
        ends = _ctm._frame_times(first.astype(np.int64) + length, ex.frame_shift)
        names = ex.phone_names
        ivs = [_ctm.CtmInterval(bb, ee, names[p] if 0 <= p < len(names) else "", p) for bb, ee, p in zip(begins, ends, pid.tolist())]
        out: List[_ctm.WordCtmInterval] = []
        lex = ex.lexicon
        for w, v, f, c in zip(self.it_word[q0:q1].tolist(), self.it_var[q0:q1].tolist(), self.it_first[q0:q1].tolist(),
                              self.it_count[q0:q1].tolist()):
            if v < 0:
                out.append(_ctm.WordCtmInterval(lex.silence_word, ex.sil_word, lex.silence_phone, [ivs[f]]))
            else:
                out.append(_ctm.WordCtmInterval(ex.word_names[w], w, ex.variant_text(w, v), ivs[f: f + c]))
        h = _ctm.HierarchicalCtm(out, text=text)
        h.likelihood = likelihood
        h.update_utterance_boundaries(begin, end)
        if text is not None:
            h.word_intervals = _ctm.fix_unk_words(text.split(), h.word_intervals, lex)
        return h


class IntervalExtractor:
    """``IntervalExtractor(tm, lexicon, frame_shift)``: tables of the transition model and the lexicon handed to the native
    library once; ``extract`` per batch of alignments, ``write_files`` for the text of the output files."""

    def __init__(self, tm, lexicon, frame_shift: float = 0.01):
        self.lib = load()
        self.tm, self.lexicon, self.frame_shift = tm, lexicon, float(frame_shift)
        if lexicon.phone_table is None:
            lexicon.build_phone_table()
        pt, wt = lexicon.phone_table, lexicon.word_table
        n_ph = max(k for k, _ in pt) + 1
        self.phone_names = [pt.find(i) for i in range(n_ph)]
        n_w = max(k for k, _ in wt) + 1
        self.word_names = [wt.find(i) for i in range(n_w)]
        self.sil_word = int(wt.find(lexicon.silence_word))
        self.oov_word = int(wt.find(lexicon.oov_word))
        sil_phone = int(pt.find(lexicon.silence_phone))
        pos_dep = bool(lexicon.position_dependent_phones)
        # variants per word id, in the order ctm.phones_to_pronunciations tries them: longest first, duplicates removed
        word_var_off, var_off, var_phones = [0], [0], []
        self._var_text: List[List[str]] = []
        for w in range(n_w):
            name = self.word_names[w]
            texts: List[str] = []
            if name and w != self.sil_word:
                prons = lexicon.word_pronunciations(name) if name != lexicon.oov_word else lexicon.word_pronunciations("\0oov\0")
                seen = set()
                for p in sorted(prons, key=lambda p: -len(p.pronunciation.split())):
                    ph = tuple(p.pronunciation.split())
                    if ph in seen:
                        continue
                    seen.add(ph)
                    labels = _ctm._position_labels(list(ph)) if pos_dep else list(ph)
                    var_phones.extend(int(pt.find(lab)) for lab in labels)
                    var_off.append(len(var_phones))
                    texts.append(" ".join(ph))
            self._var_text.append(texts)
            word_var_off.append(len(var_off) - 1)
        self._keep = dict(
            id2state=np.ascontiguousarray(tm.id2state, dtype=np.int32), id2phone=np.ascontiguousarray(tm.id2phone, dtype=np.int32),
            isl=np.ascontiguousarray(tm.is_self_loop, dtype=np.int32), isf=np.ascontiguousarray(tm.is_final, dtype=np.int32),
            wvo=np.asarray(word_var_off, dtype=np.int32), vo=np.asarray(var_off, dtype=np.int32),
            vp=np.asarray(var_phones + [0], dtype=np.int32))
        k = self._keep
        k["pno"], k["pn"] = _strings(self.phone_names)
        k["wno"], k["wn"] = _strings(self.word_names)
        cfg = _Config(int(tm.num_transition_ids), _ptr(k["id2state"]), _ptr(k["id2phone"]), _ptr(k["isl"]), _ptr(k["isf"]), n_w,
                      _ptr(k["wvo"]), _ptr(k["vo"]), _ptr(k["vp"]), sil_phone, self.sil_word, self.oov_word, self.frame_shift, n_ph,
                      _ptr(k["pno"]), _ptr(k["pn"]), _ptr(k["wno"]), _ptr(k["wn"]))
        self.h = self.lib.mfa_iv_create(C.byref(cfg))
        if not self.h:
            raise IntervalError("mfa_iv_create failed")
        self.h = C.c_void_p(self.h)

    def variant_text(self, word_id: int, variant: int) -> str:
        return self._var_text[word_id][variant]

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.mfa_iv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract(self, frame_off: np.ndarray, ali: np.ndarray, words: np.ndarray, n_words: np.ndarray,
                status: Optional[np.ndarray] = None, n_threads: Optional[int] = None) -> IntervalBatch:
        """``ali`` / ``words`` [total frames] int32 in the layout the device writes (utterance u at frame_off[u]; its word
        ids packed at the same offset, ``n_words[u]`` of them)."""
        frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
        ali = np.ascontiguousarray(ali, dtype=np.int32)
        words = np.ascontiguousarray(words, dtype=np.int32)
        n_words = np.ascontiguousarray(n_words, dtype=np.int32)
        st = None if status is None else np.ascontiguousarray(status, dtype=np.int32)
        n = frame_off.shape[0] - 1
        total = int(frame_off[-1]) if n >= 0 else 0
        if ali.shape[0] < total or words.shape[0] < total:
            raise IntervalError("alignment arrays are shorter than frame_off says")
        ph_off, it_off = np.zeros(n + 1, dtype=np.int64), np.zeros(n + 1, dtype=np.int64)
        err = np.zeros(max(n, 1), dtype=np.int32)
        rc = self.lib.mfa_iv_extract_batch(self.h, n, _ptr(frame_off), _ptr(ali), _ptr(words), _ptr(n_words), _ptr(st),
                                           _threads() if n_threads is None else int(n_threads), _ptr(ph_off), _ptr(it_off), _ptr(err))
        if rc != 0:
            raise IntervalError((self.lib.mfa_iv_last_error(self.h) or b"error").decode())
        arr = {k: np.empty(max(int(ph_off[-1]), 1), dtype=np.int32) for k in ("ph_first", "ph_len", "ph_id")}
        arr.update({k: np.empty(max(int(it_off[-1]), 1), dtype=np.int32) for k in ("it_word", "it_var", "it_first", "it_count", "it_ref")})
        self.lib.mfa_iv_fetch(self.h, _ptr(arr["ph_first"]), _ptr(arr["ph_len"]), _ptr(arr["ph_id"]), _ptr(arr["it_word"]),
                              _ptr(arr["it_var"]), _ptr(arr["it_first"]), _ptr(arr["it_count"]), _ptr(arr["it_ref"]))
        for k in ("ph_first", "ph_len", "ph_id"):
            arr[k] = arr[k][: int(ph_off[-1])]
        for k in ("it_word", "it_var", "it_first", "it_count", "it_ref"):
            arr[k] = arr[k][: int(it_off[-1])]
        arr.update(ph_off=ph_off, it_off=it_off, err=err)
        return IntervalBatch(self, frame_off, ali, words, n_words, arr)

    def write_files(self, batch: IntervalBatch, files: Sequence[dict], utt_begin: np.ndarray, utt_end: np.ndarray,
                    relabel: Sequence[tuple] = (), output_format: str = "long_textgrid", cleanup_silence: bool = True,
                    n_threads: Optional[int] = None, paths: Optional[Sequence[str]] = None):
        """``files``: one dict per output file — ``duration`` and ``speakers`` = [(speaker name, [utterance indices of the
        batch, corpus order])] in first-appearance order.  ``relabel``: (utterance index, transcript position, spelling) for
        out-of-vocabulary items.  Returns (list of bytes or None per file, list of codes: 0 ok, 1 needs the Python writer, 2 no
        data)."""
        fmt = FORMATS[output_format]
        nf = len(files)
        dur = np.array([f["duration"] for f in files], dtype=np.float64).reshape(nf)
        fso, names, suo, su = [0], [], [0], []
        for f in files:
            for spk, utts in f["speakers"]:
                names.append(spk)
                su.extend(int(x) for x in utts)
                suo.append(len(su))
            fso.append(len(names))
        fso = np.asarray(fso, dtype=np.int32)
        sno, sn = _strings(names)
        suo = np.asarray(suo, dtype=np.int32)
        su = np.asarray(su + [0], dtype=np.int32)
        rl = sorted(relabel, key=lambda r: (r[0], r[1]))
        r_u = np.asarray([r[0] for r in rl] + [0], dtype=np.int32)
        r_r = np.asarray([r[1] for r in rl] + [0], dtype=np.int32)
        r_off, r_txt = _strings([r[2] for r in rl])
        ub = np.ascontiguousarray(utt_begin, dtype=np.float64)
        ue = np.ascontiguousarray(utt_end, dtype=np.float64)
        b = batch
        cap = int(64 + 96 * int(b.ph_off[-1]) + 128 * int(b.it_off[-1]) + 1024 * nf + 64 * len(names))
        out_off = np.zeros(nf + 1, dtype=np.int64)
        ferr = np.zeros(max(nf, 1), dtype=np.int32)
        needed = C.c_int64(0)
        for _attempt in range(2):
            out = np.empty(max(cap, 1), dtype=np.uint8)
            rc = self.lib.mfa_iv_write_files(
                self.h, fmt, int(bool(cleanup_silence)), nf, _ptr(dur), _ptr(fso), _ptr(sno), _ptr(sn), _ptr(suo), _ptr(su),
                _ptr(ub), _ptr(ue), _ptr(b.ph_off), _ptr(b.ph_first), _ptr(b.ph_len), _ptr(b.ph_id), _ptr(b.it_off),
                _ptr(b.it_word), _ptr(b.it_first), _ptr(b.it_count), _ptr(b.it_ref), _ptr(b.err), len(rl),
                _ptr(r_u), _ptr(r_r), _ptr(r_off), _ptr(r_txt), _threads() if n_threads is None else int(n_threads), _ptr(out),
                cap, _ptr(out_off), _ptr(ferr), C.addressof(needed))
            if rc == -2:
                cap = int(needed.value) + 64
                continue
            if rc != 0:
                raise IntervalError((self.lib.mfa_iv_last_error(self.h) or b"error").decode())
            break
        if paths is not None:
            p_off, p_blob = _strings([str(p) for p in paths])
            io_err = np.zeros(max(nf, 1), dtype=np.int32)
            bad = self.lib.mfa_iv_save_files(nf, _ptr(p_off), _ptr(p_blob), _ptr(out), _ptr(out_off), _ptr(ferr),
                                             _threads() if n_threads is None else int(n_threads), _ptr(io_err))
            if bad:
                k = int(np.flatnonzero(io_err[:nf])[0])
                raise OSError(int(io_err[k]), os.strerror(int(io_err[k])) if io_err[k] > 0 else "write failed", str(paths[k]))
            return None, ferr[:nf].tolist()
        mv = memoryview(out)
        texts = [bytes(mv[int(out_off[f]): int(out_off[f + 1])]) if ferr[f] == 0 else None for f in range(nf)]
        return texts, ferr[:nf].tolist()
