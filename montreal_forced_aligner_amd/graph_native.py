"""Batched native training-graph compiler (libmfa_graph.so, include/mfa_graph.h) behind ``TrainingGraphCompiler``.

``graph.py`` is the specification — one utterance at a time, in Python, ≈11 ms per 10 s transcript and core.  The reference
compiles its graphs in kalpy's C++ (``TrainingGraphCompiler.compile_fst`` / ``export_graphs``,
MFA/alignment/multiprocessing.py:537-571), and a device that aligns 200 k utterances per second needs its graphs at that
rate: this module hands whole batches to ``csrc/graph_compile.cpp`` (one utterance per worker thread) and gets back the
very graphs ``graph.py`` builds — same state numbers, arc order and float32 weights (tests/test_graph_native_cpu.py).

The lexicon goes over as flat tables, once; the tree and the topology stay here and are consulted only for context windows
the native side has not seen yet."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

from . import graph as _graph
from .kaldi_io import ARC_DTYPE, Fst

_PKG = Path(__file__).resolve().parent
_SO = _PKG / "libmfa_graph.so"
_SRC = _PKG / "csrc" / "graph_compile.cpp"
_HDR = _PKG.parent / "include" / "mfa_graph.h"


def build_native(force: bool = False, verbose: bool = False) -> Path:
    """g++ -O2 -shared of csrc/graph_compile.cpp next to this file (host code only: no hipcc, no GPU)."""
    if not force and _SO.exists() and all(_SO.stat().st_mtime >= d.stat().st_mtime for d in (_SRC, _HDR) if d.exists()):
        return _SO
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
           "-fvisibility=hidden", "-o", str(_SO), str(_SRC)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _SO


class _Pron(C.Structure):
    _fields_ = [("phone_off", C.c_int32), ("n_phones", C.c_int32), ("c0_ns", C.c_double), ("c0_s", C.c_double),
                ("c0_start", C.c_double), ("w_ns", C.c_double), ("w_sil", C.c_double), ("has_sil", C.c_int32),
                ("pad", C.c_int32)]


class _Config(C.Structure):
    _fields_ = [("context_width", C.c_int32), ("share_suffixes", C.c_int32), ("sil_phone", C.c_int32),
                ("n_entries", C.c_int32), ("entry_word", C.c_void_p), ("entry_pron_off", C.c_void_p), ("prons", C.c_void_p),
                ("phones", C.c_void_p), ("cost_init_sil", C.c_double), ("cost_init_eps", C.c_double),
                ("final_ns", C.c_double), ("final_s", C.c_double), ("n_tids", C.c_int32), ("id2state", C.c_void_p),
                ("n_tstates", C.c_int32), ("self_loop_of", C.c_void_p), ("determinize", C.c_int32)]


class _Model(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("root", C.c_int32), ("kind", C.c_void_p), ("key", C.c_void_p), ("answer", C.c_void_p),
                ("a", C.c_void_p), ("b", C.c_void_p), ("yes_off", C.c_void_p), ("table", C.c_void_p), ("yes_vals", C.c_void_p),
                ("max_phone", C.c_int32), ("phone2entry", C.c_void_p), ("n_entries", C.c_int32), ("entry_state_off", C.c_void_p),
                ("fwd_class", C.c_void_p), ("slf_class", C.c_void_p), ("trans_off", C.c_void_p), ("trans_dst", C.c_void_p),
                ("n_tuples", C.c_int32), ("tuples", C.c_void_p), ("state2id", C.c_void_p)]


_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
SIGNATURES = {
    "mfa_gc_create": (_vp, [C.POINTER(_Config)]),
    "mfa_gc_destroy": (None, [_vp]),
    "mfa_gc_last_error": (C.c_char_p, [_vp]),
    "mfa_gc_add_windows": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "mfa_gc_prepare": (_i64, [_vp, _i32, _vp, _vp, _i32]),
    "mfa_gc_missing_windows": (C.c_int, [_vp, _vp]),
    "mfa_gc_set_model": (C.c_int, [_vp, C.POINTER(_Model)]),
    "mfa_gc_resolve_windows": (_i64, [_vp]),
    "mfa_gc_finish": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "mfa_gc_fetch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mfa_gc_fetch_columns": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
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


class GraphCompileError(RuntimeError):
    pass


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


class FstBatch(list):
    """The list of ``Fst`` a batch compile returns, with the concatenated arrays its elements are views of — what
    ``engine.pack_graphs`` needs, without gathering them again from the per-utterance objects."""

    state_off: np.ndarray     # int64 [n + 1]
    arc_base: np.ndarray      # int64 [n + 1]
    arc_off: np.ndarray       # int64 [total states + n]: utterance u's S_u + 1 offsets at state_off[u] + u
    arcs: np.ndarray          # ARC_DTYPE [total arcs]
    final: np.ndarray         # float32 [total states]
    # columns for the score-plan builder and the device layout (mfa_gc_fetch_columns); None when not asked for
    arc_off32: Optional[np.ndarray] = None    # int32 copy of arc_off
    arc_next: Optional[np.ndarray] = None     # int32 [total arcs]
    arc_pdf: Optional[np.ndarray] = None      # int32 [total arcs] pdf of the arc's transition-id
    staging = None                            # the object whose ``get`` allocated these arrays (engine.StagingPool), if any
    max_degree = min_ilabel = min_arcs = None  # (fetch_columns) largest out-degree, smallest input label, fewest arcs of an utterance


class NativeGraphCompiler:
    """Batch front end of a ``graph.TrainingGraphCompiler``: ``compile_batch(texts)`` → the list ``[compile_fst(t) for t in
    texts]`` (with ``scaled_log_probs``: ``add_transition_probs`` applied), built by the native library."""

    def __init__(self, compiler: _graph.TrainingGraphCompiler, n_threads: Optional[int] = None):
        self.compiler = compiler
        self.lib = load()
        from . import hostcpu
        self.n_threads = hostcpu.threads() if n_threads is None else max(1, int(n_threads))
        lex = compiler.lexicon
        if lex.phone_table is None:
            lex.build_phone_table()
        self._width = int(compiler.tree.context_width)
        # ---- lexicon tables: one entry per word form with pronunciations, plus the out-of-vocabulary entry (last)
        eps_cost = _graph._cost(1.0 - lex.initial_silence_probability)
        words = list(lex._by_word.keys())
        self._entry_of = {}
        entry_word, pron_off, prons, phones = [], [0], [], []
        self._bad_entries = set()          # a pronunciation with a phone outside the table: graph.py raises when it is used

        def add_entry(wid: int, plist) -> int:
            no_probs = all(p.probability is None for p in plist)
            mark = len(prons)
            try:
                for p in plist:
                    ids = lex.phone_ids(p)
                    pc = 0.0 if no_probs else _graph._cost(p.probability if p.probability is not None else 1.0)
                    p_after = p.silence_after_probability if p.silence_after_probability is not None else lex.silence_probability
                    nsb = _graph._cost(p.non_silence_before_correction)
                    prons.append((len(phones), len(ids), pc + nsb, pc + _graph._cost(p.silence_before_correction),
                                  eps_cost + pc + nsb, _graph._cost(1.0 - p_after), _graph._cost(p_after), 1 if p_after > 0 else 0))
                    phones.extend(ids)
            except KeyError:
                del prons[mark:]
                self._bad_entries.add(len(entry_word))
            entry_word.append(wid)
            pron_off.append(len(prons))
            return len(entry_word) - 1

        for w in words:
            self._entry_of[w] = add_entry(lex.word_table.find(w), lex._by_word[w])
        self._oov_wid = lex.word_table.find(lex.oov_word)
        self._oov_entry = add_entry(self._oov_wid, [_graph.Pronunciation(lex.oov_word, lex.oov_phone)])
        self._entry_word = np.asarray(entry_word, dtype=np.int32)
        self._pron_off = np.asarray(pron_off, dtype=np.int32)
        self._prons = (_Pron * max(1, len(prons)))()
        for k, t in enumerate(prons):
            self._prons[k] = _Pron(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], 0)
        self._phones = np.asarray(phones if phones else [0], dtype=np.int32)
        tm = compiler.tm
        self._id2state = np.ascontiguousarray(tm.id2state, dtype=np.int32)
        self._self_loop_of = np.ascontiguousarray(tm.self_loop_of, dtype=np.int32)
        cfg = _Config()
        cfg.context_width = self._width
        cfg.share_suffixes = 1 if lex.share_suffixes else 0
        cfg.sil_phone = int(lex.phone_table.find(lex.silence_phone))
        cfg.n_entries = len(entry_word)
        cfg.entry_word = _ptr(self._entry_word); cfg.entry_pron_off = _ptr(self._pron_off)
        cfg.prons = C.addressof(self._prons); cfg.phones = _ptr(self._phones)
        cfg.cost_init_sil = _graph._cost(lex.initial_silence_probability)
        cfg.cost_init_eps = eps_cost
        cfg.final_ns = _graph._cost(lex.final_non_silence_correction)
        cfg.final_s = _graph._cost(lex.final_silence_correction)
        cfg.n_tids = int(self._id2state.shape[0] - 1)
        cfg.id2state = _ptr(self._id2state)
        cfg.n_tstates = int(self._self_loop_of.shape[0] - 1)
        cfg.self_loop_of = _ptr(self._self_loop_of)
        cfg.determinize = 1 if getattr(compiler, "determinize", False) else 0
        self._h = self.lib.mfa_gc_create(C.byref(cfg))
        if not self._h:
            raise GraphCompileError("mfa_gc_create refused the configuration")
        self._set_model()

    def _set_model(self) -> None:
        """Tree, topology and transition-state table, flattened once (include/mfa_graph.h mfa_gc_model): context windows are
        then resolved inside the library; whatever it cannot answer still comes back through ``_hmm`` (which raises)."""
        tree, tm = self.compiler.tree, self.compiler.tm
        kind, key, answer, a, b, table, nodes = [], [], [], [], [], [], []
        index = {}

        def flat(node) -> int:          # EventMap objects may be shared between parents: one flat node per object
            if node is None:
                return -1
            got = index.get(id(node))
            if got is not None:
                return got
            i = len(kind)
            index[id(node)] = i
            nodes.append(node)
            kind.append({"CE": 0, "TE": 1, "SE": 2}[node.kind]); key.append(int(node.key)); answer.append(int(node.answer))
            a.append(0); b.append(0)
            if node.kind == "TE":
                kids = [flat(c) for c in node.table]
                a[i], b[i] = len(table), len(kids)
                table.extend(kids)
            elif node.kind == "SE":
                a[i], b[i] = flat(node.yes), flat(node.no)
                if a[i] < 0 or b[i] < 0:
                    raise GraphCompileError("SE node with a NULL child")
            return i

        import sys
        limit = sys.getrecursionlimit()
        sys.setrecursionlimit(max(limit, 10000))
        try:
            root = flat(tree.to_pdf)
        except (GraphCompileError, RecursionError):
            return                      # unusual tree: windows keep going through the Python callback
        finally:
            sys.setrecursionlimit(limit)
        yes_off, yes_vals = [0], []
        for node in nodes:
            if node.kind == "SE":
                yes_vals.extend(sorted(int(x) for x in node.yes_set))
            yes_off.append(len(yes_vals))
        topo = tm.topo
        p2e = np.asarray(topo.phone2idx, dtype=np.int32)
        eso, fwd, slf, toff, tdst = [0], [], [], [0], []
        for entry in topo.entries:
            for st in entry:
                fwd.append(int(st.forward_pdf_class)); slf.append(int(st.self_loop_pdf_class))
                tdst.extend(int(d) for d, _p in st.transitions)
                toff.append(len(tdst))
            eso.append(len(fwd))
        i32 = lambda x: np.ascontiguousarray(np.asarray(x if len(x) else [0], dtype=np.int32))   # noqa: E731
        self._model_keep = dict(kind=i32(kind), key=i32(key), answer=i32(answer), a=i32(a), b=i32(b), yes_off=i32(yes_off),
                                table=i32(table), yes_vals=i32(yes_vals), p2e=i32(p2e), eso=i32(eso), fwd=i32(fwd), slf=i32(slf),
                                toff=i32(toff), tdst=i32(tdst), tuples=np.ascontiguousarray(tm.tuples, dtype=np.int32),
                                s2i=np.ascontiguousarray(tm.state2id, dtype=np.int32))
        k = self._model_keep
        if k["s2i"].shape[0] < k["tuples"].shape[0] + 2:
            return
        m = _Model(len(kind), root, _ptr(k["kind"]), _ptr(k["key"]), _ptr(k["answer"]), _ptr(k["a"]), _ptr(k["b"]), _ptr(k["yes_off"]),
                   _ptr(k["table"]), _ptr(k["yes_vals"]), int(p2e.shape[0] - 1), _ptr(k["p2e"]), len(topo.entries), _ptr(k["eso"]),
                   _ptr(k["fwd"]), _ptr(k["slf"]), _ptr(k["toff"]), _ptr(k["tdst"]), int(k["tuples"].shape[0]), _ptr(k["tuples"]),
                   _ptr(k["s2i"]))
        self._check(self.lib.mfa_gc_set_model(self._h, C.byref(m)), "mfa_gc_set_model")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and self.lib is not None:
            self.lib.mfa_gc_destroy(h)

    def _check(self, rc, what):
        if rc < 0:
            msg = self.lib.mfa_gc_last_error(self._h)
            raise GraphCompileError(f"{what}: {msg.decode() if msg else 'error'}")
        return rc

    def _entries(self, text: str) -> Optional[List[int]]:
        """Lexicon entries of a transcript, or None when graph.py has to take it (a word id the tables do not hold)."""
        lex = self.compiler.lexicon
        get = self._entry_of.get
        words = (text.lower() if lex.ignore_case else text).split()
        out = [get(w) for w in words]
        if None in out:
            raw = text.split()
            if len(raw) != len(words):          # (a case mapping that changes the token count: leave it to graph.py)
                return None
            for k, e in enumerate(out):
                if e is None:
                    if lex.to_int(raw[k]) != self._oov_wid:
                        return None
                    out[k] = self._oov_entry
        if self._bad_entries and not self._bad_entries.isdisjoint(out):
            return None
        return out

    def compile_batch(self, texts: Sequence[str], scaled_log_probs: Optional[np.ndarray] = None, columns: bool = False,
                      alloc=None) -> List[Fst]:
        """``columns``: also fill ``FstBatch.arc_off32 / arc_next / arc_pdf`` (what ``engine.pack_graphs`` would otherwise
        derive with passes over the arcs).  ``alloc(name, n, dtype) -> np.ndarray``: where the batch's arrays live (e.g.
        reused pinned buffers); default fresh numpy arrays."""
        ent = [self._entries(t) for t in texts]
        native = [k for k, e in enumerate(ent) if e is not None]
        out: List[Optional[Fst]] = [None] * len(texts)
        for k, e in enumerate(ent):
            if e is None:       # (raises what graph.py raises for an unknown phone)
                f = self.compiler.compile_fst(texts[k])
                out[k] = _graph.add_transition_probs(f, scaled_log_probs) if scaled_log_probs is not None else f
        if not native:
            return out  # type: ignore[return-value]
        word_off = np.zeros(len(native) + 1, dtype=np.int64)
        for j, k in enumerate(native):
            word_off[j + 1] = word_off[j] + len(ent[k])
        entries = np.asarray([e for k in native for e in ent[k]] or [0], dtype=np.int32)
        missing = self._check(self.lib.mfa_gc_prepare(self._h, len(native), _ptr(word_off), _ptr(entries), self.n_threads),
                              "mfa_gc_prepare")
        if missing:
            missing = self._check(self.lib.mfa_gc_resolve_windows(self._h), "mfa_gc_resolve_windows")    # tree walked natively
        if missing:
            wins = np.zeros((missing, self._width), dtype=np.int32)
            self._check(self.lib.mfa_gc_missing_windows(self._h, _ptr(wins)), "mfa_gc_missing_windows")
            toff, trans, nfin = [0], [], []
            for row in wins:
                tr, n_final = self.compiler._hmm(tuple(int(x) for x in row))
                trans.extend(tr)
                toff.append(len(trans))
                nfin.append(n_final)
            t_off = np.asarray(toff, dtype=np.int32)
            t_arr = np.asarray(trans if trans else [(0, 0, 0)], dtype=np.int32).reshape(-1, 3)
            n_fin = np.asarray(nfin, dtype=np.int32)
            self._check(self.lib.mfa_gc_add_windows(self._h, int(missing), _ptr(wins), _ptr(t_off), _ptr(t_arr), _ptr(n_fin)),
                        "mfa_gc_add_windows")
        neg = None
        if scaled_log_probs is not None:
            neg = np.ascontiguousarray(-scaled_log_probs.astype(np.float32))
            if neg.shape[0] != self._id2state.shape[0]:
                raise GraphCompileError("scaled_log_probs does not match the transition model")
        n_states, n_arcs = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.mfa_gc_finish(self._h, _ptr(neg), self.n_threads, C.byref(n_states), C.byref(n_arcs)), "mfa_gc_finish")
        S, A, n = int(n_states.value), int(n_arcs.value), len(native)
        staging = getattr(alloc, "__self__", None)
        if alloc is None:
            alloc = lambda name, count, dtype: np.empty(count, dtype=dtype)   # noqa: E731
        state_off = np.zeros(n + 1, dtype=np.int64)
        arc_base = np.zeros(n + 1, dtype=np.int64)
        arc_off = alloc("arc_off", S + n, np.int64)
        arcs = alloc("arcs", A, ARC_DTYPE)
        final = alloc("final", S, np.float32)
        arc_off32 = arc_next = arc_pdf = None
        id2pdf = None
        if columns:
            arc_off32, arc_next, arc_pdf = alloc("arc_off32", S + n, np.int32), alloc("arc_next", A, np.int32), alloc("arc_pdf", A, np.int32)
            id2pdf = np.ascontiguousarray(self.compiler.tm.id2pdf, dtype=np.int32)
        stats = np.zeros(3, dtype=np.int32)
        self._check(self.lib.mfa_gc_fetch_columns(self._h, _ptr(id2pdf), self.n_threads, _ptr(state_off), _ptr(arc_base), _ptr(arc_off),
                                                  _ptr(arc_off32), _ptr(arcs), _ptr(final), _ptr(arc_next), _ptr(arc_pdf), _ptr(stats)),
                    "mfa_gc_fetch_columns")
        for j, k in enumerate(native):
            s0, s1, a0, a1 = int(state_off[j]), int(state_off[j + 1]), int(arc_base[j]), int(arc_base[j + 1])
            out[k] = Fst(0, arc_off[s0 + j: s1 + j + 1], arcs[a0:a1], final[s0:s1])
        if len(native) != len(texts):
            return out  # type: ignore[return-value]
        batch = FstBatch(out)
        batch.state_off, batch.arc_base, batch.arc_off, batch.arcs, batch.final = state_off, arc_base, arc_off, arcs, final
        batch.arc_off32, batch.arc_next, batch.arc_pdf = arc_off32, arc_next, arc_pdf
        batch.staging = staging
        batch.max_degree, batch.min_ilabel, batch.min_arcs = int(stats[0]), int(stats[1]), int(stats[2])
        return batch
