"""Lexicon and per-utterance training-graph construction (host side; SURVEY §8 row a5 / "next" N1).

Reference boundary: ``kalpy.fstext.lexicon.LexiconCompiler`` as configured at
MFA/dictionary/multispeaker.py:3105-3225 and ``kalpy.decoder.training_graphs.TrainingGraphCompiler`` as
called at MFA/alignment/multiprocessing.py:537-571 and MFA/online/alignment.py:77-96
(``compile_fst(text)``).  The reference builds H∘C∘L∘G with OpenFst (TableCompose → DeterminizeStarInLog →
MinimizeEncoded → AddSelfLoops(reorder=true)); OpenFst is not available here, so this module constructs an
*equivalent* (same weighted path set, not isomorphic) graph directly:

  words → phone graph (optional silence, pronunciation variants; Kaldi make_lexicon_fst[_silprob] structure)
        → context expansion (decision-tree window N, central position P)
        → HMM expansion with forward transition-ids
        → Kaldi AddSelfLoopsReorder semantics: states split so every incoming arc has one transition-state,
          self-loop arc appended last.

Transition probabilities are NOT in the compiled graph (compile-time scales are 0, as in the reference);
``add_transition_probs`` applies them at align time (Appendix A.8).  The output ``Fst`` is epsilon-free.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .kaldi_io import ARC_DTYPE, ContextDependency, Fst
from .model import TransitionModel


class SymbolTable:
    """Minimal stand-in for pywrapfst.SymbolTable as the reference uses it (member/find/add_symbol)."""

    def __init__(self, symbols: Optional[Iterable[str]] = None):
        self._s2i: Dict[str, int] = {}
        self._i2s: Dict[int, str] = {}
        for s in symbols or ():
            self.add_symbol(s)

    def add_symbol(self, sym: str, key: Optional[int] = None) -> int:
        if sym in self._s2i:
            return self._s2i[sym]
        if key is None:
            key = (max(self._i2s) + 1) if self._i2s else 0
        self._s2i[sym] = key
        self._i2s[key] = sym
        return key

    def member(self, sym) -> bool:
        return (sym in self._s2i) if isinstance(sym, str) else (sym in self._i2s)

    def find(self, key):
        if isinstance(key, str):
            return self._s2i.get(key, -1)
        return self._i2s.get(int(key), "")

    def num_symbols(self) -> int:
        return len(self._s2i)

    def __iter__(self):
        return iter(sorted(self._i2s.items()))

    @classmethod
    def read_text(cls, text: str) -> "SymbolTable":
        t = cls()
        for line in text.splitlines():
            parts = line.split()
            if len(parts) == 2:
                t.add_symbol(parts[0], int(parts[1]))
        return t


@dataclass
class Pronunciation:
    """Mirror of kalpy's KalpyPronunciation record (MFA/dictionary/multispeaker.py:3192-3202)."""

    orthography: str
    pronunciation: str
    probability: Optional[float] = None
    silence_after_probability: Optional[float] = None
    silence_before_correction: Optional[float] = None
    non_silence_before_correction: Optional[float] = None
    disambiguation: Optional[int] = None


def _cost(p: Optional[float]) -> float:
    if p is None:
        return 0.0
    return -math.log(p) if p > 0 else float("inf")


class LexiconCompiler:
    """Pronunciation lexicon with optional-silence model.

    Constructor keywords follow the reference call (MFA/dictionary/multispeaker.py:3122-3135).
    """

    def __init__(
        self,
        disambiguation: bool = False,
        silence_probability: float = 0.5,
        initial_silence_probability: float = 0.5,
        final_silence_correction: Optional[float] = None,
        final_non_silence_correction: Optional[float] = None,
        silence_word: str = "<eps>",
        oov_word: str = "<unk>",
        silence_phone: str = "sil",
        oov_phone: str = "spn",
        position_dependent_phones: bool = False,
        ignore_case: bool = True,
        phones: Optional[Iterable[str]] = None,
        share_suffixes: bool = True,
    ):
        self.disambiguation = disambiguation
        self.share_suffixes = share_suffixes  # merge indistinguishable states of the per-utterance phone graph
        self.silence_probability = silence_probability
        self.initial_silence_probability = initial_silence_probability
        self.final_silence_correction = final_silence_correction
        self.final_non_silence_correction = final_non_silence_correction
        self.silence_word = silence_word
        self.oov_word = oov_word
        self.silence_phone = silence_phone
        self.oov_phone = oov_phone
        self.position_dependent_phones = position_dependent_phones
        self.ignore_case = ignore_case
        self.phones = set(phones) if phones else set()
        self._fixed_inventory = bool(phones)  # model-defined phone set: pronunciations outside it are skipped
        self.pronunciations: List[Pronunciation] = []
        self.word_table = SymbolTable([silence_word, oov_word])
        self.phone_table: Optional[SymbolTable] = None
        self._by_word: Dict[str, List[Pronunciation]] = {}

    # -- tables -----------------------------------------------------------------------------------
    def build_phone_table(self, silence_phones: Optional[Sequence[str]] = None) -> SymbolTable:
        """MFA's phone-table order (tests/data/dictionaries/expected/phones.txt): <eps>, silence phones (each with
        its _B/_E/_I/_S variants when position dependent), then sorted non-silence phones.

        ``silence_phones`` lists the silence-topology phones in table order; default [silence_phone, oov_phone]."""
        t = SymbolTable(["<eps>"])
        sil = list(silence_phones) if silence_phones else [self.silence_phone, self.oov_phone]
        for need in (self.silence_phone, self.oov_phone):
            if need not in sil:
                sil.append(need)
        pos = ["", "_B", "_E", "_I", "_S"] if self.position_dependent_phones else [""]
        for s in sil:
            for p in pos:
                t.add_symbol(s + p)
        pos = ["_B", "_E", "_I", "_S"] if self.position_dependent_phones else [""]
        for ph in sorted(self.phones - set(sil)):
            for p in pos:
                t.add_symbol(ph + p)
        self.phone_table = t
        return t

    @property
    def silence_symbols(self) -> List[int]:
        """Phone ids of the optional-silence phone and its position variants (used by boost_silence)."""
        out = []
        for suffix in ("", "_B", "_E", "_I", "_S"):
            k = self.phone_table.find(self.silence_phone + suffix)
            if k != -1:
                out.append(k)
        return out

    def load_pronunciations(self, path) -> None:
        """Plain MFA dictionary: ``word [prob [silprobs…]] phone phone …`` (numeric columns optional)."""
        with open(path, encoding="utf8") as f:
            for line in f:
                parts = line.strip().split()
                if len(parts) < 2:
                    continue
                word, rest = parts[0], parts[1:]
                nums = []
                while rest and _is_float(rest[0]) and len(nums) < 4:
                    nums.append(float(rest.pop(0)))
                if not rest:
                    continue
                nums += [None] * (4 - len(nums))
                self.add_pronunciation(Pronunciation(word, " ".join(rest), *nums))

    def add_pronunciation(self, p: Pronunciation) -> None:
        w = p.orthography.lower() if self.ignore_case else p.orthography
        p.orthography = w
        for prev in self._by_word.get(w, ()):
            if prev.pronunciation == p.pronunciation:
                return
        if self._fixed_inventory:
            # MFA/dictionary/multispeaker.py:3178-3184: pronunciations with phones outside the table are dropped
            known = self.phones | {self.silence_phone, self.oov_phone}
            if any(ph not in known for ph in p.pronunciation.split()):
                return
        self.pronunciations.append(p)
        self._by_word.setdefault(w, []).append(p)
        self.word_table.add_symbol(w)
        for ph in p.pronunciation.split():
            if ph not in (self.silence_phone, self.oov_phone):
                self.phones.add(ph)

    def create_fsts(self) -> None:  # kept for API symmetry with kalpy; graphs are built per utterance
        if self.phone_table is None:
            self.build_phone_table()

    def to_int(self, word: str) -> int:
        w = word.lower() if self.ignore_case else word
        k = self.word_table.find(w)
        return k if k != -1 else self.word_table.find(self.oov_word)

    def word_pronunciations(self, word: str) -> List[Pronunciation]:
        w = word.lower() if self.ignore_case else word
        if w in self._by_word:
            return self._by_word[w]
        return [Pronunciation(self.oov_word, self.oov_phone)]

    def phone_ids(self, pron: Pronunciation) -> List[int]:
        phones = pron.pronunciation.split()
        out = []
        for i, ph in enumerate(phones):
            if self.position_dependent_phones:
                if len(phones) == 1:
                    suffix = "_S"
                elif i == 0:
                    suffix = "_B"
                elif i == len(phones) - 1:
                    suffix = "_E"
                else:
                    suffix = "_I"
                sym = ph + suffix
            else:
                sym = ph
            k = self.phone_table.find(sym)
            if k == -1:
                raise KeyError(f"phone {sym!r} not in phone table")
            out.append(k)
        return out

    # -- the lexicon as one transducer (text form) ---------------------------------------------------
    def lexicon_fst_text(self) -> str:
        """L.fst in OpenFst text form, the structure kalpy's LexiconCompiler / Kaldi's make_lexicon_fst_silprob build and
        the reference keeps an expected copy of (tests/data/dictionaries/expected/lexicon.text.fst): state 0 start, state 1
        the loop state (final), state 2 the silence state;  0→1 on <eps> (−ln(1−p_init)) or on the silence phone
        (−ln p_init);  2→1 on the silence phone;  every pronunciation leaves the loop state with its word as output on its
        first phone and returns to the loop state (−ln(1−p_sil)) or to the silence state (−ln p_sil) on its last.
        ``phone_graph`` is this transducer composed with a linear transcript (the ε arc folded into its successors);
        writing it out lets the construction be checked against the reference's own expected file."""
        def w(cost: float) -> str:
            return "" if cost == 0.0 else "\t" + repr(float(cost))

        sil = self.silence_phone
        lines = [f"0\t1\t<eps>\t<eps>{w(_cost(1.0 - self.initial_silence_probability))}",
                 f"0\t1\t{sil}\t<eps>{w(_cost(self.initial_silence_probability))}",
                 f"2\t1\t{sil}\t<eps>"]
        nxt = 3
        for word, prons in self._by_word.items():
            no_probs = all(p.probability is None for p in prons)
            for p in prons:
                syms = [self.phone_table.find(k) for k in self.phone_ids(p)]
                pc = 0.0 if no_probs else _cost(p.probability if p.probability is not None else 1.0)
                p_after = p.silence_after_probability if p.silence_after_probability is not None else self.silence_probability
                cur = 1
                for k, ph in enumerate(syms):
                    ol = word if k == 0 else "<eps>"
                    c = pc if k == 0 else 0.0
                    if k < len(syms) - 1:
                        lines.append(f"{cur}\t{nxt}\t{ph}\t{ol}{w(c)}")
                        cur = nxt
                        nxt += 1
                    else:
                        lines.append(f"{cur}\t1\t{ph}\t{ol}{w(c + _cost(1.0 - p_after))}")
                        if p_after > 0:
                            lines.append(f"{cur}\t2\t{ph}\t{ol}{w(c + _cost(p_after))}")
        lines.append("1\t0")
        return "\n".join(lines) + "\n"

    # -- phone-level graph for one transcript -------------------------------------------------------
    def phone_graph(self, words: Sequence[str]) -> "PhoneGraph":
        """Linear transcript composed with the lexicon (Kaldi make_lexicon_fst[_silprob] structure):

        per word boundary i two nodes NS_i (no silence just before) / S_i (silence just consumed);
        start → NS_0 (ε, −ln(1−p_init)) and start —sil→ S_0 (−ln p_init); each pronunciation leaves NS_i / S_i
        (cost −ln prob − ln before-correction), its last phone enters NS_{i+1} (−ln(1−p_after)) or a node that
        emits the silence phone into S_{i+1} (−ln p_after).  The ε arc at the start is folded into its successors.
        """
        g = PhoneGraph()
        sil = self.phone_table.find(self.silence_phone)
        n = len(words)
        NS = [g.add_node() for _ in range(n + 1)]
        S = [g.add_node() for _ in range(n + 1)]
        start = g.add_node()
        g.start = start
        p_init = self.initial_silence_probability
        eps_cost = _cost(1.0 - p_init)
        g.add_arc(start, S[0], sil, 0, _cost(p_init))
        for i, w in enumerate(words):
            wid = self.to_int(w)
            prons = self.word_pronunciations(w)
            no_probs = all(p.probability is None for p in prons)
            for p in prons:
                ids = self.phone_ids(p)
                pc = 0.0 if no_probs else _cost(p.probability if p.probability is not None else 1.0)
                p_after = p.silence_after_probability if p.silence_after_probability is not None else self.silence_probability
                sources = [(NS[i], pc + _cost(p.non_silence_before_correction))]
                sources.append((S[i], pc + _cost(p.silence_before_correction)))
                if i == 0:
                    sources.append((start, eps_cost + pc + _cost(p.non_silence_before_correction)))
                for src, c0 in sources:
                    cur = src
                    for k, ph in enumerate(ids):
                        first, last = k == 0, k == len(ids) - 1
                        ol = wid if first else 0
                        c = c0 if first else 0.0
                        if not last:
                            nxt = g.add_node()
                            g.add_arc(cur, nxt, ph, ol, c)
                            cur = nxt
                        else:
                            g.add_arc(cur, NS[i + 1], ph, ol, c + _cost(1.0 - p_after))
                            if p_after > 0:
                                mid = g.add_node()
                                g.add_arc(cur, mid, ph, ol, c + _cost(p_after))
                                g.add_arc(mid, S[i + 1], sil, 0, 0.0)
        g.final[NS[n]] = _cost(self.final_non_silence_correction)
        g.final[S[n]] = _cost(self.final_silence_correction)
        if n == 0:
            g.final[start] = eps_cost + g.final[NS[0]]
        g.trim()
        if self.share_suffixes:
            g.merge_suffixes()
        return g


def _is_float(s: str) -> bool:
    try:
        float(s)
        return True
    except ValueError:
        return False


@dataclass
class PhoneGraph:
    start: int = 0
    arcs: List[List[Tuple[int, int, int, float]]] = field(default_factory=list)  # per node: (dst, phone, olabel, w)
    final: Dict[int, float] = field(default_factory=dict)

    def add_node(self) -> int:
        self.arcs.append([])
        return len(self.arcs) - 1

    def add_arc(self, src: int, dst: int, phone: int, olabel: int, w: float) -> None:
        self.arcs[src].append((dst, phone, olabel, w))

    def trim(self) -> None:
        """Drop nodes that are not both accessible and co-accessible; renumber in BFS order from start."""
        n = len(self.arcs)
        fwd = [False] * n
        stack = [self.start]
        fwd[self.start] = True
        while stack:
            u = stack.pop()
            for (v, *_r) in self.arcs[u]:
                if not fwd[v]:
                    fwd[v] = True
                    stack.append(v)
        rev: List[List[int]] = [[] for _ in range(n)]
        for u in range(n):
            for (v, *_r) in self.arcs[u]:
                rev[v].append(u)
        bwd = [False] * n
        stack = [u for u in self.final if math.isfinite(self.final[u])]
        for u in stack:
            bwd[u] = True
        while stack:
            u = stack.pop()
            for v in rev[u]:
                if not bwd[v]:
                    bwd[v] = True
                    stack.append(v)
        keep = [fwd[i] and bwd[i] for i in range(n)]
        order: List[int] = []
        new_id = [-1] * n
        if keep[self.start]:
            queue = [self.start]
            new_id[self.start] = 0
            order.append(self.start)
            qi = 0
            while qi < len(queue):
                u = queue[qi]
                qi += 1
                for (v, *_r) in self.arcs[u]:
                    if keep[v] and new_id[v] == -1:
                        new_id[v] = len(order)
                        order.append(v)
                        queue.append(v)
        arcs = []
        for u in order:
            arcs.append([(new_id[v], ph, ol, w) for (v, ph, ol, w) in self.arcs[u] if keep[v]])
        self.final = {new_id[u]: w for u, w in self.final.items() if keep[u] and new_id[u] != -1 and math.isfinite(w)}
        self.arcs = arcs
        self.start = 0

    def merge_suffixes(self) -> None:
        """Merge nodes whose futures are identical (same final cost, same outgoing (phone, word, cost, successor) set).

        The chains a pronunciation hangs off NS_i and off S_i (and off the start) differ only in their first arc — the
        source-dependent cost sits there — so from the second node on they are the same automaton; the reference's
        pipeline removes the copies with MinimizeEncoded after DeterminizeStarInLog (Kaldi training-graph-compiler.cc,
        reached through kalpy TrainingGraphCompiler).  The graph is acyclic (phones only move forward through the
        transcript), so one pass in reverse topological order with hash-consing of (final, arcs) is a full backward
        minimisation.  Every path keeps its labels and its cost bit for bit; only states nobody can tell apart go."""
        n = len(self.arcs)
        indeg = [0] * n
        for a in self.arcs:
            for (v, *_r) in a:
                indeg[v] += 1
        topo = [u for u in range(n) if indeg[u] == 0]
        qi = 0
        while qi < len(topo):
            u = topo[qi]
            qi += 1
            for (v, *_r) in self.arcs[u]:
                indeg[v] -= 1
                if indeg[v] == 0:
                    topo.append(v)
        if len(topo) != n:
            return          # a cycle (caller-built graph): leave it alone
        rep = list(range(n))
        seen: Dict[tuple, int] = {}
        for u in reversed(topo):
            arcs = []
            for (v, ph, ol, w) in self.arcs[u]:
                a = (rep[v], ph, ol, w)
                if a not in arcs:
                    arcs.append(a)
            self.arcs[u] = arcs
            sig = (self.final.get(u), tuple(sorted(arcs)))
            rep[u] = seen.setdefault(sig, u)
        self.start = rep[self.start]
        self.trim()


class TrainingGraphCompiler:
    """``TrainingGraphCompiler(model_path, tree_path, lexicon_compiler, use_g2p=False, batch_size=…)`` with
    ``.compile_fst(text)`` (MFA/online/alignment.py:77-96; MFA/alignment/multiprocessing.py:537-571).

    Here the model and tree are passed as already-parsed objects (``TransitionModel``, ``ContextDependency``).
    """

    def __init__(self, transition_model: TransitionModel, tree: ContextDependency, lexicon_compiler: LexiconCompiler,
                 use_g2p: bool = False, batch_size: int = 500, determinize: bool = True):
        # determinize: DeterminizeStarInLog + MinimizeEncoded between the HMM expansion and AddSelfLoops, as Kaldi's
        # TrainingGraphCompiler::CompileGraph runs them (determinize_star_log / minimize_encoded below); False keeps the
        # round-2 graphs (phone-level suffix sharing only).
        self.determinize = bool(determinize)
        if use_g2p:
            raise NotImplementedError("use_g2p (character transcripts through a G2P lexicon) is outside the alignment path built here")
        self.tm = transition_model
        self.tree = tree
        self.lexicon = lexicon_compiler
        self.use_g2p = False
        self.batch_size = batch_size
        if lexicon_compiler.phone_table is None:
            lexicon_compiler.build_phone_table()
        if tree.context_width not in (1, 3) or tree.central_position != tree.context_width // 2:
            raise NotImplementedError(
                f"context width {tree.context_width} / central position {tree.central_position} is not supported"
            )
        self._tuple2ts: Dict[Tuple[int, int, int, int], int] = {
            tuple(int(x) for x in row): i + 1 for i, row in enumerate(self.tm.tuples)
        }
        self._hmm_cache: Dict[Tuple[int, ...], List[Tuple[int, int, int]]] = {}

    # transitions of one context-dependent phone instance: list of (src_hs, dst_hs, tid) for non-self transitions
    def _hmm(self, window: Tuple[int, ...]):
        if window in self._hmm_cache:
            return self._hmm_cache[window]
        phone = window[self.tree.central_position]
        entry = self.tm.topo.entry_for_phone(phone)
        trans = []
        for hs, st in enumerate(entry):
            if not st.transitions:
                continue
            fwd = self.tree.compute(list(window), st.forward_pdf_class)
            slf = self.tree.compute(list(window), st.self_loop_pdf_class)
            key = (phone, hs, fwd, slf)
            if key not in self._tuple2ts:
                raise KeyError(f"(phone,hmm-state,pdf) {key} is not a transition-state of the model")
            ts = self._tuple2ts[key]
            for k, (dst, _p) in enumerate(st.transitions):
                if dst != hs:
                    trans.append((hs, dst, int(self.tm.state2id[ts]) + k))
                    if dst == 0:
                        raise NotImplementedError("topologies that re-enter HMM state 0 are not supported")
        n_final = len(entry) - 1
        self._hmm_cache[window] = (trans, n_final)
        return self._hmm_cache[window]

    def compile_fst(self, text: str) -> Fst:
        words = text.split()
        pg = self.lexicon.phone_graph(words)
        cg = _expand_context(pg, self.tree.context_width)
        return self._expand_hmm(cg)

    def compile_fsts(self, texts: Sequence[str], scaled_log_probs: Optional[np.ndarray] = None,
                     n_threads: Optional[int] = None, columns: bool = False, alloc=None) -> List[Fst]:
        """``[compile_fst(t) for t in texts]`` (with ``scaled_log_probs``: ``add_transition_probs`` applied) for a whole
        batch through the native compiler — csrc/graph_compile.cpp, one utterance per worker thread, the same graphs bit
        for bit (tests/test_graph_native_cpu.py).  What kalpy's C++ ``export_graphs`` is to the reference
        (MFA/alignment/multiprocessing.py:537-571)."""
        from . import graph_native

        nat = getattr(self, "_native", None)
        if nat is None:
            nat = self._native = graph_native.NativeGraphCompiler(self, n_threads)
        elif n_threads is not None:
            nat.n_threads = max(1, int(n_threads))
        return nat.compile_batch(list(texts), scaled_log_probs, columns=columns, alloc=alloc)

    def compile_phone_graph(self, pg: PhoneGraph) -> Fst:
        return self._expand_hmm(_expand_context(pg, self.tree.context_width))

    def _expand_hmm(self, cg: "CtxGraph") -> Fst:
        """HMM expansion + AddSelfLoopsReorder.

        Every graph state is keyed (junction-or-internal node, incoming transition-state); the key's
        transition-state decides the one self-loop appended after the state's forward arcs."""
        tm = self.tm
        # phase 1: "G0" = forward transitions only.  nodes: junctions 0..J-1, then per (ctx arc, hmm state>0).
        J = cg.num_nodes
        g0_arcs: List[List[Tuple[int, int, int, float]]] = [[] for _ in range(J)]  # (dst, tid, olabel, w)

        def new_node() -> int:
            g0_arcs.append([])
            return len(g0_arcs) - 1

        for u in range(J):
            for (v, window, ol, w) in cg.arcs[u]:
                trans, n_final = self._hmm(window)
                node_of = {0: u, n_final: v}
                for (hs, dst, tid) in trans:
                    for s in (hs, dst):
                        if s not in node_of:
                            node_of[s] = new_node()
                for (hs, dst, tid) in trans:
                    first = hs == 0
                    g0_arcs[node_of[hs]].append((node_of[dst], tid, ol if first else 0, w if first else 0.0))
        g0_final = {u: w for u, w in cg.final.items() if u < J}
        g0_start = cg.start
        if self.determinize:
            det = determinize_star_log(g0_arcs, g0_final, g0_start)
            if det is not None:                 # (None: an output string longer than one label would be needed — kept as it was)
                g0_arcs, g0_final = minimize_encoded(*det)
                g0_start = 0
        # phase 2: split states by incoming transition-state (MakePrecedingInputSymbolsSameClass) and add self loops.
        key2id: Dict[Tuple[int, int], int] = {(g0_start, 0): 0}
        order: List[Tuple[int, int]] = [(g0_start, 0)]
        out_arcs: List[List[Tuple[int, int, float, int]]] = []
        finals: List[float] = []
        qi = 0
        while qi < len(order):
            node, ts_in = order[qi]
            qi += 1
            arcs = []
            for (dst, tid, ol, w) in g0_arcs[node]:
                k = (dst, int(tm.id2state[tid]))
                if k not in key2id:
                    key2id[k] = len(order)
                    order.append(k)
                arcs.append((tid, ol, w, key2id[k]))
            if ts_in > 0:
                sl = int(tm.self_loop_of[ts_in])
                if sl != 0:
                    arcs.append((sl, 0, 0.0, key2id[(node, ts_in)]))
            out_arcs.append(arcs)
            finals.append(g0_final.get(node, float("inf")))
        S = len(out_arcs)
        offs = np.zeros(S + 1, dtype=np.int64)
        for s in range(S):
            offs[s + 1] = offs[s] + len(out_arcs[s])
        arcs_np = np.zeros(int(offs[-1]), dtype=ARC_DTYPE)
        k = 0
        for s in range(S):
            for (il, ol, w, nx) in out_arcs[s]:
                arcs_np[k] = (il, ol, w, nx)
                k += 1
        return Fst(0, offs, arcs_np, np.asarray(finals, dtype=np.float32))


KALDI_DELTA = 1.0 / 1024.0          # fst::kDelta: weight equality in DeterminizeStar, quantum of MinimizeEncoded


def _log_add(a: float, b: float) -> float:
    """⊕ of the log semiring: −ln(e^−a + e^−b)."""
    if a == float("inf"):
        return b
    if b == float("inf"):
        return a
    return (a if a < b else b) - math.log1p(math.exp(-abs(a - b)))


def determinize_star_log(arcs, final, start, delta: float = KALDI_DELTA):
    """Kaldi DeterminizeStarInLog on the forward-transition graph (fstext/determinize-star-inl.h, called from
    TrainingGraphCompiler::CompileGraph — reached through kalpy's TrainingGraphCompiler, MFA/alignment/multiprocessing.py:
    537-571): weighted subset construction over the input labels (transition-ids) in the LOG semiring.  A subset is a list of
    (state, residual output string, residual weight); an input label's arc carries ⊕ of its elements' weights and the
    longest common prefix of their output strings, the elements keep the remainders; elements that meet in the same state
    with the same string are ⊕-added; subsets are identified up to ``delta`` on their weights.  The input here has no
    ε input labels, so the "star" (ε-closure) part of the algorithm has nothing to do.

    arcs: per node a list of (dst, transition-id, output label or 0, weight); final: {node: weight}.  Returns
    (arcs, final) of the deterministic graph, start state 0, states numbered in order of discovery (breadth first), arcs of a
    state by ascending transition-id — or None when an arc would have to emit more than one output label (then ε-input arcs
    would be needed, as DeterminizeStar creates them) or a residual string grows beyond two labels; transcripts through a
    lexicon never get there (a word's label sits on its first phone and is agreed on at once)."""
    inf = float("inf")
    first = ((start, (), 0.0),)
    ids = {((start, ()),): [(first, 0)]}          # (state, string) signature → [(subset, id)] (weights compared up to delta)
    subsets = [first]
    out_arcs: List[List[Tuple[int, int, int, float]]] = []
    out_final: Dict[int, float] = {}
    qi = 0
    while qi < len(subsets):
        P = subsets[qi]
        qi += 1
        fin = inf
        by_label: Dict[int, List[Tuple[int, tuple, float]]] = {}
        for (q, s, w) in P:
            fq = final.get(q)
            if fq is not None and fq != inf:
                if s:
                    return None
                fin = _log_add(fin, w + fq)
            for (dst, tid, ol, aw) in arcs[q]:
                s2 = s + (ol,) if ol else s
                if len(s2) > 2:
                    return None
                by_label.setdefault(tid, []).append((dst, s2, w + aw))
        if fin != inf:
            out_final[qi - 1] = fin
        row = []
        for tid in sorted(by_label):
            elems: Dict[Tuple[int, tuple], float] = {}
            for (dst, s2, w2) in by_label[tid]:          # same state, same string: ⊕
                k = (dst, s2)
                elems[k] = _log_add(elems[k], w2) if k in elems else w2
            keys = sorted(elems)
            tot = inf
            for k in keys:
                tot = _log_add(tot, elems[k])
            strings = [k[1] for k in keys]
            n_common = min(len(x) for x in strings)
            for j in range(n_common):
                if any(x[j] != strings[0][j] for x in strings):
                    n_common = j
                    break
            if n_common > 1:
                return None
            subset = tuple((k[0], k[1][n_common:], elems[k] - tot) for k in keys)
            sig = tuple((q, x) for (q, x, _w) in subset)
            found = None
            for cand, cid in ids.get(sig, ()):
                if all(abs(a[2] - b[2]) <= delta for a, b in zip(cand, subset)):
                    found = cid
                    break
            if found is None:
                found = len(subsets)
                subsets.append(subset)
                ids.setdefault(sig, []).append((subset, found))
            row.append((found, tid, strings[0][0] if n_common == 1 else 0, tot))
        out_arcs.append(row)
    return out_arcs, out_final


def _sccs(arcs) -> List[List[int]]:
    """Strongly connected components, successors before predecessors (Tarjan, iterative)."""
    n = len(arcs)
    index = [-1] * n
    low = [0] * n
    on = [False] * n
    stack: List[int] = []
    out: List[List[int]] = []
    counter = 0
    for root in range(n):
        if index[root] != -1:
            continue
        work = [(root, 0)]
        index[root] = low[root] = counter
        counter += 1
        stack.append(root)
        on[root] = True
        while work:
            u, k = work[-1]
            if k < len(arcs[u]):
                work[-1] = (u, k + 1)
                v = arcs[u][k][0]
                if index[v] == -1:
                    index[v] = low[v] = counter
                    counter += 1
                    stack.append(v)
                    on[v] = True
                    work.append((v, 0))
                elif on[v]:
                    low[u] = min(low[u], index[v])
            else:
                work.pop()
                if work:
                    p = work[-1][0]
                    low[p] = min(low[p], low[u])
                if low[u] == index[u]:
                    comp = []
                    while True:
                        v = stack.pop()
                        on[v] = False
                        comp.append(v)
                        if v == u:
                            break
                    out.append(comp)
    return out


def minimize_encoded(arcs, final, delta: float = KALDI_DELTA):
    """Kaldi MinimizeEncoded (fstext/fstext-utils-inl.h): weights quantised to multiples of ``delta`` (QuantizeMapper:
    floor(w / delta + 0.5) · delta), every (input label, output label, weight) triple read as one symbol, and the resulting
    unweighted acceptor minimised — states with the same final weight and the same set of (symbol, successor class) are one
    state.  The graph is acyclic except inside silence models (MFA's 5-state silence topology moves back and forth between
    its inner states), so the classes are found bottom-up over the strongly connected components: a single state by its
    (final, arcs) signature; a cyclic component as a whole — its states ordered by their own (symbol) sets, arcs that stay
    inside written as positions — and two components with the same signature are merged state by state.  (Never merges
    states that differ; a component whose states cannot be told apart by their symbols is left alone.)
    Start state 0; states renumbered breadth first, arc order kept."""
    def quant(w: float) -> float:
        return math.floor(w / delta + 0.5) * delta

    n = len(arcs)
    arcs = [[(dst, tid, ol, quant(w)) for (dst, tid, ol, w) in row] for row in arcs]
    final = {u: quant(w) for u, w in final.items()}
    rep = list(range(n))
    seen: Dict[tuple, int] = {}
    seen_comp: Dict[tuple, List[int]] = {}
    for comp in _sccs(arcs):
        if len(comp) == 1 and all(v != comp[0] for (v, *_r) in arcs[comp[0]]):
            u = comp[0]
            row = []
            for (v, tid, ol, w) in arcs[u]:
                a = (rep[v], tid, ol, w)
                if a not in row:
                    row.append(a)
            arcs[u] = row
            sig = (final.get(u), tuple(sorted(row)))
            rep[u] = seen.setdefault(sig, u)
            continue
        inside = set(comp)
        own = {u: tuple(sorted((tid, ol, w) for (_v, tid, ol, w) in arcs[u])) for u in comp}
        ordered = sorted(comp, key=lambda u: own[u])
        pos = {u: i for i, u in enumerate(ordered)}
        for u in comp:
            arcs[u] = [(v if v in inside else rep[v], tid, ol, w) for (v, tid, ol, w) in arcs[u]]
        if len(set(own.values())) != len(comp):
            continue
        sig = tuple((final.get(u), tuple(sorted(((0, pos[v]) if v in inside else (1, v), tid, ol, w) for (v, tid, ol, w) in arcs[u])))
                    for u in ordered)
        first = seen_comp.setdefault(sig, ordered)
        if first is not ordered:
            for u, r in zip(ordered, first):
                rep[u] = r
    # renumber breadth first from the start state's class
    new_id = {rep[0]: 0}
    order = [rep[0]]
    qi = 0
    while qi < len(order):
        u = order[qi]
        qi += 1
        for (v, *_r) in arcs[u]:
            v = rep[v]
            if v not in new_id:
                new_id[v] = len(order)
                order.append(v)
    out = [[(new_id[rep[v]], tid, ol, w) for (v, tid, ol, w) in arcs[u]] for u in order]
    return out, {new_id[u]: w for u, w in final.items() if u in new_id}


@dataclass
class CtxGraph:
    num_nodes: int
    start: int
    arcs: List[List[Tuple[int, Tuple[int, ...], int, float]]]  # (dst, phone window, olabel, w)
    final: Dict[int, float]


def _expand_context(pg: PhoneGraph, width: int) -> CtxGraph:
    """Phone graph → arcs labelled with full context windows.

    width 1: identity.  width 3 (triphone): a state is (pending phone-graph arc e, left phone l); the pending
    phone c of e is emitted as window (l, c, r) when a following arc with phone r is chosen, and as (l, c, 0) at a
    final node (Kaldi's subsequential symbol is phone 0 in the window).  The word label and weight of e travel with
    its emission, so every context arc carries exactly the label/weight of the phone it emits.  The virtual start's
    ε arcs to the first pending arcs are folded into the start state.
    """
    if width == 1:
        arcs = [[(v, (ph,), ol, w) for (v, ph, ol, w) in a] for a in pg.arcs]
        return CtxGraph(len(pg.arcs), pg.start, arcs, dict(pg.final))
    # enumerate phone-graph arcs
    earcs: List[Tuple[int, int, int, int, float]] = []  # (src, dst, phone, olabel, w)
    out_of: List[List[int]] = [[] for _ in pg.arcs]
    for u, a in enumerate(pg.arcs):
        for (v, ph, ol, w) in a:
            out_of[u].append(len(earcs))
            earcs.append((u, v, ph, ol, w))
    key2id: Dict[Tuple[int, int], int] = {}
    order: List[Tuple[int, int]] = []
    arcs: List[List[Tuple[int, Tuple[int, ...], int, float]]] = [[]]  # state 0 = start
    final: Dict[int, float] = {}
    END = -1

    def sid(key):
        if key not in key2id:
            key2id[key] = len(order) + 1
            order.append(key)
            arcs.append(None)
        return key2id[key]

    def expand(e: int, l: int):
        (_u, v, c, ol, w) = earcs[e]
        out = []
        for e2 in out_of[v]:
            r = earcs[e2][2]
            out.append((sid((e2, c)), (l, c, r), ol, w))
        if v in pg.final:
            out.append((END, (l, c, 0), ol, w + pg.final[v]))
        return out

    for e in out_of[pg.start]:
        arcs[0] += expand(e, 0)
    qi = 0
    while qi < len(order):
        e, l = order[qi]
        arcs[qi + 1] = expand(e, l)
        qi += 1
    end = len(arcs)
    arcs.append([])
    final[end] = 0.0
    for a in arcs:
        for i, (dst, win, ol, w) in enumerate(a):
            if dst == END:
                a[i] = (end, win, ol, w)
    if pg.start in pg.final:
        final[0] = pg.final[pg.start]
    return CtxGraph(len(arcs), 0, arcs, final)


def add_transition_probs(fst: Fst, scaled_log_probs: np.ndarray) -> Fst:
    """Kaldi AddTransitionProbs: weight ⊗= −scaled_log_prob(tid) for every arc whose ilabel is a transition-id
    (Appendix A.8; reference values transition_scale 1.0, self_loop_scale 0.1 from MFA/alignment/mixins.py:193-203)."""
    il = fst.arcs["ilabel"]
    n_ids = scaled_log_probs.shape[0] - 1
    if np.any((il < 0) | (il > n_ids)):
        raise ValueError("AddTransitionProbs: invalid symbol on graph input side")
    arcs = fst.arcs.copy()
    add = (-scaled_log_probs.astype(np.float32))[il]
    arcs["weight"] = np.where(il > 0, (arcs["weight"] + add).astype(np.float32), arcs["weight"])
    return Fst(fst.start, fst.arc_offsets, arcs, fst.final)
