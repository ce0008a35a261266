"""A monophone model and a lexicon built from the gold TextGrids the reference's tests ship
(tests/data/textgrid/{acoustic_corpus,cold_corpus,cold_corpus3}.TextGrid with their wavs; the gold side of
``--reference_directory``, /root/reference/tests/conftest.py:366-375, tests/test_commandline_align.py:456-476 — copied as
data under tests/golden/ref_fixtures/).

Why: the reference's bundled ``mono_model.zip`` is an untrained plumbing model (it aligns ``acoustic_corpus.wav`` almost
uniformly: word boundaries seconds away from the gold file), so it cannot say whether boundaries are right.  The gold
TextGrids are the only reference-held artefacts that carry boundary information.  Here they supervise a small model:

  * lexicon  = every word interval of the three files with the phones of the phone tier that lie inside it (stress digits
               dropped, lower case) — dictionary data only;
  * acoustic = one Gaussian per HMM state (3-state Bakis per phone, 5-state silence), estimated from the phone tiers of
               the TRAINING files only, optionally refined by Viterbi re-alignment of those files.

Aligning the held-out file with that model and comparing with ITS gold TextGrid checks the whole chain — MFCC, CMVN, deltas,
scoring, beam Viterbi, SplitToPhones, word grouping — against boundaries the reference holds.  What it cannot pin is the
arithmetic of kalpy itself (the model is ours): "parity unpinned" stays, narrowed to "boundaries within tens of ms of the
reference's gold alignment on held-out audio".  Test-side code; nothing here is product code."""
from __future__ import annotations

import re
from pathlib import Path
from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np

import synth_workload as S
from montreal_forced_aligner_amd import ctm as C
from montreal_forced_aligner_amd import graph as G
from montreal_forced_aligner_amd import kaldi_io as K

REF = Path(__file__).resolve().parent / "golden" / "ref_fixtures"
NAMES = ("acoustic_corpus", "cold_corpus", "cold_corpus3")
SR = 16000


def _phone(label: str) -> str:
    return re.sub(r"\d", "", label).lower()


class Gold:
    """One gold file: PCM, word tier, phone tier (labels normalised), and the transcript its word tier spells."""

    def __init__(self, name: str):
        self.name = name
        pcm, sr = K.read_wav_pcm16(REF / f"{name}.wav")
        assert sr == SR
        self.pcm = pcm[0]
        tg = C.read_short_textgrid(REF / f"{name}.TextGrid")
        self.words = [(b, e, w) for b, e, w in tg["words"] if w != ""]
        self.phones = [(b, e, _phone(p)) for b, e, p in tg["phones"] if p != ""]
        self.text = " ".join(w for _, _, w in self.words)

    def pronunciations(self) -> List[Tuple[str, Tuple[str, ...]]]:
        out = []
        for b, e, w in self.words:
            ph = tuple(p for pb, pe, p in self.phones if pb >= b - 1e-6 and pe <= e + 1e-6)
            if ph:
                out.append((w, ph))
        return out

    def segments(self) -> List[Tuple[str, int, int]]:
        """(phone, first sample, end sample) covering the file; everything outside a phone interval is silence."""
        segs, pos = [], 0
        for b, e, p in self.phones:
            a, z = int(round(b * SR)), int(round(e * SR))
            if a > pos:
                segs.append(("sil", pos, a))
            segs.append((p, a, z))
            pos = z
        if pos < len(self.pcm):
            segs.append(("sil", pos, len(self.pcm)))
        return segs


def build_lexicon(golds: Sequence[Gold]) -> G.LexiconCompiler:
    lex = G.LexiconCompiler(position_dependent_phones=False, silence_phone="sil", oov_phone="spn")
    seen = set()
    for g in golds:
        for w, ph in g.pronunciations():
            if (w, ph) not in seen:
                seen.add((w, ph))
                lex.add_pronunciation(G.Pronunciation(w, " ".join(ph)))
    lex.build_phone_table()
    return lex


def state_labels(lex: G.LexiconCompiler, segs, n_frames: int) -> List[Tuple[int, int]]:
    class _W:          # synth_workload.state_labels only needs `.lexicon`
        lexicon = lex
    return S.state_labels(_W, segs, n_frames)


def accumulate(stats: Dict, x: np.ndarray, labels: Sequence[Tuple[int, int]]) -> None:
    keys = np.array([p * 8 + s for p, s in labels])
    for k in np.unique(keys):
        rows = x[keys == k].astype(np.float64)
        st = stats.setdefault((int(k) // 8, int(k) % 8), [0.0, 0.0, 0])
        st[0] = st[0] + rows.sum(axis=0)
        st[1] = st[1] + (rows * rows).sum(axis=0)
        st[2] += rows.shape[0]


def labels_from_alignment(tm, ali: np.ndarray) -> List[Tuple[int, int]]:
    """(phone id, hmm state) per frame of a transition-id alignment."""
    st = tm.id2state[np.asarray(ali)]
    return [(int(tm.tuples[s - 1][0]), int(tm.tuples[s - 1][1])) for s in st]


def train(train_golds: Sequence[Gold], lex: G.LexiconCompiler, feature_fn: Callable[[np.ndarray], np.ndarray],
          align_fn: Callable = None, realign_iters: int = 0):
    """``feature_fn(pcm) -> [T, 39]``; ``align_fn(model, gold, feats) -> transition-ids or None`` for the re-alignment
    rounds.  Returns a synth_workload.SynthModel."""
    feats = [feature_fn(g.pcm) for g in train_golds]
    stats: Dict = {}
    for g, x in zip(train_golds, feats):
        accumulate(stats, x, state_labels(lex, g.segments(), x.shape[0]))
    model = S.monophone_from_stats(lex, stats)
    for _ in range(realign_iters):
        stats = {}
        for g, x in zip(train_golds, feats):
            ali = align_fn(model, g, x)
            labels = labels_from_alignment(model.tm, ali) if ali is not None else state_labels(lex, g.segments(), x.shape[0])
            accumulate(stats, x, labels)
        model = S.monophone_from_stats(lex, stats)
    return model


def boundary_report(gold: Gold, word_intervals, phone_intervals) -> Dict[str, np.ndarray]:
    """Absolute differences (seconds) between aligned and gold boundaries; sequences must already agree."""
    mine_w = [(w.begin, w.end, w.label) for w in word_intervals]
    assert [m[2] for m in mine_w] == [g[2] for g in gold.words], "word sequence differs from the gold tier"
    mine_p = [(p.begin, p.end, str(p.label)) for p in phone_intervals]
    out = {
        "word_begin": np.array([m[0] - g[0] for m, g in zip(mine_w, gold.words)]),
        "word_end": np.array([m[1] - g[1] for m, g in zip(mine_w, gold.words)]),
    }
    if [m[2] for m in mine_p] == [g[2] for g in gold.phones]:
        out["phone_begin"] = np.array([m[0] - g[0] for m, g in zip(mine_p, gold.phones)])
        out["phone_end"] = np.array([m[1] - g[1] for m, g in zip(mine_p, gold.phones)])
    return out
