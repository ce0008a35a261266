"""Seeded synthetic corpora and acoustic models for the BASELINE.json configurations.

There is no network for datasets or checkpoints, so the benchmark and the large-size parity tests run on speech-shaped
synthetic audio: every phone is a stationary source–filter sound (3 formants, voiced = impulse train, unvoiced = noise),
words come from a seeded lexicon, and the acoustic models are *estimated from that audio* (monophone: one Gaussian per
HMM state from the generator's ground-truth segmentation; triphone: 5 context classes per side → ≈5k leaves, 32 Gaussians
per leaf) so that beam-10 alignment behaves as it does on real speech.  Seeds follow MFA's default seed 1234
(MFA/config.py:146).  Used by tests/ and bench.py only (workload generator, not part of the product path).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from montreal_forced_aligner_amd import graph as G
from montreal_forced_aligner_amd import kaldi_io as K
from montreal_forced_aligner_amd import model as M

SEED = 1234
SR = 16000
N_PHONES = 66
N_VOICES = 8
UTT_SAMPLES = 160000


def _resonator(x: np.ndarray, f: float, bw: float) -> np.ndarray:
    from scipy.signal import lfilter

    r = np.exp(-np.pi * bw / SR)
    th = 2 * np.pi * f / SR
    return lfilter([1.0 - r], [1.0, -2 * r * np.cos(th), r * r], x)


@dataclass
class SynthWorld:
    """Phone inventory, voices, lexicon."""

    phones: List[str]
    lexicon: G.LexiconCompiler
    words: List[str]
    bank: np.ndarray  # [voices, phones(+sil,spn), SR] float32 unit-RMS one-second loops
    phone_index: Dict[str, int]

    @classmethod
    def build(cls, n_words: int = 2000, seed: int = SEED) -> "SynthWorld":
        rng = np.random.default_rng(seed)
        phones = [f"p{i:02d}" for i in range(N_PHONES)]
        formants = rng.uniform(250, 3500, size=(N_PHONES, 3))
        voiced = rng.random(N_PHONES) < 0.6
        f0 = rng.uniform(80, 250, size=N_VOICES)
        vtl = rng.uniform(0.9, 1.1, size=N_VOICES)
        bank = np.zeros((N_VOICES, N_PHONES + 2, SR), dtype=np.float32)
        for v in range(N_VOICES):
            for p in range(N_PHONES):
                if voiced[p]:
                    x = np.zeros(SR + 2000)
                    period = SR / f0[v]
                    x[(np.arange(0, SR + 2000, period)).astype(int)] = 1.0
                else:
                    x = rng.normal(size=SR + 2000)
                y = sum(_resonator(x, formants[p, k] * vtl[v], 80.0) for k in range(3))[2000:]
                bank[v, p] = (y / np.sqrt(np.mean(y * y))).astype(np.float32)
            bank[v, N_PHONES] = rng.normal(size=SR).astype(np.float32) * 0.01        # sil: noise floor (−60 dB rel.)
            y = _resonator(rng.normal(size=SR + 2000), 1200.0 * vtl[v], 800.0)[2000:]  # spn: broadband noise
            bank[v, N_PHONES + 1] = (y / np.sqrt(np.mean(y * y)) * 0.3).astype(np.float32)
        lex = G.LexiconCompiler(position_dependent_phones=False, phones=phones, silence_phone="sil", oov_phone="spn")
        words = []
        for w in range(n_words):
            word = f"w{w:04d}"
            words.append(word)
            n = int(rng.integers(2, 7))
            lex.add_pronunciation(G.Pronunciation(word, " ".join(phones[i] for i in rng.integers(0, N_PHONES, n))))
            if rng.random() < 0.1:
                n = int(rng.integers(2, 7))
                lex.add_pronunciation(G.Pronunciation(word, " ".join(phones[i] for i in rng.integers(0, N_PHONES, n))))
        lex.build_phone_table()
        idx = {p: i for i, p in enumerate(phones)}
        idx["sil"] = N_PHONES
        idx["spn"] = N_PHONES + 1
        return cls(phones, lex, words, bank, idx)

    # ------------------------------------------------------------------ utterances
    def utterance(self, index: int, n_words: int = 30, samples: int = UTT_SAMPLES, speaker: Optional[int] = None):
        """Returns (pcm int16 [samples], text, segments [(phone name, start sample, end sample)], speaker id)."""
        rng = np.random.default_rng(SEED + index)
        spk = int(rng.integers(0, 1000)) if speaker is None else speaker
        voice = spk % N_VOICES
        gain = 10 ** (rng.uniform(-3, 3) / 20)
        seq: List[Tuple[str, int]] = [("sil", int(0.2 * SR))]
        text = []
        for _ in range(n_words):
            w = self.words[int(rng.integers(0, len(self.words)))]
            prons = self.lexicon.word_pronunciations(w)
            pron = prons[int(rng.integers(0, len(prons)))]
            text.append(w)
            for ph in pron.pronunciation.split():
                seq.append((ph, int(0.03 * SR) + int(rng.geometric(1.0 / (0.05 * SR)))))
            if rng.random() < 0.5:
                seq.append(("sil", int(0.05 * SR) + int(rng.geometric(1.0 / (0.1 * SR)))))
        tail = int(0.2 * SR)
        body = samples - tail
        out = np.zeros(samples, dtype=np.float32)
        segs = []
        pos = 0
        n_done_words = 0
        for ph, dur in seq:
            if pos >= body:
                break
            dur = min(dur, body - pos)
            off = int(rng.integers(0, SR - 1))
            idx = (off + np.arange(dur)) % SR
            out[pos: pos + dur] = self.bank[voice, self.phone_index[ph], idx]
            segs.append((ph, pos, pos + dur))
            pos += dur
        # words that did not fit are dropped from the transcript: recount from the segments actually emitted
        emitted = [s[0] for s in segs if s[0] != "sil"]
        kept, k = [], 0
        for w, in zip(text):
            # find which pronunciation was emitted is unnecessary: count phones of the chosen pronunciation greedily
            matched = None
            for pron in self.lexicon.word_pronunciations(w):
                ph = pron.pronunciation.split()
                if emitted[k: k + len(ph)] == ph:
                    matched = ph
                    break
            if matched is None:
                break
            kept.append(w)
            k += len(matched)
        # drop a trailing partial word's phones (turn them into silence) so transcript and audio agree
        if k < len(emitted):
            cnt = 0
            for i, (ph, a, b) in enumerate(segs):
                if ph != "sil":
                    cnt += 1
                    if cnt > k:
                        out[a:b] = self.bank[voice, N_PHONES, (np.arange(b - a)) % SR]
                        segs[i] = ("sil", a, b)
        if pos < samples:
            out[pos:] = self.bank[voice, N_PHONES, np.arange(samples - pos) % SR]
            segs.append(("sil", pos, samples))
        out += self.bank[voice, N_PHONES, (np.arange(samples) + 777) % SR]  # noise floor everywhere
        pcm = np.clip(out * (3276.8 * gain), -32768, 32767).astype(np.int16)
        return pcm, " ".join(kept), segs, spk


# ---------------------------------------------------------------------------------------------------- models
def _topology(n_phone_ids: int, sil_ids: Sequence[int]) -> K.HmmTopology:
    bakis = [K.HmmState(i, i, [(i, 0.75), (i + 1, 0.25)]) for i in range(3)] + [K.HmmState(-1, -1, [])]
    sil = [K.HmmState(0, 0, [(0, 0.25), (1, 0.25), (2, 0.25), (3, 0.25)])]
    for i in (1, 2, 3):
        sil.append(K.HmmState(i, i, [(1, 0.25), (2, 0.25), (3, 0.25), (4, 0.25)]))
    sil.append(K.HmmState(4, 4, [(4, 0.75), (5, 0.25)]))
    sil.append(K.HmmState(-1, -1, []))
    phones = np.arange(1, n_phone_ids + 1, dtype=np.int32)
    phone2idx = np.full(n_phone_ids + 1, -1, dtype=np.int32)
    for p in phones:
        phone2idx[p] = 1 if int(p) in sil_ids else 0
    return K.HmmTopology(phones, phone2idx, [bakis, sil])


def _log_probs(topo: K.HmmTopology, tuples: np.ndarray) -> np.ndarray:
    lp = [0.0]
    for phone, hs, _f, _s in tuples:
        for _dst, p in topo.entry_for_phone(int(phone))[int(hs)].transitions:
            lp.append(np.log(np.float32(p)))
    return np.asarray(lp, dtype=np.float32)


def state_labels(world: SynthWorld, segs, n_frames: int, shift: int = 160) -> List[Tuple[int, int]]:
    """Ground-truth (phone id, hmm state) per frame from the generator's sample-level segmentation (frame centre rule:
    frame t covers samples around t*shift + shift/2); a phone's frames are split evenly over its emitting states."""
    pt = world.lexicon.phone_table
    lab: List[Tuple[int, int]] = [(0, 0)] * n_frames
    centre = np.arange(n_frames) * shift + shift // 2
    for ph, a, b in segs:
        fr = np.nonzero((centre >= a) & (centre < b))[0]
        if fr.size == 0:
            continue
        n_states = 5 if ph in ("sil", "spn") else 3
        pid = pt.find(ph)
        for j, t in enumerate(fr):
            lab[t] = (pid, min(n_states - 1, j * n_states // fr.size))
    return lab


@dataclass
class SynthModel:
    tm: M.TransitionModel
    am: M.DiagGmmModel
    tree: K.ContextDependency
    lda: Optional[np.ndarray] = None           # [40, 91]
    fmllr: Optional[np.ndarray] = None         # [n_spk, 40, 41]


def _accumulate(world, feature_fn, n_train, first_index, speaker_of=None):
    """Σx, Σx², n per (phone id, state) over n_train generated utterances."""
    stats: Dict[Tuple[int, int], List] = {}
    for i in range(n_train):
        pcm, _text, segs, spk = world.utterance(first_index + i)
        x = feature_fn(pcm, spk)
        lab = state_labels(world, segs, x.shape[0])
        keys = np.array([p * 8 + s for p, s in lab])
        for k in np.unique(keys):
            rows = x[keys == k].astype(np.float64)
            st = stats.setdefault((int(k) // 8, int(k) % 8), [0.0, 0.0, 0])
            st[0] = st[0] + rows.sum(axis=0)
            st[1] = st[1] + (rows * rows).sum(axis=0)
            st[2] += rows.shape[0]
    return stats


def _gauss(mean, var, weight=1.0):
    inv = 1.0 / var
    gc = np.log(weight) - 0.5 * (mean.shape[0] * np.log(2 * np.pi) + np.log(var).sum() + (mean * mean * inv).sum())
    return np.float32(gc), (mean * inv).astype(np.float32), inv.astype(np.float32)


def monophone_from_stats(lexicon: G.LexiconCompiler, stats: Dict[Tuple[int, int], List], silence_names=("sil", "spn")) -> SynthModel:
    """3-state Bakis per phone + 5-state silence topology, one Gaussian per state from Σx, Σx², n per (phone id, state)
    (variance floor 0.01; states with fewer than 5 frames take the global mean and variance)."""
    pt = lexicon.phone_table
    ids = [k for k, s in pt if s != "<eps>"]
    sil_ids = [pt.find(s) for s in silence_names if pt.find(s) != -1]
    topo = _topology(max(ids), sil_ids)
    dim = next(iter(stats.values()))[0].shape[0]
    glob = [sum(s[0] for s in stats.values()), sum(s[1] for s in stats.values()), sum(s[2] for s in stats.values())]
    gmean = glob[0] / glob[2]
    gvar = np.maximum(glob[1] / glob[2] - gmean * gmean, 0.01)
    tuples, gcs, mis, ivs, offs = [], [], [], [], [0]
    table: List[Optional[K.EventMap]] = [None]
    pdf = 0
    for pid in range(1, max(ids) + 1):
        n_states = 5 if pid in sil_ids else 3
        per_state = []
        for hs in range(n_states):
            st = stats.get((pid, hs))
            if st is not None and st[2] >= 5:
                mean = st[0] / st[2]
                var = np.maximum(st[1] / st[2] - mean * mean, 0.01)
            else:
                mean, var = gmean, gvar
            g = _gauss(mean, var)
            gcs.append([g[0]]); mis.append(g[1][None]); ivs.append(g[2][None]); offs.append(offs[-1] + 1)
            tuples.append((pid, hs, pdf, pdf))
            per_state.append(K.EventMap("CE", answer=pdf))
            pdf += 1
        table.append(K.EventMap("TE", key=-1, table=per_state))
    tuples = np.asarray(tuples, dtype=np.int32)
    raw = K.RawTransitionModel(topo, tuples, _log_probs(topo, tuples))
    am = M.DiagGmmModel(dim, np.concatenate(gcs).astype(np.float32), np.concatenate(mis), np.concatenate(ivs),
                        np.asarray(offs, dtype=np.int32))
    tree = K.ContextDependency(1, 0, K.EventMap("TE", key=0, table=table))
    return SynthModel(M.TransitionModel(raw), am, tree)


def train_monophone(world: SynthWorld, feature_fn: Callable[[np.ndarray, int], np.ndarray], n_train: int = 200,
                    first_index: int = 1_000_000) -> SynthModel:
    """BASELINE config 2: 3-state Bakis per phone + 5-state silence topology, 1 Gaussian per state, D = 39."""
    return monophone_from_stats(world.lexicon, _accumulate(world, feature_fn, n_train, first_index))


def seeded_lda(seed: int = SEED) -> np.ndarray:
    rng = np.random.default_rng(seed + 7)
    q, _ = np.linalg.qr(rng.normal(size=(91, 91)))
    return np.ascontiguousarray(q[:40]).astype(np.float32)


def seeded_fmllr(n_spk: int, seed: int = SEED) -> np.ndarray:
    rng = np.random.default_rng(seed + 11)
    a = np.eye(40)[None] + 0.05 * rng.normal(size=(n_spk, 40, 40))
    b = 0.1 * rng.normal(size=(n_spk, 40, 1))
    return np.concatenate([a, b], axis=2).astype(np.float32)


def train_triphone(world: SynthWorld, feature_fn: Callable[[np.ndarray, int], np.ndarray], n_train: int = 200,
                   first_index: int = 1_000_000, n_gauss: int = 32, n_classes: int = 5, seed: int = SEED) -> SynthModel:
    """BASELINE config 3: context-dependent model, pdf = table over (left class, centre, right class, state):
    66·3·25 + 2·5 ≈ 5k leaves, 32 Gaussians per leaf = leaf mean + N(0, 0.3σ) perturbations, Dirichlet(1) weights.
    ``feature_fn`` must return the 40-dim LDA(+fMLLR) features the model is to live in."""
    rng = np.random.default_rng(seed + 3)
    pt = world.lexicon.phone_table
    ids = [k for k, s in pt if s != "<eps>"]
    n_ids = max(ids)
    sil_ids = [pt.find("sil"), pt.find("spn")]
    topo = _topology(n_ids, sil_ids)
    stats = _accumulate(world, feature_fn, n_train, first_index)
    dim = next(iter(stats.values()))[0].shape[0]
    glob = [sum(s[0] for s in stats.values()), sum(s[1] for s in stats.values()), sum(s[2] for s in stats.values())]
    gmean = glob[0] / glob[2]
    gvar = np.maximum(glob[1] / glob[2] - gmean * gmean, 0.01)
    cls = np.concatenate([[0], rng.integers(0, n_classes, size=n_ids)])  # context class of each phone id; 0 = boundary
    cls[sil_ids] = 0
    tuples, gcs, mis, ivs, offs = [], [], [], [], [0]
    centre_table: List[Optional[K.EventMap]] = [None]
    pdf = 0

    def add_leaf(mean, var):
        nonlocal pdf
        # n_gauss = 0: a mixture size per leaf as a trained model has them (occupancy-driven: log-normal, median 11,
        # 1..48 — mostly small slots, a few leaves beyond one 32-row block)
        ng = n_gauss if n_gauss > 0 else int(np.clip(np.round(rng.lognormal(2.4, 0.6)), 1, 48))
        w = rng.dirichlet(np.ones(ng))
        sd = np.sqrt(var)
        for k in range(ng):
            g = _gauss(mean + 0.3 * sd * rng.normal(size=dim), var, w[k])
            gcs.append(g[0]); mis.append(g[1]); ivs.append(g[2])
        offs.append(offs[-1] + ng)
        pdf += 1
        return pdf - 1

    for pid in range(1, n_ids + 1):
        silence = pid in sil_ids
        n_states = 5 if silence else 3
        per_state = []
        for hs in range(n_states):
            st = stats.get((pid, hs))
            if st is not None and st[2] >= 5:
                mean = st[0] / st[2]
                var = np.maximum(st[1] / st[2] - mean * mean, 0.01)
            else:
                mean, var = gmean, gvar
            if silence:
                leaf = add_leaf(mean, var)
                tuples.append((pid, hs, leaf, leaf))
                per_state.append(K.EventMap("CE", answer=leaf))
            else:
                # table over left phone → table over right phone → leaf shared by context class pair (EventMap objects of
                # identical context classes are shared: the same tree, 25 leaves and 5 right-tables per state to pickle)
                leaves = {}
                leaf_maps = {}
                right_tabs = {}
                left_tab = []
                for lp in range(n_ids + 1):
                    if int(cls[lp]) not in right_tabs:
                        right_tab = []
                        for rp in range(n_ids + 1):
                            key = (int(cls[lp]), int(cls[rp]))
                            if key not in leaves:
                                shift = 0.15 * np.sqrt(var) * rng.normal(size=dim)
                                leaves[key] = add_leaf(mean + shift, var)
                                leaf_maps[key] = K.EventMap("CE", answer=leaves[key])
                                tuples.append((pid, hs, leaves[key], leaves[key]))
                            right_tab.append(leaf_maps[key])
                        right_tabs[int(cls[lp])] = K.EventMap("TE", key=2, table=right_tab)
                    left_tab.append(right_tabs[int(cls[lp])])
                per_state.append(K.EventMap("TE", key=0, table=left_tab))
        centre_table.append(K.EventMap("TE", key=-1, table=per_state))
    tuples = np.asarray(sorted(set(tuples)), dtype=np.int32)
    raw = K.RawTransitionModel(topo, tuples, _log_probs(topo, tuples))
    am = M.DiagGmmModel(dim, np.asarray(gcs, dtype=np.float32), np.stack(mis), np.stack(ivs), np.asarray(offs, dtype=np.int32))
    tree = K.ContextDependency(3, 1, K.EventMap("TE", key=1, table=centre_table))
    return SynthModel(M.TransitionModel(raw), am, tree)
