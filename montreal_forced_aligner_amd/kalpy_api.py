"""Host-side mirror of the kalpy objects MFA's alignment path calls — same names, argument meaning and
error behaviour, backed by libmfa_hip.so instead of Kaldi.

With these in place ``align_utterance_online``-style code (MFA/online/alignment.py:29-123) runs unchanged:

    graph_compiler = TrainingGraphCompiler(model_path, tree_path, lexicon_compiler)
    utterance.generate_mfccs(mfcc_computer); cmvn = CmvnComputer().compute_cmvn_from_features([utterance.mfccs])
    utterance.apply_cmvn(cmvn); feats = utterance.generate_features(mfcc_computer, None, lda_mat=…, fmllr_trans=…)
    fst = graph_compiler.compile_fst(text)
    aligner = GmmAligner(model_path, beam=10, retry_beam=40, transition_scale=1.0, acoustic_scale=0.1, self_loop_scale=0.1)
    alignment = aligner.align_utterance(fst, feats)          # None when the utterance cannot be aligned
    ctm = lexicon_compiler.phones_to_pronunciations(alignment.words, alignment.generate_ctm(tm, phone_table, shift), …)

Matrices cross this API as numpy arrays (kalpy: FloatMatrix/DoubleMatrix).  Every numeric step runs on the GPU through
the C ABI; a missing library or GPU raises ``MfaHipError`` — there is no CPU path.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from . import ctm as _ctm
from . import graph as _graph
from . import kaldi_io
from . import model as _model
from .graph import LexiconCompiler as _LexiconCompiler
from .graph import Pronunciation as KalpyPronunciation  # noqa: F401  (name used by the reference)

logger = logging.getLogger("kalpy.align")      # the reference's per-job logger name (MFA/utils.py:1435-1476)

_ENGINE = None


def get_engine(device: int = 0):
    """One engine per process (the reference: one aligner object per worker, never shared across processes)."""
    global _ENGINE
    if _ENGINE is None:
        from .engine import AlignmentEngine

        _ENGINE = AlignmentEngine(device)
    return _ENGINE


# ------------------------------------------------------------------------------------------------ audio / features
class Segment:
    """``Segment(path, begin, end, channel)`` — PCM16 16 kHz wav only (resampling is outside the parity domain)."""

    def __init__(self, file_path, begin: Optional[float] = None, end: Optional[float] = None, channel: int = 0):
        self.file_path = str(file_path)
        self.begin, self.end, self.channel = begin, end, channel or 0

    def load_audio(self) -> np.ndarray:
        pcm, sr = kaldi_io.read_wav_pcm16(self.file_path)
        if sr != 16000:
            raise kaldi_io.KaldiFormatError(f"{self.file_path}: {sr} Hz audio; only native 16 kHz PCM16 is supported")
        x = pcm[min(self.channel, pcm.shape[0] - 1)]
        b = 0 if self.begin is None else int(round(self.begin * sr))
        e = x.shape[0] if self.end is None else int(round(self.end * sr))
        return np.ascontiguousarray(x[b:e])


class MfccComputer:
    """``MfccComputer(**mfcc_options)`` with MFA's option names (MFA/corpus/features.py:780-820)."""

    _MAP = dict(sample_frequency="sample_frequency", frame_length="frame_length_ms", frame_shift="frame_shift_ms",
                preemphasis_coefficient="preemphasis", low_frequency="low_frequency", high_frequency="high_frequency",
                cepstral_lifter="cepstral_lifter", energy_floor="energy_floor", num_mel_bins="num_mel_bins",
                num_coefficients="num_coefficients", snip_edges="snip_edges", remove_dc_offset="remove_dc_offset",
                use_energy="use_energy", raw_energy="raw_energy")

    def __init__(self, **options):
        self.parameters = dict(options)
        dither = options.get("dither", 0.0)
        if dither not in (0, 0.0, None):
            raise ValueError("dither must be 0: dithered features are random and outside the parity domain")
        self._opts = {self._MAP[k]: v for k, v in options.items() if k in self._MAP and v is not None}
        self.frame_shift = float(options.get("frame_shift", 10)) / 1000.0

    def _configure(self):
        eng = get_engine()
        eng.configure_mfcc(**{k: (int(v) if isinstance(v, bool) else v) for k, v in self._opts.items()})
        return eng

    def compute_mfccs(self, segment: Union[Segment, np.ndarray]) -> np.ndarray:
        import torch

        eng = self._configure()
        pcm = segment.load_audio() if isinstance(segment, Segment) else np.asarray(segment, dtype=np.int16)
        out, _ = eng.mfcc(torch.from_numpy(pcm.copy()).to(eng.device), np.array([0, pcm.shape[0]], dtype=np.int64))
        return out.cpu().numpy()

    def compute_mfccs_for_export(self, segment, compress: bool = True):
        """``compress=True``: the matrix wrapped so that ``kaldi_io.write_ark_entry`` stores it as a Kaldi CompressedMatrix
        (what MfccFunction writes, MFA/corpus/features.py:235); ``compress=False``: the float32 matrix (FineTuneFunction)."""
        m = self.compute_mfccs(segment)
        return CompressedFeatures(m) if compress else m

    def export_feats(self, file_name, segments: Iterable[Tuple[str, Union["Segment", np.ndarray]]], write_scp: bool = True,
                     compress: bool = True, batch_size: int = 256) -> None:
        """``feats.*.ark`` (+ ``.scp``) for (key, segment) pairs, MFCCs computed in device batches (MfccFunction._run)."""
        import torch

        eng = self._configure()
        ark = Path(file_name)
        scp = ark.with_suffix(".scp") if write_scp else None
        lines = []
        with open(ark, "wb") as f:
            batch: List[Tuple[str, np.ndarray]] = []

            def flush():
                if not batch:
                    return
                so = np.concatenate([[0], np.cumsum([len(x) for _, x in batch])]).astype(np.int64)
                out, fo = eng.mfcc(torch.from_numpy(np.concatenate([x for _, x in batch])).to(eng.device), so)
                out = out.cpu().numpy()
                for i, (key, _) in enumerate(batch):
                    off = kaldi_io.write_ark_entry(f, key, out[fo[i]: fo[i + 1]], "compressed_matrix" if compress else "matrix")
                    lines.append(f"{key} {ark}:{off}\n")
                batch.clear()

            for key, seg in segments:
                pcm = seg.load_audio() if isinstance(seg, Segment) else np.asarray(seg, dtype=np.int16)
                batch.append((key, np.ascontiguousarray(pcm, dtype=np.int16)))
                if len(batch) >= batch_size:
                    flush()
            flush()
        if scp is not None:
            scp.write_text("".join(lines), encoding="utf8")


class CompressedFeatures(np.ndarray):
    """A float32 matrix tagged for 8-bit storage: ``FeatureArchive``/``write_features`` write it as Kaldi "CM"."""

    def __new__(cls, m):
        return np.asarray(m, dtype=np.float32).view(cls)


class CmvnComputer:
    def compute_cmvn_from_features(self, feats: Sequence[np.ndarray]) -> np.ndarray:
        """Kaldi layout float64 [2, dim+1]: sums + count / sums of squares."""
        import torch

        eng = get_engine()
        frame_off = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int64)
        d = torch.from_numpy(np.concatenate(feats).astype(np.float32)).to(eng.device)
        return eng.cmvn_stats(d, frame_off, np.zeros(len(feats), dtype=np.int32), 1).cpu().numpy()[0]


    def export_cmvn(self, file_name, feature_archive, spk2utt: Dict[str, Sequence[str]], write_scp: bool = True) -> None:
        """Per-speaker statistics → ``cmvn.ark`` (+ ``.scp``) of float64 [2, dim+1] matrices keyed by speaker
        (calc_cmvn, MFA/corpus/acoustic_corpus.py:1315-1367).  ``feature_archive``: (utt key, matrix) pairs or a dict."""
        feats = dict(feature_archive) if not isinstance(feature_archive, dict) else feature_archive
        entries = []
        for spk in sorted(spk2utt):
            mats = [np.asarray(feats[u]) for u in spk2utt[spk] if u in feats]
            if mats:
                entries.append((str(spk), self.compute_cmvn_from_features(mats)))
        ark = Path(file_name)
        kaldi_io.write_table(ark, entries, "matrix", ark.with_suffix(".scp") if write_scp else None)


class FeatureArchive:
    """``FeatureArchive(scp, utt2spk=, cmvn_file_name=, lda_mat_file_name=, transform_file_name=, deltas=, …)``: iterable of
    (utt_id, final feature matrix) — Job.construct_feature_archive's chain (MFA/db.py:2101-2136): base features from the
    scp (compressed or not) → per-speaker CMVN → Δ+ΔΔ or splice(±3)+LDA → per-speaker fMLLR.  Utterances are pushed
    through the device in batches of ``batch_size``; ``utt2spk`` is a dict or a Kaldi utt2spk file."""

    def __init__(self, file_name, utt2spk=None, cmvn_file_name=None, lda_mat_file_name=None, transform_file_name=None,
                 vad_file_name=None, deltas: bool = False, splices: bool = False, splice_frames: int = 3, subsample_n: int = 0,
                 use_sliding_cmvn: bool = False, batch_size: int = 256):
        if vad_file_name is not None or subsample_n or use_sliding_cmvn:
            raise NotImplementedError("vad / subsampling / sliding CMVN are not part of the alignment path")
        self.file_name = Path(file_name)
        self.utt2spk = self._read_map(utt2spk)
        self.cmvn_read_specifier = cmvn_file_name
        self.lda_mat_file_name = lda_mat_file_name
        self.transform_read_specifier = transform_file_name
        self.use_splices = bool(splices or lda_mat_file_name)
        self.use_deltas = bool(deltas) and not self.use_splices
        self.splice_frames = splice_frames
        self.batch_size = batch_size
        self._entries = kaldi_io.read_scp(self.file_name) if self.file_name.suffix == ".scp" else None
        self._cache: Dict[str, bytes] = {}
        self._cmvn = self._read_table(cmvn_file_name)
        self._trans = self._read_table(transform_file_name)
        self._lda = None if lda_mat_file_name is None else kaldi_io.read_matrix_file(Path(lda_mat_file_name).read_bytes())

    @staticmethod
    def _read_map(m):
        if m is None or isinstance(m, dict):
            return m
        return dict(line.split()[:2] for line in Path(m).read_text(encoding="utf8").splitlines() if line.strip())

    def _read_table(self, name) -> Optional[Dict[str, np.ndarray]]:
        if name is None:
            return None
        name = Path(name)
        if name.suffix == ".scp":
            return {k: kaldi_io.read_scp_object(self._cache, p, o, "matrix") for k, p, o in kaldi_io.read_scp(name)}
        return dict(kaldi_io.read_ark(name.read_bytes(), "matrix"))

    def _base(self) -> Iterator[Tuple[str, np.ndarray]]:
        if self._entries is not None:
            for key, path, off in self._entries:
                yield key, kaldi_io.read_scp_object(self._cache, path, off, "matrix")
        else:
            yield from kaldi_io.read_ark(self.file_name.read_bytes(), "matrix")

    def __iter__(self) -> Iterator[Tuple[str, np.ndarray]]:
        import torch

        eng = get_engine()
        batch: List[Tuple[str, np.ndarray]] = []

        def flush():
            spk_names = [self.utt2spk.get(k, k) if self.utt2spk else k for k, _ in batch]
            ids = {s: i for i, s in enumerate(dict.fromkeys(spk_names))}
            u2s = np.array([ids[s] for s in spk_names], dtype=np.int32)
            dim = batch[0][1].shape[1]
            cm = None
            if self._cmvn is not None:
                st = np.zeros((len(ids), 2, dim + 1), dtype=np.float64)
                st[:, 0, dim] = 1.0   # a speaker without statistics keeps its features (mean 0 over count 1)
                for s, i in ids.items():
                    if s in self._cmvn:
                        st[i] = self._cmvn[s]
                cm = torch.from_numpy(st).to(eng.device)
            lda = None if self._lda is None else torch.from_numpy(self._lda.astype(np.float32)).to(eng.device)
            fm = None
            if self._trans is not None and lda is not None:
                R = lda.shape[0]
                ft = np.tile(np.eye(R, R + 1, dtype=np.float32), (len(ids), 1, 1))
                for s, i in ids.items():
                    if s in self._trans:
                        ft[i] = self._trans[s]
                fm = torch.from_numpy(ft).to(eng.device)
            fo = np.concatenate([[0], np.cumsum([m.shape[0] for _, m in batch])]).astype(np.int64)
            d = torch.from_numpy(np.concatenate([m for _, m in batch]).astype(np.float32)).to(eng.device)
            if self.use_splices and lda is None:
                raise NotImplementedError("spliced features without an LDA matrix")
            if not self.use_deltas and lda is None:
                out = d if cm is None else None
                if out is None:   # CMVN only (FinalFeatureFunction's output): Δ kernel's first block is the CMVN-applied base
                    out = eng.features(d, fo, u2s, cm)[:, :dim]
                out = out.cpu().numpy()
            else:
                out = eng.features(d, fo, u2s, cm, lda=lda, fmllr=fm, splice_context=self.splice_frames).cpu().numpy()
            res = [(k, out[fo[i]: fo[i + 1]].copy()) for i, (k, _) in enumerate(batch)]
            batch.clear()
            return res

        for key, m in self._base():
            batch.append((key, m))
            if len(batch) >= self.batch_size:
                yield from flush()
        if batch:
            yield from flush()

    def close(self) -> None:
        self._cache.clear()


class AlignmentArchive:
    """``AlignmentArchive(ali, words_file_name=, likelihood_file_name=)``: iterable of Alignment and ``archive[utt_id]``
    (KeyError when absent) over the arks ``GmmAligner.export_alignments`` writes (MFA/alignment/multiprocessing.py:1856)."""

    def __init__(self, file_name, words_file_name=None, likelihood_file_name=None):
        self._ali = dict(kaldi_io.read_ark(Path(file_name).read_bytes(), "int_vector"))
        self._words = dict(kaldi_io.read_ark(Path(words_file_name).read_bytes(), "int_vector")) if words_file_name else {}
        self._likes = dict(kaldi_io.read_ark(Path(likelihood_file_name).read_bytes(), "vector")) if likelihood_file_name else {}

    def _make(self, key) -> "Alignment":
        pf = self._likes.get(key)
        like = float(np.sum(pf)) if pf is not None else float("nan")
        return Alignment(key, self._ali[key].tolist(), self._words.get(key, np.zeros(0, np.int32)).tolist(), like, pf)

    def __getitem__(self, key) -> "Alignment":
        if key not in self._ali:
            raise KeyError(key)
        return self._make(key)

    def __iter__(self):
        for key in self._ali:
            yield self._make(key)

    def close(self) -> None:
        pass


class Utterance:
    def __init__(self, segment: Segment, transcript: str, cmvn: Optional[np.ndarray] = None, fmllr: Optional[np.ndarray] = None):
        self.segment, self.transcript = segment, transcript
        self.cmvn, self.fmllr = cmvn, fmllr
        self.mfccs: Optional[np.ndarray] = None
        self._cmvn_applied: Optional[np.ndarray] = None

    def generate_mfccs(self, mfcc_computer: MfccComputer) -> None:
        self.mfccs = mfcc_computer.compute_mfccs(self.segment)

    def apply_cmvn(self, cmvn: np.ndarray) -> None:
        self.cmvn = np.asarray(cmvn, dtype=np.float64)

    def generate_features(self, mfcc_computer: MfccComputer, pitch_computer=None, lda_mat: Optional[np.ndarray] = None,
                          fmllr_trans: Optional[np.ndarray] = None, splice_context: int = 3) -> np.ndarray:
        """CMVN → Δ+ΔΔ, or splice+LDA (→ fMLLR) when ``lda_mat`` is given (MFA/alignment/multiprocessing.py:1287-1304)."""
        import torch

        if pitch_computer is not None:
            raise NotImplementedError("pitch features are not part of this engine")
        if self.mfccs is None:
            self.generate_mfccs(mfcc_computer)
        eng = get_engine()
        dev = eng.device
        frame_off = np.array([0, self.mfccs.shape[0]], dtype=np.int64)
        d = torch.from_numpy(self.mfccs.astype(np.float32)).to(dev)
        cm = None if self.cmvn is None else torch.from_numpy(self.cmvn[None].copy()).to(dev)
        u2s = np.zeros(1, dtype=np.int32)
        lda = None if lda_mat is None else torch.from_numpy(np.asarray(lda_mat, dtype=np.float32)).to(dev)
        fm = fmllr_trans if fmllr_trans is not None else self.fmllr
        fm = None if fm is None else torch.from_numpy(np.asarray(fm, dtype=np.float32)[None].copy()).to(dev)
        if fm is not None and lda is None:
            raise NotImplementedError("fMLLR without LDA is not supported by the feature kernel")
        return eng.features(d, frame_off, u2s, cm, lda=lda, fmllr=fm, splice_context=splice_context).cpu().numpy()


# ------------------------------------------------------------------------------------------------ lexicon / graphs
class LexiconCompiler(_LexiconCompiler):
    def phones_to_pronunciations(self, words, intervals, transcription: bool = False, text: Optional[str] = None):
        return _ctm.phones_to_pronunciations(self, words, intervals, transcription, text)


def _read_model(path_or_bytes) -> Tuple[_model.TransitionModel, _model.DiagGmmModel]:
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else Path(path_or_bytes).read_bytes()
    return _model.load_model_bytes(bytes(data))


def read_transition_model(path):
    return _read_model(path)[0]


def read_gmm_model(path):
    return _read_model(path)


class TrainingGraphCompiler(_graph.TrainingGraphCompiler):
    """``TrainingGraphCompiler(model_path, tree_path, lexicon_compiler, use_g2p=False, batch_size=…)``."""

    def __init__(self, model_path, tree_path, lexicon_compiler, use_g2p: bool = False, batch_size: int = 500):
        tm = model_path if isinstance(model_path, _model.TransitionModel) else _read_model(model_path)[0]
        tree = tree_path if isinstance(tree_path, kaldi_io.ContextDependency) else kaldi_io.read_tree(Path(tree_path).read_bytes())
        super().__init__(tm, tree, lexicon_compiler, use_g2p, batch_size)

    def export_graphs(self, file_name, records: Iterable[Tuple[str, str]], write_scp: bool = False, callback=None,
                      interjection_words=None) -> None:
        """Writes ``fsts.*.ark`` (key + OpenFst VectorFst<StdArc> binary per utterance).  ``records``:
        ``(key, text)`` pairs, or the utterance dicts CompileTrainGraphsFunction passes (``kaldi_id`` plus
        ``normalized_text`` / ``normalized_character_text`` with ``use_g2p``; MFA/alignment/multiprocessing.py:547-571).
        ``interjection_words`` belong to the transcript-verification workflow (``:512-536``; empty for alignment) and are
        refused when non-empty rather than ignored."""
        if interjection_words:
            raise NotImplementedError("interjection words (transcript verification) are not part of the alignment path")
        text_column = "normalized_character_text" if self.use_g2p else "normalized_text"
        with open(file_name, "wb") as f:
            chunk: List[Tuple[str, str]] = []

            def flush():
                for (key, _t), fst in zip(chunk, self.compile_fsts([t for _k, t in chunk])):   # native, batch_size at a time
                    kaldi_io.write_ark_entry(f, key, fst, "fst")
                    if callback:
                        callback(key)
                chunk.clear()

            for rec in records:
                chunk.append((rec["kaldi_id"], rec[text_column]) if isinstance(rec, dict) else tuple(rec))
                if len(chunk) >= max(1, int(self.batch_size)):
                    flush()
            flush()


class FstArchive:
    """``FstArchive(path)`` — the training-graph table ``fsts.*.ark`` AlignFunction opens (MFA/alignment/multiprocessing.py:
    828): iterable of ``(key, Fst)`` in file order, ``archive[key]`` for random access (KeyError when absent)."""

    def __init__(self, file_name):
        self.file_name = str(file_name)
        self._index: Optional[Dict[str, kaldi_io.Fst]] = None

    def __iter__(self):
        return kaldi_io.read_ark(Path(self.file_name).read_bytes(), "fst")

    def __getitem__(self, key: str) -> kaldi_io.Fst:
        if self._index is None:
            self._index = dict(iter(self))
        return self._index[key]

    def close(self) -> None:
        self._index = None


# ------------------------------------------------------------------------------------------------ alignment
@dataclass
class Alignment:
    utterance_id: Optional[str]
    alignment: List[int]
    words: List[int]
    likelihood: float
    per_frame_likelihoods: Optional[np.ndarray] = None
    status: int = 0                 # 0 aligned with the first beam, 1 with the retry beam (include/mfa_hip.h)

    def generate_ctm(self, transition_model, phone_table, frame_shift: float = 0.01):
        return _ctm.generate_ctm(self.alignment, transition_model, phone_table, frame_shift)


class GmmAligner:
    """``GmmAligner(model_path, beam=, retry_beam=, transition_scale=, acoustic_scale=, self_loop_scale=,
    disambiguation_symbols=)`` (MFA/alignment/multiprocessing.py:814; MFA/online/alignment.py:97-104)."""

    def __init__(self, acoustic_model_path, beam: float = 10, retry_beam: float = 40, transition_scale: float = 1.0,
                 acoustic_scale: float = 0.1, self_loop_scale: float = 0.1, disambiguation_symbols=None,
                 careful: bool = False):
        if careful:
            raise NotImplementedError("careful alignment is not implemented")
        # (AlignFunction._run tests `.endswith(".alimdl")` on it, MFA/alignment/multiprocessing.py:842)
        self.acoustic_model_path = acoustic_model_path if isinstance(acoustic_model_path, (bytes, bytearray)) else str(acoustic_model_path)
        self.transition_model, self.acoustic_model = _read_model(acoustic_model_path)
        self.beam, self.retry_beam = float(beam), float(retry_beam)
        if self.retry_beam != 0 and self.retry_beam <= self.beam:
            self.retry_beam = 4 * self.beam  # MFA/alignment/mixins.py:91-92
        self.transition_scale, self.acoustic_scale, self.self_loop_scale = transition_scale, acoustic_scale, self_loop_scale
        self.disambiguation_symbols = sorted(disambiguation_symbols or [])
        self._scaled = self.transition_model.scaled_log_probs(transition_scale, self_loop_scale)
        self._loaded = False

    def boost_silence(self, silence_weight: float, silence_phones: Sequence[int]) -> None:
        self.acoustic_model.boost_silence(silence_weight, _model.pdfs_of_phones(self.transition_model, silence_phones))
        self._loaded = False

    def _engine(self):
        eng = get_engine()
        if not self._loaded or eng.gmm is not self.acoustic_model:
            eng.load_gmm(self.acoustic_model)
            self._loaded = True
        return eng

    def align_utterances(self, fsts: Sequence[kaldi_io.Fst], feats: Sequence[np.ndarray], utterance_ids=None) -> List[Optional[Alignment]]:
        """Batched form of ``align_utterance``: one device launch per stage for the whole list."""
        import torch

        eng = self._engine()
        general = [k for k, f in enumerate(fsts) if eng.needs_general_decoder(f)]
        if general and len(general) < len(fsts):       # mixed batch: the two decoders take their own utterances
            fast = [k for k in range(len(fsts)) if k not in set(general)]
            out: List[Optional[Alignment]] = [None] * len(fsts)
            for part in (fast, general):
                ids = [utterance_ids[k] for k in part] if utterance_ids else None
                for k, r in zip(part, self.align_utterances([fsts[k] for k in part], [feats[k] for k in part], ids)):
                    out[k] = r
            return out
        frame_off = np.concatenate([[0], np.cumsum([x.shape[0] for x in feats])]).astype(np.int64)
        d_feats = torch.from_numpy(np.concatenate(feats).astype(np.float32)).to(eng.device)
        if general:
            # graphs with epsilon input arcs (kalpy-compiled fsts.*.ark) or very wide states: FasterDecoder as Kaldi runs
            # it, ProcessNonemitting included (mfa_align_general_batch)
            graphs = eng.pack_graphs_general([_graph.add_transition_probs(f, self._scaled) for f in fsts], self.transition_model)
            res = eng.align_general(graphs, d_feats, frame_off, beam=self.beam, retry_beam=self.retry_beam,
                                    acoustic_scale=self.acoustic_scale, want_frame_likes=True)
            res = {k: res[k].cpu().numpy() for k in self._RESULT_KEYS}
            return self._collect(res, frame_off, len(fsts), utterance_ids)
        scaled_fsts = [_graph.add_transition_probs(f, self._scaled) for f in fsts]
        graphs = eng.pack_graphs(scaled_fsts, self.transition_model)
        # features in, alignments out — the decodable is evaluated lazily, as Kaldi's is: per window of frames only the
        # pdfs that arcs near the live tokens can emit are scored (mfa_align_features_batch)
        res = eng.align_features(graphs, d_feats, frame_off, beam=self.beam, retry_beam=self.retry_beam,
                                 acoustic_scale=self.acoustic_scale, want_frame_likes=True)
        res = {k: res[k].cpu().numpy() for k in self._RESULT_KEYS}      # (not the score scratch: ΣT·P floats)
        # Token / back-pointer capacity overflows (status 3 / 4) are not alignment failures — FasterDecoder has no such
        # limit: those utterances are decoded again with the hard bounds (one token per graph state), which cannot
        # overflow, exactly as CorpusAligner._pass does.
        over = [u for u in range(len(fsts)) if int(res["status"][u]) in (3, 4)]
        if over:
            def take(us):
                fo_s = np.concatenate([[0], np.cumsum([frame_off[u + 1] - frame_off[u] for u in us])]).astype(np.int64)
                return fo_s, eng.gather_rows(d_feats, np.concatenate([np.arange(frame_off[u], frame_off[u + 1]) for u in us]))

            def merge(us, fo_s, r):
                r = {k: r[k].cpu().numpy() for k in self._RESULT_KEYS}
                for j, u in enumerate(us):
                    a, b, a2, b2 = int(frame_off[u]), int(frame_off[u + 1]), int(fo_s[j]), int(fo_s[j + 1])
                    for k in ("ali", "words", "frame_like"):
                        res[k][a:b] = r[k][a2:b2]
                    for k in ("n_words", "like", "status"):
                        res[k][u] = r[k][j]

            sub = eng.pack_graphs([scaled_fsts[u] for u in over], self.transition_model)
            fo2, f2 = take(over)
            mt, bp = sub.hard_bounds()
            merge(over, fo2, eng.align_features(sub, f2, fo2, beam=self.beam, retry_beam=self.retry_beam,
                                                acoustic_scale=self.acoustic_scale, max_tokens=mt, bp_tokens_per_frame=bp,
                                                want_frame_likes=True))
            # what still reports a capacity status (the epsilon closure's pop budget on a pathological epsilon sub-graph)
            # goes to the general decoder: Kaldi's loops as they are, no budget
            still = [u for u in over if int(res["status"][u]) in (3, 4)]
            if still:
                gg = eng.pack_graphs_general([scaled_fsts[u] for u in still], self.transition_model)
                fo3, f3 = take(still)
                merge(still, fo3, eng.align_general(gg, f3, fo3, beam=self.beam, retry_beam=self.retry_beam,
                                                    acoustic_scale=self.acoustic_scale, bp_tokens_per_frame=2 * gg.max_states + 64,
                                                    want_frame_likes=True))
        return self._collect(res, frame_off, len(fsts), utterance_ids)

    _RESULT_KEYS = ("ali", "words", "n_words", "like", "status", "frame_like")

    @staticmethod
    def _collect(res, frame_off, n, utterance_ids) -> List[Optional[Alignment]]:
        out: List[Optional[Alignment]] = []
        for u in range(n):
            st = int(res["status"][u])
            if st not in (0, 1):
                # the reference returns None and lets the caller count the failure (statuses beyond "no final token"
                # concern this utterance only, never the rest of the batch; they are logged with their code)
                if st != 2:
                    logger.warning("utterance %s: %s", utterance_ids[u] if utterance_ids else u, _lib.status_reason(st))
                out.append(None)
                continue
            a, b = int(frame_off[u]), int(frame_off[u + 1])
            nw = int(res["n_words"][u])
            out.append(Alignment(utterance_ids[u] if utterance_ids else None, res["ali"][a:b].tolist(),
                                 res["words"][a: a + nw].tolist(), float(res["like"][u]), res["frame_like"][a:b].copy(), st))
        return out

    def align_utterance(self, training_graph: kaldi_io.Fst, features: np.ndarray, utterance_id: Optional[str] = None):
        return self.align_utterances([training_graph], [features], [utterance_id])[0]

    def export_alignments(self, file_name, training_graph_archive, feature_archive, word_file_name=None,
                          likelihood_file_name=None, callback=None, batch_size: int = 256) -> None:
        """``training_graph_archive`` / ``feature_archive``: iterables of (key, Fst) / (key, matrix) with equal key order
        (kalpy: FstArchive, FeatureArchive).  Writes Kaldi int32-vector / float-vector arks keyed by utterance."""
        fa = open(file_name, "wb")
        fw = open(word_file_name, "wb") if word_file_name else None
        fl = open(likelihood_file_name, "wb") if likelihood_file_name else None
        try:
            feats = dict(feature_archive) if not isinstance(feature_archive, dict) else feature_archive
            batch: List[Tuple[str, kaldi_io.Fst]] = []

            def flush():
                if not batch:
                    return
                keys = [k for k, _ in batch]
                res = self.align_utterances([f for _, f in batch], [feats[k] for k in keys], keys)
                for key, al in zip(keys, res):
                    if al is not None:
                        kaldi_io.write_ark_entry(fa, key, np.asarray(al.alignment, dtype=np.int32), "int_vector")
                        if fw:
                            kaldi_io.write_ark_entry(fw, key, np.asarray(al.words, dtype=np.int32), "int_vector")
                        if fl:
                            kaldi_io.write_ark_entry(fl, key, al.per_frame_likelihoods.astype(np.float32), "vector")
                    if callback:
                        callback((key, al.likelihood if al is not None else None))
                batch.clear()

            for key, fst in training_graph_archive:
                if key not in feats:
                    continue
                batch.append((key, fst))
                if len(batch) >= batch_size:
                    flush()
            flush()
        finally:
            for f in (fa, fw, fl):
                if f:
                    f.close()


HierarchicalCtm = _ctm.HierarchicalCtm
CtmInterval = _ctm.CtmInterval
WordCtmInterval = _ctm.WordCtmInterval
