#!/usr/bin/env python3
"""Capture REFERENCE golden vectors through kalpy — to be run by hand on a machine that has kalpy/Kaldi.

The parity of this repository's oracle is UNPINNED: the reference's arithmetic lives in kalpy → Kaldi → OpenFst, which
cannot be installed in the build container, and the reference's own tests hold no frame-level expectations.  This
script is the missing link: wherever `import kalpy` works (a stock `conda install -c conda-forge montreal-forced-aligner`
environment, Kalpy 0.6.x), it pushes the SAME inputs tests/golden/make_golden.py uses through the SAME kalpy calls the
reference makes (MFA/online/alignment.py:77-122, MFA/corpus/features.py:235, MFA/alignment/multiprocessing.py:814-853,
:1415) and writes the results under the SAME keys as tests/golden/oracle_vectors.npz, into
tests/golden/kalpy_vectors.npz.  tests/test_golden_cpu.py then checks the oracle against it when the file exists
(`MFA_KALPY_GOLDEN=path` or the default location): from that moment the oracle is pinned by reference output.

Nothing in the pipeline depends on this script; it is never run in the build container or on the GPU box, and it
contains no reference code — only calls into the reference's public dependency.

    python tools/capture_kalpy_golden.py [--out tests/golden/kalpy_vectors.npz]

Inputs (all under tests/golden/ref_fixtures/, data files the reference's tests ship): mono_model.zip,
acoustic_g2p_output_model.zip (lda.mat), acoustic_corpus.wav, test_acoustic.txt.
The training graph is the one THIS repository builds for the text (written as an OpenFst binary and handed to kalpy), so
that alignments are compared on identical graphs; the graph kalpy's own TrainingGraphCompiler builds for the same text is
stored beside it (`kalpy_graph_*`) for the graph-builder comparison.
"""
from __future__ import annotations

import argparse
import io
import sys
import tempfile
import zipfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

SECONDS = 3.0
TEXT = "this is the acoustic corpus i'm talking"


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden" / "kalpy_vectors.npz"))
    args = ap.parse_args()
    try:
        from _kalpy.fstext import VectorFst
        from _kalpy.gmm import gmm_compute_likes
        from _kalpy.matrix import FloatMatrix
        from kalpy.feat.cmvn import CmvnComputer
        from kalpy.feat.mfcc import MfccComputer
        from kalpy.fstext.lexicon import LexiconCompiler as KalpyLexiconCompiler
        from kalpy.decoder.training_graphs import TrainingGraphCompiler as KalpyGraphCompiler
        from kalpy.gmm.align import GmmAligner
        from kalpy.gmm.utils import read_gmm_model
        from kalpy.utterance import Segment
        from kalpy.utterance import Utterance as KalpyUtterance
    except ImportError as e:
        print(f"kalpy is not importable here ({e}); run this where the reference's dependencies are installed", file=sys.stderr)
        return 2

    from montreal_forced_aligner_amd import graph as G
    from montreal_forced_aligner_amd import kaldi_io as K
    from montreal_forced_aligner_amd import model as M

    ref = ROOT / "tests" / "golden" / "ref_fixtures"
    tmp = Path(tempfile.mkdtemp(prefix="kalpy_golden_"))
    with zipfile.ZipFile(ref / "mono_model.zip") as z:
        z.extractall(tmp)
    with zipfile.ZipFile(ref / "acoustic_g2p_output_model.zip") as z:
        z.extractall(tmp / "g2p")
    mono_dir = next(p.parent for p in tmp.rglob("final.mdl") if "g2p" not in p.parts)
    lda_path = next((tmp / "g2p").rglob("lda.mat"))
    wav = ref / "acoustic_corpus.wav"
    out = {"pcm_samples": np.int64(int(16000 * SECONDS)), "text": np.array(TEXT)}

    # ---- MFCC (MfccFunction._run → compute_mfccs_for_export(seg, compress=False); options = MFA defaults, dither 0)
    seg = Segment(str(wav), 0.0, SECONDS, 0)
    for snip in (0, 1):
        mfcc_computer = MfccComputer(sample_frequency=16000, frame_length=25, frame_shift=10, dither=0.0, low_frequency=20,
                                     high_frequency=7800, num_mel_bins=23, num_coefficients=13, use_energy=False,
                                     energy_floor=0.0, raw_energy=True, cepstral_lifter=22, preemphasis_coefficient=0.97,
                                     snip_edges=bool(snip), remove_dc_offset=True, window_type="povey")
        m = mfcc_computer.compute_mfccs_for_export(seg, compress=False)
        out[f"mfcc_snip{snip}"] = np.array(m.numpy() if hasattr(m, "numpy") else m, dtype=np.float32)
    mfcc_computer = MfccComputer(sample_frequency=16000, frame_length=25, frame_shift=10, dither=0.0, low_frequency=20,
                                 high_frequency=7800, num_mel_bins=23, num_coefficients=13, use_energy=False,
                                 energy_floor=0.0, raw_energy=True, cepstral_lifter=22, preemphasis_coefficient=0.97,
                                 snip_edges=False, remove_dc_offset=True, window_type="povey")

    # ---- CMVN statistics and the two feature chains (align_utterance_online: :83-94)
    utt = KalpyUtterance(seg, TEXT)
    utt.generate_mfccs(mfcc_computer)
    cmvn = CmvnComputer().compute_cmvn_from_features([utt.mfccs])
    out["cmvn_stats"] = np.array(cmvn.numpy(), dtype=np.float64)
    utt.apply_cmvn(cmvn)
    feats = utt.generate_features(mfcc_computer, None)                      # Δ+ΔΔ path (mono_model: deltas true, no LDA)
    x = np.array(feats.numpy(), dtype=np.float32)
    out["delta_feats"] = x
    utt2 = KalpyUtterance(seg, TEXT)
    utt2.generate_mfccs(mfcc_computer)
    utt2.apply_cmvn(cmvn)
    lda_mat = FloatMatrix()
    from _kalpy.util import ReadKaldiObject
    ReadKaldiObject(str(lda_path), lda_mat)
    out["lda_feats"] = np.array(utt2.generate_features(mfcc_computer, None, lda_mat=lda_mat).numpy(), dtype=np.float32)

    # ---- all-pdf log-likelihoods of the first 50 frames (gmm_compute_likes, MFA/alignment/multiprocessing.py:1415)
    tm_k, am_k = read_gmm_model(str(mono_dir / "final.mdl"))
    likes = gmm_compute_likes(am_k, FloatMatrix.from_numpy(x[:50].copy()) if hasattr(FloatMatrix, "from_numpy") else feats)
    out["loglikes_first50_allpdfs"] = np.array(likes.numpy() if hasattr(likes, "numpy") else likes, dtype=np.float32)[:50]

    # ---- this repository's training graph for TEXT, through kalpy's aligner (beam 100 / retry 400 as the reference's tests)
    import yaml
    meta = yaml.safe_load((mono_dir / "meta.yaml").read_text())
    tm, am = M.load_model_bytes((mono_dir / "final.mdl").read_bytes())
    tree = K.read_tree((mono_dir / "tree").read_bytes())
    lex = G.LexiconCompiler(position_dependent_phones=True, phones=meta["phones"], silence_phone="sp", oov_phone="spn")
    lex.load_pronunciations(ref / "test_acoustic.txt")
    lex.build_phone_table(["sil", "sp", "spn"])
    fst_plain = G.TrainingGraphCompiler(tm, tree, lex).compile_fst(TEXT)     # before AddTransitionProbs: kalpy adds them
    fst = G.add_transition_probs(fst_plain, tm.scaled_log_probs(1.0, 0.1))
    out["graph_arc_offsets"], out["graph_final"] = fst.arc_offsets.astype(np.int64), fst.final.astype(np.float32)
    for k in ("ilabel", "olabel", "weight", "nextstate"):
        out[f"graph_{k}"] = np.ascontiguousarray(fst.arcs[k])
    out["graph_start"] = np.int32(fst.start)
    buf = io.BytesIO()
    K.write_fst(buf, fst_plain)
    fst_path = tmp / "graph.fst"
    fst_path.write_bytes(buf.getvalue())
    vfst = VectorFst.Read(str(fst_path))
    aligner = GmmAligner(str(mono_dir / "final.mdl"), beam=100, retry_beam=400, transition_scale=1.0, acoustic_scale=0.1,
                         self_loop_scale=0.1)
    alignment = aligner.align_utterance(vfst, feats)
    if alignment is None:
        print("kalpy could not align the fixture (beam 100/400)", file=sys.stderr)
        return 1
    out["ali"] = np.asarray(alignment.alignment, dtype=np.int32)
    out["words"] = np.asarray(alignment.words, dtype=np.int32)
    out["like"] = np.float32(alignment.likelihood)
    out["status"] = np.int32(0)
    ctm = alignment.generate_ctm(aligner.transition_model, None, 0.01) if False else None   # phone table differs: intervals from ali
    from montreal_forced_aligner_amd import ctm as C
    out["phone_intervals"] = np.asarray(C.split_to_phones(out["ali"], tm), dtype=np.int32)

    # ---- kalpy's own graph for the same text (graph-builder comparison)
    try:
        klex = KalpyLexiconCompiler(position_dependent_phones=True, silence_phone="sp", oov_phone="spn")
        klex.load_pronunciations(str(ref / "test_acoustic.txt"))
        kgc = KalpyGraphCompiler(str(mono_dir / "final.mdl"), str(mono_dir / "tree"), klex)
        kf = kgc.compile_fst(TEXT)
        kpath = tmp / "kalpy_graph.fst"
        kf.Write(str(kpath))
        kg = K.read_fst(K.BinaryReader(kpath.read_bytes()))
        out["kalpy_graph_arc_offsets"] = kg.arc_offsets.astype(np.int64)
        out["kalpy_graph_final"] = kg.final.astype(np.float32)
        for k in ("ilabel", "olabel", "weight", "nextstate"):
            out[f"kalpy_graph_{k}"] = np.ascontiguousarray(kg.arcs[k])
        out["kalpy_graph_start"] = np.int32(kg.start)
    except Exception as e:   # the lexicon/phone-table conventions of the installed kalpy may differ: the rest is still valid
        print(f"kalpy's own graph not captured: {e}", file=sys.stderr)

    np.savez_compressed(args.out, **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
    print(f"wrote {args.out}; re-run `pytest tests/test_golden_cpu.py` to pin the oracle against it")
    return 0


if __name__ == "__main__":
    sys.exit(main())
