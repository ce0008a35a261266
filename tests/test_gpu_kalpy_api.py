"""GPU test of the kalpy-shaped host API: the call sequence of align_utterance_online
(MFA/online/alignment.py:77-122) on the reference's own plumbing fixture, checked against the oracle."""
import wave

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers

pytestmark = pytest.mark.gpu


def test_align_utterance_online_sequence(fx, tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from montreal_forced_aligner_amd import kalpy_api as KA
    from montreal_forced_aligner_amd import kaldi_io as K

    ar = K.load_acoustic_model_archive(helpers.REF / "mono_model.zip")
    (tmp_path / "final.mdl").write_bytes(ar["final.mdl"])
    (tmp_path / "tree").write_bytes(ar["tree"])
    seg_pcm = fx.pcm[: 16000 * 4 + 3200]
    wav_path = tmp_path / "utt.wav"
    with wave.open(str(wav_path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(seg_pcm.tobytes())
    text = "this is the acoustic corpus i'm talking pretty fast here"

    lexicon_compiler = KA.LexiconCompiler(position_dependent_phones=True, phones=fx.mono_meta["phones"],
                                          silence_phone="sp", oov_phone="spn")
    lexicon_compiler.load_pronunciations(helpers.REF / "test_acoustic.txt")
    lexicon_compiler.build_phone_table(["sil", "sp", "spn"])
    mfcc_computer = KA.MfccComputer(sample_frequency=16000, frame_length=25, frame_shift=10, num_mel_bins=23,
                                    num_coefficients=13, snip_edges=False, dither=0.0, use_energy=False)
    utterance = KA.Utterance(KA.Segment(wav_path, 0.0, None, 0), text)
    graph_compiler = KA.TrainingGraphCompiler(tmp_path / "final.mdl", tmp_path / "tree", lexicon_compiler)
    utterance.generate_mfccs(mfcc_computer)
    cmvn = KA.CmvnComputer().compute_cmvn_from_features([utterance.mfccs])
    utterance.apply_cmvn(cmvn)
    feats = utterance.generate_features(mfcc_computer, None)
    assert feats.shape == (420, 39)
    fst = graph_compiler.compile_fst(text)
    aligner = KA.GmmAligner(tmp_path / "final.mdl", beam=100, retry_beam=400, transition_scale=1.0, acoustic_scale=0.1,
                            self_loop_scale=0.1)
    assert KA.GmmAligner(tmp_path / "final.mdl", beam=10, retry_beam=40).align_utterance(fst, feats[:30]) is None
    alignment = aligner.align_utterance(fst, feats)
    assert alignment is not None and len(alignment.alignment) == 420
    phone_intervals = alignment.generate_ctm(aligner.transition_model, lexicon_compiler.phone_table, mfcc_computer.frame_shift)
    ctm = lexicon_compiler.phones_to_pronunciations(alignment.words, phone_intervals, transcription=False, text=text)
    ctm.likelihood = alignment.likelihood
    ctm.update_utterance_boundaries(0.0, 4.2)
    words = [w.label for w in ctm.word_intervals if w.label != lexicon_compiler.silence_word]
    assert words == text.split()
    out = tmp_path / "utt.TextGrid"
    ctm.export_textgrid(out, file_duration=4.2, output_format="long_textgrid", silence_words=(lexicon_compiler.silence_word,))
    assert 'name = "words"' in out.read_text() and 'name = "phones"' in out.read_text()

    # oracle on the same audio/graph
    x = fx.mono_feats(seg_pcm)
    am, tm = fx.mono_am, fx.mono_tm
    pl = np.unique(tm.id2pdf[fst.arcs["ilabel"]])
    ref = helpers.oracle_align(tm, fx.mono_graph(text), O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars,
                                                                        am.pdf_offsets, pl), pl, beam=100.0, retry_beam=400.0)
    assert ref["status"] in (0, 1)
    assert np.array_equal(np.asarray(alignment.alignment), ref["ali"])
    assert abs(alignment.likelihood - ref["like"]) / 420 < 1e-3


def test_archives_follow_the_corpus_file_flow(fx, tmp_path):
    """feats.ark/scp (compressed) → cmvn.ark → FeatureArchive (CMVN + deltas) → export_alignments → AlignmentArchive:
    the table files the corpus path passes between MfccFunction, calc_cmvn, AlignFunction and the extraction step
    (SURVEY §8b "on-disk formats at the boundary")."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from montreal_forced_aligner_amd import kaldi_io as K
    from montreal_forced_aligner_amd import kalpy_api as KA

    ar = K.load_acoustic_model_archive(helpers.REF / "mono_model.zip")
    (tmp_path / "final.mdl").write_bytes(ar["final.mdl"])
    (tmp_path / "tree").write_bytes(ar["tree"])
    cuts = {"1-1": (0.0, 4.2, "this is the acoustic corpus i'm talking pretty fast here"),
            "1-2": (4.0, 6.5, "there's nothing going else going on")}
    mfcc_computer = KA.MfccComputer(sample_frequency=16000, frame_length=25, frame_shift=10, num_mel_bins=23,
                                    num_coefficients=13, snip_edges=False, dither=0.0, use_energy=False)
    segs = [(k, fx.pcm[int(a * 16000): int(b * 16000)]) for k, (a, b, _) in cuts.items()]
    mfcc_computer.export_feats(tmp_path / "feats.ark", segs, write_scp=True, compress=True)
    raw = dict(K.read_ark((tmp_path / "feats.ark").read_bytes(), "matrix"))
    direct = {k: mfcc_computer.compute_mfccs(x) for k, x in segs}
    for k in raw:   # 8-bit codec: coarse but close
        assert raw[k].shape == direct[k].shape and np.abs(raw[k] - direct[k]).max() < 0.02 * (direct[k].max() - direct[k].min())
    KA.CmvnComputer().export_cmvn(tmp_path / "cmvn.ark", raw, {"1": ["1-1", "1-2"]})
    cm = dict(K.read_ark((tmp_path / "cmvn.ark").read_bytes(), "matrix"))["1"]
    assert cm.shape == (2, 14) and cm.dtype == np.float64 and cm[0, 13] == sum(v.shape[0] for v in raw.values())
    fa = KA.FeatureArchive(tmp_path / "feats.scp", utt2spk={"1-1": "1", "1-2": "1"}, cmvn_file_name=tmp_path / "cmvn.scp",
                           deltas=True)
    feats = dict(fa)
    assert fa.use_deltas and not fa.use_splices
    for k, m in raw.items():   # same chain through the oracle on the decompressed matrices
        ref = O.deltas(O.cmvn_apply(cm, m))
        assert np.abs(feats[k] - ref).max() < 1e-3
    lex = KA.LexiconCompiler(position_dependent_phones=True, phones=fx.mono_meta["phones"], silence_phone="sp", oov_phone="spn")
    lex.load_pronunciations(helpers.REF / "test_acoustic.txt")
    lex.build_phone_table(["sil", "sp", "spn"])
    gc = KA.TrainingGraphCompiler(tmp_path / "final.mdl", tmp_path / "tree", lex)
    gc.export_graphs(tmp_path / "fsts.ark", [(k, t) for k, (_a, _b, t) in cuts.items()])
    graphs = list(K.read_ark((tmp_path / "fsts.ark").read_bytes(), "fst"))
    aligner = KA.GmmAligner(tmp_path / "final.mdl", beam=100, retry_beam=400)
    seen = []
    assert aligner.acoustic_model_path.endswith("final.mdl")    # AlignFunction tests the suffix (multiprocessing.py:842)
    aligner.export_alignments(tmp_path / "ali.ark", KA.FstArchive(tmp_path / "fsts.ark"), feats, word_file_name=tmp_path / "words.ark",
                              likelihood_file_name=tmp_path / "likes.ark", callback=seen.append)
    assert [k for k, _ in seen] == list(cuts) and all(like is not None for _, like in seen)
    aa = KA.AlignmentArchive(tmp_path / "ali.ark", words_file_name=tmp_path / "words.ark", likelihood_file_name=tmp_path / "likes.ark")
    al = aa["1-2"]
    assert len(al.alignment) == feats["1-2"].shape[0] and len(al.words) == len(cuts["1-2"][2].split())
    assert [a.utterance_id for a in aa] == list(cuts)
    with pytest.raises(KeyError):
        aa["1-3"]
    direct_al = aligner.align_utterance(graphs[1][1], feats["1-2"])
    assert direct_al.alignment == al.alignment and direct_al.words == al.words


def test_native_graph_batch_packs_like_a_list_of_graphs(engine, fx):
    """A batch from the native compiler (graph_native.FstBatch: views of concatenated arrays) and the same graphs as a plain
    list of per-utterance objects give the same device layout, tensor for tensor — and the graphs are graph.py's."""
    import torch

    from montreal_forced_aligner_amd import graph as G

    engine.load_gmm(fx.mono_am)
    texts = ["this is the acoustic corpus", "um and that should be all thanks", "", "there's nothing going else going on zzzoov"]
    scaled = fx.mono_tm.scaled_log_probs(1.0, 0.1)
    batch = fx.mono_gc.compile_fsts(texts, scaled)
    assert getattr(batch, "arcs", None) is not None
    ref = [G.add_transition_probs(fx.mono_gc.compile_fst(t), scaled) for t in texts]
    for a, b in zip(batch, ref):
        assert np.array_equal(a.arc_offsets, b.arc_offsets) and np.array_equal(a.arcs, b.arcs) and np.array_equal(a.final, b.final)
    keep = [0, 1, 3]                                   # (the empty transcript's graph has no arcs: not for the decoder)
    sub = fx.mono_gc.compile_fsts([texts[k] for k in keep], scaled)
    p1 = engine.pack_graphs(sub, fx.mono_tm)
    p2 = engine.pack_graphs([ref[k] for k in keep], fx.mono_tm)
    assert (p1.n_utt, p1.max_states, p1.max_arcs, p1.total_arcs, p1.groups) == (p2.n_utt, p2.max_states, p2.max_arcs, p2.total_arcs, p2.groups)
    for k in p1.tensors:
        assert torch.equal(p1.tensors[k], p2.tensors[k]), k
    for name in ("pdf_list", "pdf_off", "class_counts", "pdf_first_frame", "pdf_last_depth", "state_depth", "group_counts"):
        assert torch.equal(getattr(p1, name), getattr(p2, name)), name
    assert np.array_equal(p1.pdf_off_host, p2.pdf_off_host)
