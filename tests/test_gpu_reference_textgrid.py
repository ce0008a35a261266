"""The DEVICE path against boundaries the reference itself holds.

(1) The reference's whole fixture — ``acoustic_corpus.wav`` + ``.lab`` + ``test_acoustic.txt`` + ``mono_model.zip`` — through
    ``CorpusAligner`` with the reference's test beams (beam 100 / retry 400, /root/reference/tests/conftest.py:1035-1037):
    2 672 frames in one utterance, frame-identical to the oracle, word tier compared with the gold
    ``acoustic_corpus.TextGrid``.  The bundled model is an untrained plumbing model — it spreads the words almost uniformly
    over the file — so the only bound the data supports is the ORDER of the words and their confinement to the file; the
    measured distance to the gold boundaries (median ≈ 3.6 s) is asserted only as "this model cannot pin boundaries".
(2) A model the gold TextGrids themselves supervise (tests/ref_supervised.py): closed set and held-out files through the
    device path (device MFCC → CMVN → Δ → lazy scoring → windowed decoder → intervals), same bounds as the oracle's CPU test
    (tests/test_reference_textgrid_cpu.py), and frame-identical to the oracle on the device's features."""
import difflib

import numpy as np
import pytest
import torch

from montreal_forced_aligner_amd import ctm as C
from montreal_forced_aligner_amd import graph as G
from montreal_forced_aligner_amd.aligner import AlignOptions, CorpusAligner, CorpusUtterance
from oracle import oracle as O
from tests import helpers
from tests import ref_supervised as RS
from tests.test_reference_textgrid_cpu import check_bounds

pytestmark = pytest.mark.gpu


def test_reference_fixture_whole_file_against_its_gold_textgrid(engine, fx, tmp_path):
    sr = 16000
    utt = CorpusUtterance("michael-0", "michael", fx.pcm, fx.text, file_name="acoustic_corpus", file_duration=len(fx.pcm) / sr)
    al = CorpusAligner(fx.mono_tm, fx.mono_am, fx.mono_tree, fx.mono_lex, options=AlignOptions(beam=100.0, retry_beam=400.0),
                       engine=engine)
    (res,) = al.align([utt])
    assert res is not None and res.num_frames == 2672
    # the oracle on its own features: frame-identical alignment, per-frame likelihood within 1e-3
    f = fx.mono_graph(fx.text)
    ref = helpers.oracle_align_feats(fx.mono_tm, f, fx.mono_feats(fx.pcm), fx.mono_am, beam=100.0, retry_beam=400.0)
    assert ref["status"] == 0 and np.array_equal(res.alignment, ref["ali"]) and np.array_equal(res.words, ref["words"])
    assert abs(res.per_frame_likelihood - ref["like"] / 2672) < 1e-3
    # word tier vs the gold file
    gold = C.read_short_textgrid(helpers.REF / "acoustic_corpus.TextGrid")
    gold_words = [e for e in gold["words"] if e[2] != ""]
    mine = [(w.begin, w.end, w.label) for w in res.ctm.word_intervals if w.label != fx.mono_lex.silence_word]
    assert [m[2] for m in mine] == fx.text.split()                       # <unk> intervals carry their transcript spelling
    sm = difflib.SequenceMatcher(a=[e[2] for e in gold_words], b=[m[2] for m in mine], autojunk=False)
    pairs = [(gold_words[i + k], mine[j + k]) for i, j, n in sm.get_matching_blocks() for k in range(n)]
    assert len(pairs) == 57 and len(gold_words) == 58 and len(mine) == 59   # gold drops an "uh" and writes "word" once
    assert all(a[1] <= b[0] + 1e-9 for a, b in zip(mine, mine[1:])) and mine[0][0] >= 0 and mine[-1][1] <= len(fx.pcm) / sr + 1e-6
    dist = np.abs(np.array([m[0] - e[0] for e, m in pairs]))
    assert np.median(dist) > 1.0        # seconds: mono_model.zip carries no boundary information (see the module docstring)
    # and the exported file is the gold file's shape: two tiers, same duration, words in the same order
    (path,) = al.export_textgrids([utt], [res], tmp_path, output_format="short_textgrid")
    back = C.read_short_textgrid(path)
    assert list(back) == ["words", "phones"] and back["words"][-1][1] == gold["words"][-1][1] == 26.72325


@pytest.fixture(scope="module")
def golds():
    g = {n: RS.Gold(n) for n in RS.NAMES}
    return g, RS.build_lexicon(list(g.values()))


def device_feats(engine, pcm):
    so = np.array([0, len(pcm)], dtype=np.int64)
    mfcc, fo = engine.mfcc(torch.from_numpy(np.ascontiguousarray(pcm)).to(engine.device), so)
    u2s = np.zeros(1, dtype=np.int32)
    return engine.features(mfcc, fo, u2s, engine.cmvn_stats(mfcc, fo, u2s, 1)).cpu().numpy()


def device_intervals(engine, model, lex, gold):
    al = CorpusAligner(model.tm, model.am, model.tree, lex, options=AlignOptions(beam=100.0, retry_beam=400.0), engine=engine)
    (res,) = al.align([CorpusUtterance(gold.name, "spk", gold.pcm, gold.text)])
    assert res is not None, al.failure_reasons
    words = [w for w in res.ctm.word_intervals if w.label != lex.silence_word]
    return res, words, [p for w in words for p in w.phones]


def oracle_on_device_features(engine, model, lex, gold):
    gc = G.TrainingGraphCompiler(model.tm, model.tree, lex)
    fst = G.add_transition_probs(gc.compile_fst(gold.text), model.tm.scaled_log_probs(1.0, 0.1))
    return helpers.oracle_align_feats(model.tm, fst, device_feats(engine, gold.pcm), model.am, beam=100.0, retry_beam=400.0)


def test_device_chain_reproduces_the_gold_boundaries_closed_set(engine, golds):
    g, lex = golds
    engine.configure_mfcc()
    model = RS.train(list(g.values()), lex, lambda pcm: device_feats(engine, pcm))
    gold = g["acoustic_corpus"]
    res, words, phones = device_intervals(engine, model, lex, gold)
    rep = RS.boundary_report(gold, words, phones)
    check_bounds(rep, median_ms=10, frac20=0.85, signed_ms=8)
    assert np.mean(np.abs(rep["word_begin"]) <= 0.0501) >= 0.93
    ref = oracle_on_device_features(engine, model, lex, gold)
    assert ref["status"] in (0, 1) and np.array_equal(res.alignment, ref["ali"])
    assert abs(res.per_frame_likelihood - ref["like"] / len(ref["ali"])) < 1e-3


@pytest.mark.parametrize("held_out", ["cold_corpus", "cold_corpus3"])
def test_device_chain_on_held_out_gold_file(engine, golds, held_out):
    g, lex = golds
    engine.configure_mfcc()
    model = RS.train([g[n] for n in RS.NAMES if n != held_out], lex, lambda pcm: device_feats(engine, pcm))
    gold = g[held_out]
    res, words, phones = device_intervals(engine, model, lex, gold)
    check_bounds(RS.boundary_report(gold, words, phones), median_ms=20, frac20=0.55)
    ref = oracle_on_device_features(engine, model, lex, gold)
    assert np.array_equal(res.alignment, ref["ali"])
