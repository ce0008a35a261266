"""CPU tests of the interval / TextGrid layer."""
import numpy as np

from montreal_forced_aligner_amd import ctm as C
from oracle import oracle as O
from tests import helpers


def test_textgrid_short_format_round_trip_matches_reference_fixture(fx, tmp_path):
    """The reference ships tests/data/textgrid/acoustic_corpus.TextGrid (Praat short format, xmax 26.72325, 71 word
    intervals + a phone tier).  Re-exporting its labelled intervals must reproduce the file line for line: same blank-interval
    insertion, same tier headers, same quoting."""
    src = helpers.REF / "acoustic_corpus.TextGrid"
    tiers = C.read_short_textgrid(src)
    assert list(tiers) == ["words", "phones"] and len(tiers["words"]) == 71 and len(tiers["phones"]) == 216
    data = {"spk": {name: [C.CtmInterval(b, e, lab) for b, e, lab in ent if lab != ""] for name, ent in tiers.items()}}
    out = tmp_path / "out.TextGrid"
    C.export_textgrid(data, out, 26.72325, 0.01, "short_textgrid")
    # line-for-line identical, except that the fixture's authoring tool printed whole numbers as "0"/"6" where MFA's
    # writer (and this one) prints Python floats "0.0"/"6.0"
    a, b = out.read_text().split("\n"), src.read_text().split("\n")
    assert len(a) == len(b) == 879
    diff = [(x, y) for x, y in zip(a, b) if x != y]
    assert diff and all(float(x) == float(y) and y == str(int(float(y))) for x, y in diff), diff
    long_out = tmp_path / "long.TextGrid"
    C.export_textgrid(data, long_out, 26.72325, 0.01, "long_textgrid")
    txt = long_out.read_text()
    assert txt.startswith('File type = "ooTextFile"\nObject class = "TextGrid"\n\nxmin = 0 \nxmax = 26.72325 \n')
    assert '        intervals [2]:\n            xmin = 1.05 \n            xmax = 1.2 \n            text = "this" \n' in txt


def test_last_interval_snaps_to_file_end_and_overlaps_are_clipped(tmp_path):
    iv = [C.CtmInterval(0.0, 0.5, "a"), C.CtmInterval(0.49, 0.995, "b")]
    out = tmp_path / "t.TextGrid"
    C.export_textgrid({"s": {"words": iv}}, out, 1.0, 0.01, "short_textgrid")
    t = C.read_short_textgrid(out)["words"]
    assert t == [(0.0, 0.5, "a"), (0.5, 1.0, "b")]  # MFA/textgrid.py:548-556


def test_generate_ctm_and_word_grouping_on_an_oracle_alignment(fx):
    tm = fx.mono_tm
    rng = np.random.default_rng(9)
    text = "this is the acoustic corpus"
    f = fx.mono_graph(text)
    pdfs = np.arange(tm.num_pdfs, dtype=np.int32)
    ll = rng.normal(-60.0, 10.0, size=(200, tm.num_pdfs)).astype(np.float32)
    r = helpers.oracle_align(tm, f, ll, pdfs, beam=1e4, retry_beam=0.0)
    assert r["status"] == 0
    # host SplitToPhones == oracle SplitToPhones
    ref, ok = O.split_to_phones(r["ali"], tm.id2state, tm.is_self_loop, tm.is_final, tm.tuples)
    assert ok
    got = C.split_to_phones(r["ali"], tm)
    assert [tuple(map(int, x)) for x in ref] == got
    iv = C.generate_ctm(r["ali"], tm, fx.mono_lex.phone_table, 0.01)
    assert iv[0].begin == 0.0 and iv[-1].end == 2.0 and all(a.end == b.begin for a, b in zip(iv, iv[1:]))
    h = C.phones_to_pronunciations(fx.mono_lex, r["words"], iv, text=text)
    words = [w.label for w in h.word_intervals if w.label != fx.mono_lex.silence_word]
    assert words == text.split()
    assert [w.pronunciation for w in h.word_intervals if w.label == "acoustic"] == ["ah k uw s t ih k"]
    assert sum(len(w.phones) for w in h.word_intervals) == len(iv)
    h.update_utterance_boundaries(1.5, 3.49)
    assert abs(h.word_intervals[0].begin - 1.5) < 1e-9 and h.word_intervals[-1].end <= 3.49
