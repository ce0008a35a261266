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


def _toy_lexicon(position_dependent):
    from montreal_forced_aligner_amd import graph as G

    lex = G.LexiconCompiler(position_dependent_phones=position_dependent, phones=["a", "b", "c", "d"], silence_phone="sil",
                            oov_phone="spn")
    for w, p in (("ab", "a b"), ("ab", "a"), ("c", "c"), ("bc", "b c"), ("abc", "a b c"), ("d", "d"), ("d", "d d")):
        lex.add_pronunciation(G.Pronunciation(w, p))
    lex.build_phone_table()
    return lex


def _ivs(labels):
    return [C.CtmInterval(0.1 * i, 0.1 * (i + 1), lab, 0) for i, lab in enumerate(labels)]


def test_word_grouping_backtracks_when_a_variant_is_a_prefix_of_another():
    """Non-position-dependent models (english_mfa style): "ab" may be spelt "a b" or "a"; for the word sequence [ab, bc]
    over phones a b c the greedy longest-first choice (a b) leaves "c" for "bc" and fails — the search must back off to
    "a" + "b c" (MFA/alignment/multiprocessing.py:1741-1747 composes with the align lexicon, which finds it)."""
    lex = _toy_lexicon(False)
    ids = [lex.to_int("ab"), lex.to_int("bc")]
    h = C.phones_to_pronunciations(lex, ids, _ivs(["sil", "a", "b", "c", "sil"]))
    got = [(w.label, w.pronunciation) for w in h.word_intervals]
    assert got == [("<eps>", "sil"), ("ab", "a"), ("bc", "b c"), ("<eps>", "sil")]
    # unambiguous input still takes the longest variant, silences between words become <eps> entries
    h = C.phones_to_pronunciations(lex, [lex.to_int("ab"), lex.to_int("c")], _ivs(["a", "b", "sil", "c"]))
    assert [(w.label, w.pronunciation) for w in h.word_intervals] == [("ab", "a b"), ("<eps>", "sil"), ("c", "c")]
    # repeated single-phone variants: d ("d" | "d d") twice over d d d has two spellings; one is returned, all phones covered
    h = C.phones_to_pronunciations(lex, [lex.to_int("d"), lex.to_int("d")], _ivs(["d", "d", "d"]))
    assert sorted(w.pronunciation for w in h.word_intervals) == ["d", "d d"]
    import pytest
    with pytest.raises(C.CtmError):
        C.phones_to_pronunciations(lex, [lex.to_int("ab")], _ivs(["a", "c"]))


def test_word_grouping_uses_word_position_suffixes():
    """Position-dependent phones: the aligned labels say where words end (_E/_S), so [ab, c] over a_B b_E c_S groups as
    "a b" + "c" even though "a" alone is also a variant of ab — and a_S b_B c_E forces "a" + "b c"."""
    lex = _toy_lexicon(True)
    h = C.phones_to_pronunciations(lex, [lex.to_int("ab"), lex.to_int("c")], _ivs(["a_B", "b_E", "c_S"]))
    assert [(w.label, w.pronunciation) for w in h.word_intervals] == [("ab", "a b"), ("c", "c")]
    h = C.phones_to_pronunciations(lex, [lex.to_int("ab"), lex.to_int("bc")], _ivs(["a_S", "sil", "b_B", "c_E"]))
    assert [(w.label, w.pronunciation) for w in h.word_intervals] == [("ab", "a"), ("<eps>", "sil"), ("bc", "b c")]
    import pytest
    with pytest.raises(C.CtmError):   # a_B cannot end a word
        C.phones_to_pronunciations(lex, [lex.to_int("ab"), lex.to_int("bc")], _ivs(["a_B", "b_B", "c_E"]))
    # the out-of-vocabulary word is a one-phone word: spn_S (tests/data/dictionaries/expected/lexicon.text.fst)
    h = C.phones_to_pronunciations(lex, [lex.to_int("zzz"), lex.to_int("c")], _ivs(["spn_S", "c_S"]))
    assert [(w.label, w.pronunciation) for w in h.word_intervals] == [("<unk>", "spn"), ("c", "c")]


def test_fix_unk_words_restores_transcript_spelling():
    """MFA/helper.py:772-833 at MFA/alignment/multiprocessing.py:1749-1751: <unk> intervals get the transcript's word,
    silence intervals stay, known words are untouched."""
    lex = _toy_lexicon(False)
    h = C.phones_to_pronunciations(lex, [lex.to_int("ab"), lex.to_int("zork"), lex.to_int("c"), lex.to_int("blip")],
                                   _ivs(["sil", "a", "b", "spn", "sil", "c", "spn"]))
    assert [w.label for w in h.word_intervals] == ["<eps>", "ab", "<unk>", "<eps>", "c", "<unk>"]
    fixed = C.fix_unk_words("ab zork c blip".split(), h.word_intervals, lex)
    assert [w.label for w in fixed] == ["<eps>", "ab", "zork", "<eps>", "c", "blip"]
    assert [len(w.phones) for w in fixed] == [1, 2, 1, 1, 1, 1]
    # a transcript with more words than intervals (a word the decoder dropped) leaves the rest aligned
    fixed = C.fix_unk_words("ab extra zork c blip".split(), C.phones_to_pronunciations(
        lex, [lex.to_int("ab"), lex.to_int("zork"), lex.to_int("c"), lex.to_int("blip")],
        _ivs(["a", "b", "spn", "c", "spn"])).word_intervals, lex)
    assert [w.label for w in fixed][0] == "ab" and [w.label for w in fixed][-2:] == ["c", "blip"]


def test_split_to_phones_array_form_equals_the_frame_loop(fx):
    """``split_to_phones`` (array form) against ``_split_to_phones_loop`` (Kaldi's rule frame by frame): random complete
    paths through training graphs give the same phones; corrupted alignments raise in both."""
    import pytest

    rng = np.random.default_rng(11)
    tm = fx.mono_tm
    texts = ["this is the acoustic corpus", "um and that should be all thanks", "there's nothing going else going on", ""]
    n_ok = n_bad = 0
    for text in texts:
        f = fx.mono_gc.compile_fst(text)
        for _ in range(25):
            s, seq = 0, []
            while True:
                a0, a1 = int(f.arc_offsets[s]), int(f.arc_offsets[s + 1])
                if np.isfinite(f.final[s]) and (a0 == a1 or rng.random() < 0.3):
                    break
                arc = f.arcs[int(rng.integers(a0, a1))]
                seq.append(int(arc["ilabel"]))
                s = int(arc["nextstate"])
            ali = np.asarray(seq, dtype=np.int32)
            assert C.split_to_phones(ali, tm) == C._split_to_phones_loop(ali, tm)
            n_ok += 1
            if len(seq) > 4:
                for kind in range(3):
                    bad = ali.copy()
                    k = int(rng.integers(1, len(seq) - 1))
                    if kind == 0:
                        bad = bad[:k]                                   # may end inside a phone
                    elif kind == 1:
                        bad[k] = int(rng.integers(1, tm.num_transition_ids + 1))
                    else:
                        bad[k:] = bad[k:][::-1]
                    try:
                        ref = C._split_to_phones_loop(bad, tm)
                    except C.CtmError:
                        with pytest.raises(C.CtmError):
                            C.split_to_phones(bad, tm)
                        n_bad += 1
                    else:
                        assert C.split_to_phones(bad, tm) == ref
    assert n_ok == 100 and n_bad > 20


def test_frame_times_are_pythons_rounding():
    """generate_ctm's interval ends are ``round(frame * frame_shift, 6)`` (MFA/data.py:2074-2075 rounds to 6 places): the
    array form returns the same doubles for whole-microsecond shifts and takes the scalar path otherwise."""
    rng = np.random.default_rng(4)
    frames = np.concatenate([np.arange(0, 4000), rng.integers(0, 10 ** 7, size=4000)])
    for shift in (0.01, 0.0125, 0.005, 0.02, 0.010001, 0.0100001, 1.0 / 3.0):
        want = [round(int(f) * shift, 6) for f in frames]
        assert C._frame_times(frames, shift) == want, shift


def test_interval_stage_failure_concerns_one_utterance_only(fx, monkeypatch):
    """The reference catches per utterance in its extraction loop (MFA/alignment/multiprocessing.py:1739-1770): an
    alignment whose word ids cannot be spelt must not take the other utterances' results with it."""
    from montreal_forced_aligner_amd.aligner import CorpusAligner, CorpusUtterance

    class StubEngine:            # the interval stage is host code: no device needed
        device = "cpu"

        def configure_mfcc(self, **kw):
            pass

        def load_gmm(self, am):
            pass

        def num_frames(self, n_samples):
            return n_samples // 160

    tm = fx.mono_tm
    text = "this is the acoustic corpus"
    f = fx.mono_graph(text)
    pdfs = np.arange(tm.num_pdfs, dtype=np.int32)
    ll = np.random.default_rng(3).normal(-60.0, 10.0, size=(200, tm.num_pdfs)).astype(np.float32)
    r = helpers.oracle_align(tm, f, ll, pdfs, beam=1e4, retry_beam=0.0)
    from montreal_forced_aligner_amd.aligner import _BatchOut

    T, nw = 200, len(r["words"])
    words = np.zeros(3 * T, dtype=np.int32)
    for k, w in enumerate((r["words"], r["words"][::-1], r["words"])):     # the middle one: word ids that do not spell the phones
        words[k * T: k * T + nw] = w
    bo = _BatchOut([0, 1, 2], np.array([0, T, 2 * T, 3 * T]), np.tile(r["ali"], 3).astype(np.int32), words,
                   np.full(3, nw, dtype=np.int32), np.full(3, r["like"], dtype=np.float32), np.zeros(3, dtype=np.int32))
    al = CorpusAligner(tm, fx.mono_am, fx.mono_tree, fx.mono_lex, engine=StubEngine())
    monkeypatch.setattr(al, "speaker_cmvn", lambda utts: ({"s": 0}, None))
    monkeypatch.setattr(al, "_submit_compile", lambda utts, idx: None)
    monkeypatch.setattr(al, "_pass", lambda utts, spk_ids, cmvn, fmllr, **kw: ([(bo, 0), (bo, 1), (bo, 2)], []))
    pcm = np.zeros(32000, dtype=np.int16)
    res = al.align([CorpusUtterance(f"s-{k}", "s", pcm, text) for k in range(3)])
    assert [x is not None for x in res] == [True, True, True] and al.failed == []
    assert res[0].ctm is not None and res[2].ctm is not None and res[1].ctm is None
    assert al.ctm_failed == ["s-1"] and "interval extraction failed" in al.failure_reasons["s-1"]
    assert np.array_equal(res[1].alignment, r["ali"])                 # the alignment itself is kept
