"""Randomised twin test of the native interval / file layer (libmfa_intervals.so) against the Python specification (ctm.py):
random transcripts from the fixture lexicon (with out-of-vocabulary words), random frame counts, random — awkward — utterance
offsets, random grouping into files and speakers; objects and file bytes must be identical.  CPU only.
python tools/intervals_fuzz.py [n_cases] [first_seed]"""
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, ".")
import numpy as np                                                       # noqa: E402

from montreal_forced_aligner_amd import ctm as C                         # noqa: E402
from montreal_forced_aligner_amd import intervals_native as N            # noqa: E402
from tests import helpers                                                # noqa: E402
from tests import test_intervals_native_cpu as T                         # noqa: E402

fx = helpers.Fixtures()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
vocab = [w for w in fx.mono_lex._by_word.keys() if w.isalpha() or "'" in w]
ex = N.IntervalExtractor(fx.mono_tm, fx.mono_lex, 0.01)
bad = 0
tmp = Path(tempfile.mkdtemp())
for seed in range(seed0, seed0 + n_cases):
    rng = np.random.default_rng(31000 + seed)
    n_utt = int(rng.integers(1, 7))
    texts, frames = [], []
    for _ in range(n_utt):
        n_w = int(rng.integers(1, 9))
        ws = [str(rng.choice(vocab)) if rng.random() > 0.15 else "".join(rng.choice(list("qxzv"), size=int(rng.integers(2, 6)))) for _ in range(n_w)]
        texts.append(" ".join(ws))
        frames.append(int(rng.integers(20 * n_w + 10, 60 * n_w + 40)))
    try:
        res = T._oracle_alignments(fx, texts, frames, seed=seed)
    except AssertionError:
        continue
    fo, ali, words, nw = T._pack(res)
    b = ex.extract(fo, ali, words, nw)
    try:
        # objects
        for u, (r, text) in enumerate(zip(res, texts)):
            begin = float(rng.choice([0.0, 0.1 + 0.2, 1e-5, 1.0 / 3.0, float(rng.random() * 50), 12345.678901234]))
            end = None if rng.random() < 0.3 else begin + len(r["ali"]) * 0.01 - float(rng.choice([0.0, 0.004, 0.0049, 1e-7]))
            ivs = C.generate_ctm(r["ali"], fx.mono_tm, fx.mono_lex.phone_table, 0.01)
            ref = C.phones_to_pronunciations(fx.mono_lex, r["words"], ivs, text=text)
            ref.update_utterance_boundaries(begin, end)
            ref.word_intervals = C.fix_unk_words(text.split(), ref.word_intervals, fx.mono_lex)
            T._same_ctm(b.ctm(u, text=text, begin=begin, end=end), ref)
        # files: utterances dealt onto files and speakers, laid out one after the other (sometimes overlapping slightly)
        dur = [len(r["ali"]) * 0.01 for r in res]
        n_files = int(rng.integers(1, n_utt + 1))
        cursor = {}
        utts = []
        for k in range(n_utt):
            name = f"f{int(rng.integers(0, n_files))}"
            spk = str(rng.choice(["anna", 'bob "b", jr', "carl"]))
            t0 = cursor.get(name, 0.0) + float(rng.choice([0.0, 0.25, 1.0 / 3.0, -0.004]))
            t0 = max(t0, 0.0)
            utts.append([name, spk, t0, t0 + dur[k] - float(rng.choice([0.0, 0.003])), None])
            cursor[name] = t0 + dur[k]
        for k in range(n_utt):
            utts[k][4] = cursor[utts[k][0]] + float(rng.choice([0.0, 0.015, 1.5]))
        utts = [tuple(u) for u in utts]
        fmt = str(rng.choice(["long_textgrid", "short_textgrid", "json", "csv"]))
        cleanup = bool(rng.random() < 0.7)
        try:
            ref_files = T._python_files(fx, res, texts, utts, fmt, tmp, cleanup)
            ref_err = None
        except Exception as e:           # the reference writer raises on a collapsed interval: the native one reports a code
            ref_files, ref_err = None, e
        files, order = [], {}
        for k, (name, spk, begin, end, fdur) in enumerate(utts):
            if name not in order:
                order[name] = len(files)
                files.append(dict(name=name, duration=0.0, speakers=[]))
            f = files[order[name]]
            f["duration"] = max(f["duration"], fdur or end)
            for s in f["speakers"]:
                if s[0] == spk:
                    s[1].append(k)
                    break
            else:
                f["speakers"].append((spk, [k]))
        relabel = b.relabels(texts)
        out, codes = ex.write_files(b, files, np.array([u[2] for u in utts]), np.array([u[3] for u in utts]), relabel, fmt, cleanup)
        if ref_err is not None:
            assert any(c != 0 for c in codes), f"python writer raised {ref_err!r}, native codes {codes}"
        else:
            for f, t, c in zip(files, out, codes):
                if c != 0:
                    raise AssertionError(f"native code {c} for {f['name']} where the python writer wrote a file")
                assert t == ref_files[f["name"]], f"{f['name']} ({fmt}, cleanup={cleanup}) differs"
        print(seed, "ok", n_utt, fmt, flush=True)
    except AssertionError as e:
        bad += 1
        print(seed, "MISMATCH", str(e)[:300], flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
