"""Corpus-level driver (aligner.CorpusAligner): per-speaker CMVN over a shard, length-bucketed batches, alignment, word /
phone intervals, TextGrid files — the coarse flow of PretrainedAligner.align + export (MFA/alignment/mixins.py:282-380,
MFA/alignment/base.py:510-539, MFA/textgrid.py:463-572) — checked against the oracle run with the same per-speaker CMVN."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers

pytestmark = pytest.mark.gpu


def test_corpus_aligner_on_reference_fixture(engine, fx, tmp_path):
    from montreal_forced_aligner_amd import ctm as C
    from montreal_forced_aligner_amd.aligner import AlignOptions, CorpusAligner, CorpusUtterance

    sr = 16000
    cuts = [("spkA", 0.0, 4.2, "this is the acoustic corpus i'm talking pretty fast here"),
            ("spkA", 4.0, 6.5, "there's nothing going else going on"),
            ("spkB", 23.5, 26.72, "um and that should be all thanks")]
    utts = [CorpusUtterance(f"{s}-{k}", s, fx.pcm[int(a * sr): int(b * sr)], t, begin=a, file_name="acoustic_corpus",
                            file_duration=len(fx.pcm) / sr) for k, (s, a, b, t) in enumerate(cuts)]
    al = CorpusAligner(fx.mono_tm, fx.mono_am, fx.mono_tree, fx.mono_lex, options=AlignOptions(beam=100.0, retry_beam=400.0),
                       engine=engine)
    res = al.align(utts)
    assert al.failed == [] and all(r is not None for r in res)
    # oracle with the same per-speaker CMVN
    mf = [O.mfcc(u.pcm.astype(np.float32), O.default_mfcc_opts()) for u in utts]
    stats = {"spkA": O.cmvn_stats([mf[0], mf[1]]), "spkB": O.cmvn_stats([mf[2]])}
    am, tm = fx.mono_am, fx.mono_tm
    for u, r, m in zip(utts, res, mf):
        x = O.deltas(O.cmvn_apply(stats[u.speaker], m))
        fst = fx.mono_graph(u.text)
        pl = np.unique(tm.id2pdf[fst.arcs["ilabel"]])
        ref = helpers.oracle_align(tm, fst, O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl), pl,
                                   beam=100.0, retry_beam=400.0)
        assert np.array_equal(r.alignment, ref["ali"]) and np.array_equal(r.words, ref["words"])
        assert abs(r.per_frame_likelihood - ref["like"] / len(ref["ali"])) < 1e-3
        words = [w.label for w in r.ctm.word_intervals if w.label != fx.mono_lex.silence_word]
        assert words == u.text.split()
        assert r.ctm.word_intervals[0].begin >= u.begin - 1e-9   # shifted to file time
    paths = al.export_textgrids(utts, res, tmp_path / "out")
    assert [p.name for p in paths] == ["acoustic_corpus.TextGrid"]
    text = paths[0].read_text(encoding="utf8")
    assert 'name = "spkA - words"' in text and 'name = "spkB - phones"' in text
    assert text.count('text = "acoustic"') == 1 and 'xmax = %s' % round(len(fx.pcm) / sr, 6) in text
    # the one-rank form of the sharded entry point goes through the same speaker assignment and gather
    from montreal_forced_aligner_amd.aligner import align_sharded
    res_s = align_sharded(lambda: al, utts, rank=0, world_size=1)
    assert all(np.array_equal(a.alignment, b.alignment) for a, b in zip(res, res_s))
    # a transcript the lexicon cannot spell and audio too short to hold it: counted as failed, not raised
    bad = CorpusUtterance("spkB-x", "spkB", fx.pcm[: sr // 4], "this is the acoustic corpus i'm talking pretty fast here")
    res2 = al.align([utts[2], bad])
    assert res2[0] is not None and res2[1] is None and al.failed == ["spkB-x"]



def test_corpus_path_double_compression_flag(engine, fx):
    """`mfa align` aligns features that went through Kaldi's 8-bit CompressedMatrix twice (raw MFCC, then CMVN-applied:
    MFA/corpus/features.py:235, :356-365); AlignOptions(corpus_compression=True) reproduces both quantisations.  Checked
    against the oracle fed the same round-tripped features: identical alignment; and against the float path: the
    quantisation is visible but small."""
    from montreal_forced_aligner_amd import kaldi_io as K
    from montreal_forced_aligner_amd.aligner import AlignOptions, CorpusAligner, CorpusUtterance

    sr = 16000
    cuts = [("spkA", 0.0, 4.2, "this is the acoustic corpus i'm talking pretty fast here"),
            ("spkA", 4.0, 6.5, "there's nothing going else going on")]
    utts = [CorpusUtterance(f"{s}-{k}", s, fx.pcm[int(a * sr): int(b * sr)], t) for k, (s, a, b, t) in enumerate(cuts)]
    al = CorpusAligner(fx.mono_tm, fx.mono_am, fx.mono_tree, fx.mono_lex, engine=engine,
                       options=AlignOptions(beam=100.0, retry_beam=400.0, corpus_compression=True))
    res = al.align(utts, make_ctm=False)
    plain = CorpusAligner(fx.mono_tm, fx.mono_am, fx.mono_tree, fx.mono_lex, engine=engine,
                          options=AlignOptions(beam=100.0, retry_beam=400.0)).align(utts, make_ctm=False)
    assert all(r is not None for r in res) and all(r is not None for r in plain)
    am, tm = fx.mono_am, fx.mono_tm
    # the device MFCCs (what the aligner compressed), CMVN statistics from the round-tripped matrices, second round trip
    so = np.concatenate([[0], np.cumsum([len(u.pcm) for u in utts])]).astype(np.int64)
    import torch
    mfcc, fo = engine.mfcc(torch.from_numpy(np.concatenate([u.pcm for u in utts])).to(engine.device), so)
    mf = [K.compress_round_trip(mfcc.cpu().numpy()[fo[k]: fo[k + 1]]) for k in range(2)]
    stats = O.cmvn_stats(mf)
    moved = 0
    for u, r, p, m in zip(utts, res, plain, mf):
        x = O.deltas(K.compress_round_trip(O.cmvn_apply(stats, m)))
        fst = fx.mono_graph(u.text)
        pl = np.unique(tm.id2pdf[fst.arcs["ilabel"]])
        ref = helpers.oracle_align(tm, fst, O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl), pl,
                                   beam=100.0, retry_beam=400.0)
        assert np.array_equal(r.alignment, ref["ali"])
        assert abs(r.per_frame_likelihood - ref["like"] / len(ref["ali"])) < 1e-3
        assert abs(r.per_frame_likelihood - p.per_frame_likelihood) < 0.5     # same utterance, slightly different features
        moved += int((r.alignment != p.alignment).sum())
    assert moved < 0.2 * sum(len(r.alignment) for r in res)


def test_two_rank_sharded_corpus_to_textgrids(tmp_path):
    """BASELINE configs[3] at rehearsal scale (tools/sharded_corpus_demo.py): two ``torch.distributed`` ranks (gloo; on a
    one-GPU box they share the device), speakers assigned by the reference's rule, results gathered on the host, TextGrids
    written by rank 0 — alignments, likelihoods and files identical to the one-rank run."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(root / "tools" / "sharded_corpus_demo.py"), "--out", str(tmp_path), "--backend", "gloo"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rep = json.loads((tmp_path / "report.json").read_text())
    assert rep["world_size"] == 2 and rep["all_aligned"] and rep["identical_to_one_rank"] and rep["identical_files"]
    assert rep["textgrids"] == ["file0.TextGrid", "file1.TextGrid", "file2.TextGrid", "file3.TextGrid"]
    text = (tmp_path / "textgrids" / "file0.TextGrid").read_text(encoding="utf8")
    assert 'name = "spk0 - words"' in text and 'name = "spk4 - phones"' in text
