"""BASELINE-sized GPU checks (10 s / 16 kHz synthetic utterances, configs[1]-style monophone model estimated from the
generator): size-independent properties over the whole batch plus an oracle spot check.

Properties: every utterance aligns at beam 10; one transition-id per frame; the alignment splits into complete phones that
spell the transcript; word ids equal the transcript; phone boundaries land within two frames of the generator's ground
truth; a second run is bit-identical (no order-dependent atomics leak into results); a shuffled batch order gives the same
per-utterance results."""
import numpy as np
import pytest
import torch

from montreal_forced_aligner_amd import ctm as C
from montreal_forced_aligner_amd import graph as G
from oracle import oracle as O
from tests import helpers, synth

pytestmark = pytest.mark.gpu

N_UTT = 96


@pytest.fixture(scope="module")
def world_and_model(engine):
    world = synth.SynthWorld.build()
    engine.configure_mfcc()

    def feature_fn(pcm, spk):
        sample_off = np.array([0, len(pcm)], dtype=np.int64)
        mfcc, frame_off = engine.mfcc(torch.from_numpy(pcm.copy()).to(engine.device), sample_off)
        u2s = np.zeros(1, dtype=np.int32)
        return engine.features(mfcc, frame_off, u2s, engine.cmvn_stats(mfcc, frame_off, u2s, 1)).cpu().numpy()

    model = synth.train_monophone(world, feature_fn, n_train=60)
    return world, model


def _run(engine, world, model, order):
    utts = [world.utterance(i) for i in order]
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    engine.load_gmm(model.am)
    pcm = torch.from_numpy(np.concatenate([u[0] for u in utts])).to(engine.device)
    sample_off = np.arange(len(utts) + 1, dtype=np.int64) * synth.UTT_SAMPLES
    mfcc, frame_off = engine.mfcc(pcm, sample_off)
    spk = np.array([u[3] for u in utts])
    ids, inv = np.unique(spk, return_inverse=True)
    stats = engine.cmvn_stats(mfcc, frame_off, inv.astype(np.int32), len(ids))
    feats = engine.features(mfcc, frame_off, inv.astype(np.int32), stats)
    graphs = engine.pack_graphs(fsts, model.tm)
    ll, ll_off, ll_cols = engine.score(feats, frame_off, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts)
    res = engine.align(graphs, ll, ll_off, ll_cols, frame_off, beam=10.0, retry_beam=40.0, max_tokens=256,
                       bp_tokens_per_frame=128)
    out = {k: (v.cpu().numpy() if v is not None else None) for k, v in res.items()}
    out.update(frame_off=frame_off, utts=utts, fsts=fsts, feats=feats.cpu().numpy(), graphs=graphs)
    return out


def test_full_size_batch_properties(engine, world_and_model):
    world, model = world_and_model
    order = list(range(N_UTT))
    r = _run(engine, world, model, order)
    assert np.all(r["status"] == 0), np.unique(r["status"], return_counts=True)
    fo = r["frame_off"]
    errs = []
    for u in range(N_UTT):
        a, b = fo[u], fo[u + 1]
        assert b - a == 1000
        ali = r["ali"][a:b]
        assert np.all(ali > 0)
        phones = C.split_to_phones(ali, model.tm)  # raises unless the path is a sequence of complete phones
        assert sum(n for _, n, _ in phones) == 1000
        pcm, text, segs, _spk = r["utts"][u]
        words = [world.lexicon.word_table.find(int(w)) for w in r["words"][a: a + r["n_words"][u]]]
        assert words == text.split()
        got = [(world.lexicon.phone_table.find(p), f * 0.01) for f, n, p in phones if world.lexicon.phone_table.find(p) != "sil"]
        truth = [(p, s / 16000.0) for p, s, e in segs if p != "sil"]
        assert [g[0] for g in got] == [t[0] for t in truth]
        errs += [abs(g[1] - t[1]) for g, t in zip(got, truth)]
    errs = np.array(errs)
    assert errs.mean() < 0.01 and np.percentile(errs, 99) < 0.03, (errs.mean(), errs.max())

    # determinism and batch-order independence
    r2 = _run(engine, world, model, order)
    assert np.array_equal(r["ali"], r2["ali"]) and np.array_equal(r["like"], r2["like"])
    perm = list(np.random.default_rng(0).permutation(N_UTT))
    r3 = _run(engine, world, model, perm)
    for k, u in enumerate(perm):
        assert np.array_equal(r3["ali"][r3["frame_off"][k]: r3["frame_off"][k + 1]], r["ali"][fo[u]: fo[u + 1]])
        assert r3["like"][k] == r["like"][u]

    # oracle spot check on the device's own features (decoder parity at full size)
    am = model.am
    for u in (0, 17, 95):
        x = r["feats"][fo[u]: fo[u + 1]]
        pl = r["graphs"].pdf_lists_host[u]
        ref = helpers.oracle_align(model.tm, r["fsts"][u], O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars,
                                                                          am.pdf_offsets, pl), pl, beam=10.0, retry_beam=40.0)
        assert ref["status"] == 0 and np.array_equal(ref["ali"], r["ali"][fo[u]: fo[u + 1]])
        assert abs(ref["like"] - r["like"][u]) / 1000 < 1e-3
