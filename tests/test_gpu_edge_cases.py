"""Edge cases at the C-ABI boundary: empty utterances inside a batch, inputs the kernels refuse (loudly, never silently),
single-frame utterances, and a batch-order permutation including the degenerate members."""
import ctypes as C

import numpy as np
import pytest
import torch

from montreal_forced_aligner_amd import _lib
from montreal_forced_aligner_amd import kaldi_io as K
from oracle import oracle as O
from tests import helpers

pytestmark = pytest.mark.gpu


def _dev(engine, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(engine.device)


def test_empty_and_tiny_utterances_inside_a_batch(engine, fx):
    """snip_edges=True gives ZERO frames for audio shorter than a window: such an utterance must flow through every stage
    as an empty range (status = failed, nothing written for it) without disturbing its neighbours; a one-frame utterance
    aligns only if its graph allows a one-frame path (it does not: failed, not a crash)."""
    tm, am = fx.mono_tm, fx.mono_am
    sr = 16000
    segs = [fx.pcm[: int(4.2 * sr)], fx.pcm[:200], fx.pcm[int(4.0 * sr): int(6.5 * sr)], fx.pcm[:400]]
    texts = ["this is the acoustic corpus i'm talking pretty fast here", "this", "there's nothing going else going on", "this"]
    engine.configure_mfcc(snip_edges=1)
    engine.load_gmm(am)
    try:
        so = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
        mfcc, fo = engine.mfcc(_dev(engine, np.concatenate(segs)), so)
        assert list(np.diff(fo)) == [418, 0, 248, 1]
        spk = np.arange(4, dtype=np.int32)
        feats = engine.features(mfcc, fo, spk, engine.cmvn_stats(mfcc, fo, spk, 4))
        fsts = [fx.mono_graph(t) for t in texts]
        graphs = engine.pack_graphs(fsts, tm)
        ll, ll_off, ll_cols = engine.score(feats, fo, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts,
                                           pdf_first_frame=graphs.pdf_first_frame)
        res = engine.align(graphs, ll, ll_off, ll_cols, fo, beam=100.0, retry_beam=400.0)
        status = res["status"].cpu().numpy()
        assert status[1] == 2 and status[3] == 2 and status[0] in (0, 1) and status[2] in (0, 1)
        ali = res["ali"].cpu().numpy()
        for u in (0, 2):   # neighbours equal their stand-alone alignment through the oracle
            x = O.deltas(O.cmvn_apply(O.cmvn_stats([O.mfcc(segs[u].astype(np.float32), O.default_mfcc_opts(snip_edges=1))]),
                                      O.mfcc(segs[u].astype(np.float32), O.default_mfcc_opts(snip_edges=1))))
            pl = graphs.pdf_lists_host[u]
            ref = helpers.oracle_align(tm, fsts[u], O.gmm_loglikes(x, am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets, pl),
                                       pl, beam=100.0, retry_beam=400.0)
            assert np.array_equal(ali[fo[u]: fo[u + 1]], ref["ali"])
    finally:
        engine.configure_mfcc()


def test_unsupported_inputs_are_refused_loudly(engine, fx):
    lib, ctx = engine.lib, engine.ctx
    # more utterances than a launch can index
    rc = lib.mfa_mfcc_batch(ctx, None, None, None, 70000, 10, None)
    assert rc != 0 and b"65535" in lib.mfa_last_error(ctx)
    # beams that make no sense (AlignUtteranceWrapper's own check)
    with pytest.raises(_lib.MfaHipError):
        g = engine.pack_graphs([fx.mono_graph("this")], fx.mono_tm)
        engine.load_gmm(fx.mono_am)
        x = _dev(engine, np.zeros((5, 39), np.float32))
        fo = np.array([0, 5], np.int64)
        ll, ll_off, ll_cols = engine.score(x, fo, g.pdf_list, g.pdf_off_host, g.class_counts)
        engine.align(g, ll, ll_off, ll_cols, fo, beam=10.0, retry_beam=5.0)
    # graphs the device decoder does not take: states with more than 64 arcs of a kind, labels outside the model
    # (epsilon input labels are taken since round 3: tests/test_gpu_general.py)
    f = fx.mono_graph("this")
    arcs = f.arcs.copy()
    arcs["ilabel"][0] = 0
    g_eps = engine.pack_graphs([K.Fst(f.start, f.arc_offsets, arcs, f.final)], fx.mono_tm)
    assert "state_nemit" in g_eps.tensors
    arcs["ilabel"][0] = fx.mono_tm.num_transition_ids + 5
    with pytest.raises(_lib.MfaHipError, match="transition-ids"):
        engine.pack_graphs([K.Fst(f.start, f.arc_offsets, arcs, f.final)], fx.mono_tm)
    n = 70
    wide = np.zeros(n, dtype=K.ARC_DTYPE)
    wide["ilabel"] = 1; wide["nextstate"] = 1
    with pytest.raises(_lib.MfaHipError, match="64"):
        engine.pack_graphs([K.Fst(0, np.array([0, n, n], np.int64), wide, np.array([np.inf, 0.0], np.float32))], fx.mono_tm)
    # a feature window that is not 512 points is outside the MFCC kernel
    with pytest.raises(_lib.MfaHipError):
        engine.configure_mfcc(frame_length_ms=40.0)
    engine.configure_mfcc()
