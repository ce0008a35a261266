"""Host side of lazy scoring (no GPU): mfa_build_score_plan's columns and depth keys, and the claim the device path rests
on — for ANY set of live states and any window length K, every arc a token can take within the window (an arc leaving a
state reachable in fewer than K arcs) has its column inside the index range the band rule selects.  Brute force on random
graphs (cycles, skips, dead ends, unreachable states) and on a real training graph."""
import ctypes as C

import numpy as np

from montreal_forced_aligner_amd import _lib
from tests.test_gpu_parity import _random_graph  # plain numpy graph generator (module import needs no GPU)


def _plan(f, pdf_of_arc, pdf_class, span, groups=1):
    lib = _lib.lib()
    na = f.num_arcs
    off = np.ascontiguousarray(f.arc_offsets, dtype=np.int32)
    nxt = np.ascontiguousarray(f.arcs["nextstate"], dtype=np.int32)
    pdf = np.ascontiguousarray(pdf_of_arc, dtype=np.int32)
    cls = np.ascontiguousarray(pdf_class, dtype=np.int32)
    sd = np.empty((f.num_states, 2), np.int32)
    col, cp, cf, cl = (np.empty(max(na, 1), np.int32) for _ in range(4))
    cc = np.zeros(6, np.int32)
    n = C.c_int32(0)
    gc = np.zeros(groups, np.int32)
    if groups == 1:
        rc = lib.mfa_build_score_plan(f.num_states, off.ctypes.data, nxt.ctypes.data, pdf.ctypes.data, int(f.start), len(cls),
                                      cls.ctypes.data, span, sd.ctypes.data, col.ctypes.data, cp.ctypes.data, cf.ctypes.data,
                                      cl.ctypes.data, cc.ctypes.data, C.byref(n))
        gc[0] = cc[0]
    else:
        rc = lib.mfa_build_score_plan_grouped(f.num_states, off.ctypes.data, nxt.ctypes.data, pdf.ctypes.data, int(f.start),
                                              len(cls), cls.ctypes.data, span, groups, sd.ctypes.data, col.ctypes.data,
                                              cp.ctypes.data, cf.ctypes.data, cl.ctypes.data, cc.ctypes.data, gc.ctypes.data,
                                              C.byref(n))
    assert rc == 0
    k = n.value
    return sd, col[:na], cp[:k], cf[:k], cl[:k], cc, gc


def _check_graph(rng, f, pdf_of_arc, pdf_class, span, trials=30, groups=1):
    sd, col, cp, cf, cl, cc, gc = _plan(f, pdf_of_arc, pdf_class, span, groups)
    # runs the band rule is applied to: class 0 split into `groups` runs (pdf id mod groups), the other classes whole
    assert gc.sum() == cc[0]
    run_sizes = list(gc) + list(cc[1:])
    run_class = [0] * groups + [1, 2, 3, 4, 5]
    run_group = list(range(groups)) + [0] * 5
    S = f.num_states
    src = np.repeat(np.arange(S), np.diff(f.arc_offsets))
    nxt = f.arcs["nextstate"].astype(np.int64)
    # columns: right pdf, class-sorted, keys non-decreasing inside a class, first = min depth of the column's sources
    assert np.array_equal(cp[col], pdf_of_arc)
    assert cc.sum() == len(cp)
    bounds = np.concatenate([[0], np.cumsum(run_sizes)])
    for k in range(len(run_sizes)):
        seg = slice(bounds[k], bounds[k + 1])
        assert np.all(pdf_class[cp[seg]] == run_class[k])
        if run_class[k] == 0:
            assert np.all(cp[seg] % groups == run_group[k])
        assert np.all(np.diff(cf[seg]) >= 0) and np.all(np.diff(cl[seg]) >= 0)
    # BFS depth by relaxation; reachable set of the start state
    INF = 1 << 30
    d = np.full(S, INF, np.int64)
    d[f.start] = 0
    for _ in range(S + 1):
        nd = d.copy()
        np.minimum.at(nd, nxt, d[src] + 1)
        if np.array_equal(nd, d):
            break
        d = nd
    reach0 = d < INF
    assert np.array_equal(sd[reach0, 0], d[reach0])
    first = np.full(len(cp), INF, np.int64)
    np.minimum.at(first, col, np.where(reach0[src], d[src], INF))
    live_cols = first < INF
    assert np.array_equal(cf[live_cols], first[live_cols])
    # m = smallest depth reachable: non-decreasing along arcs between reachable states, and <= own depth
    m = sd[:, 1].astype(np.int64)
    ok = reach0[src]
    assert np.all(m[nxt[ok]] >= m[src[ok]]) and np.all(m[reach0] <= d[reach0])
    # brute force: random live sets, random K
    adj = [nxt[f.arc_offsets[s]: f.arc_offsets[s + 1]] for s in range(S)]
    run_of_col = np.searchsorted(bounds, np.arange(len(cp)), side="right") - 1
    pos_in_run = np.arange(len(cp)) - bounds[run_of_col]
    states = np.nonzero(reach0)[0]
    for _ in range(trials):
        K = int(rng.choice([1, 2, 5, 16, 64]))
        live = rng.choice(states, size=min(len(states), int(rng.integers(1, 12))), replace=False)
        lo, hi = int(m[live].min()), int(d[live].max()) + K - 1
        frontier, seen = set(int(x) for x in live), set(int(x) for x in live)
        for _step in range(K - 1):            # states a token can sit on at frames t0 .. t0+K-1
            nf = set()
            for s in frontier:
                for t in adj[s]:
                    if int(t) not in seen:
                        seen.add(int(t)); nf.add(int(t))
            frontier = nf
        need = set()
        for s in seen:
            need.update(int(c) for c in col[f.arc_offsets[s]: f.arc_offsets[s + 1]])
        for k in range(len(run_sizes)):
            seg = slice(bounds[k], bounds[k + 1])
            lo_idx = int((cl[seg] < lo).sum())
            hi_idx = int((cf[seg] <= hi).sum())
            for c_ in need:
                if run_of_col[c_] == k:
                    assert lo_idx <= pos_in_run[c_] < hi_idx, (K, lo, hi, c_, cf[c_], cl[c_])


def test_band_rule_is_a_superset_on_random_graphs(fx):
    rng = np.random.default_rng(77)
    tm = fx.mono_tm
    pdf_class = rng.integers(0, 6, size=tm.num_pdfs).astype(np.int32)
    for trial in range(25):
        f = _random_graph(rng, tm, int(rng.choice([3, 8, 40, 150])))
        pdf_of_arc = tm.id2pdf[f.arcs["ilabel"]].astype(np.int32)
        _check_graph(rng, f, pdf_of_arc, pdf_class, span=int(rng.choice([0, 2, 8, 32])), groups=int(rng.choice([1, 2, 8, 16])))


def test_band_rule_on_a_training_graph_and_column_clusters(fx):
    rng = np.random.default_rng(5)
    tm = fx.mono_tm
    f = fx.mono_graph("this is the acoustic corpus i'm talking pretty fast here this is the acoustic corpus")
    pdf_of_arc = tm.id2pdf[f.arcs["ilabel"]].astype(np.int32)
    pdf_class = np.full(tm.num_pdfs, 5, np.int32)
    _check_graph(rng, f, pdf_of_arc, pdf_class, span=32, trials=60)
    _check_graph(rng, f, pdf_of_arc, np.zeros(tm.num_pdfs, np.int32), span=32, trials=60, groups=8)   # all single-block pdfs, 8 runs
    # one column per pdf without clustering; more columns, each of bounded depth extent, with it
    sd, col, cp0, cf0, cl0, cc0, _g = _plan(f, pdf_of_arc, pdf_class, 0)
    assert len(cp0) == len(np.unique(pdf_of_arc))
    sd, col, cp, cf, cl, cc, _g = _plan(f, pdf_of_arc, pdf_class, 32)
    assert len(cp) > len(cp0)                       # the text repeats its words: repeated pdfs got their own columns
    src = np.repeat(np.arange(f.num_states), np.diff(f.arc_offsets))
    ext_lo = np.full(len(cp), 1 << 30); ext_hi = np.zeros(len(cp), np.int64)
    np.minimum.at(ext_lo, col, sd[src, 0]); np.maximum.at(ext_hi, col, sd[src, 0])
    assert np.all(ext_hi - ext_lo <= 32)


def test_batched_plans_equal_the_per_utterance_call(fx):
    """mfa_build_score_plans_batch (a whole batch over host threads — what engine.pack_graphs calls) writes, per utterance,
    exactly what mfa_build_score_plan_grouped writes; a bad utterance is reported by index with the per-utterance code."""
    rng = np.random.default_rng(9)
    tm = fx.mono_tm
    lib = _lib.lib()
    pdf_class = rng.integers(0, 6, size=tm.num_pdfs).astype(np.int32)
    fsts = [_random_graph(rng, tm, int(rng.choice([3, 8, 40, 150]))) for _ in range(23)]
    for groups, span, threads in ((1, 0, 1), (8, 32, 4), (16, 8, 7)):
        S = np.array([f.num_states for f in fsts]); A = np.array([f.num_arcs for f in fsts])
        state_off = np.concatenate([[0], np.cumsum(S)]).astype(np.int64)
        arc_base = np.concatenate([[0], np.cumsum(A)]).astype(np.int64)
        arc_off = np.concatenate([f.arc_offsets.astype(np.int32) for f in fsts])
        nxt = np.concatenate([f.arcs["nextstate"] for f in fsts]).astype(np.int32)
        pdf = np.concatenate([tm.id2pdf[f.arcs["ilabel"]] for f in fsts]).astype(np.int32)
        starts = np.array([f.start for f in fsts], dtype=np.int32)
        sd = np.full((int(S.sum()), 2), -7, np.int32)
        col, cp, cf, cl = (np.full(int(A.sum()), -7, np.int32) for _ in range(4))
        cc = np.zeros((len(fsts), 6), np.int32); gc = np.zeros((len(fsts), groups), np.int32)
        ncol = np.zeros(len(fsts), np.int32); bad = C.c_int32(-1)
        rc = lib.mfa_build_score_plans_batch(len(fsts), state_off.ctypes.data, arc_base.ctypes.data, arc_off.ctypes.data,
                                             nxt.ctypes.data, pdf.ctypes.data, starts.ctypes.data, len(pdf_class),
                                             pdf_class.ctypes.data, span, groups, threads, sd.ctypes.data, col.ctypes.data,
                                             cp.ctypes.data, cf.ctypes.data, cl.ctypes.data, cc.ctypes.data, gc.ctypes.data,
                                             ncol.ctypes.data, C.byref(bad))
        assert rc == 0
        for u, f in enumerate(fsts):
            sd1, col1, cp1, cf1, cl1, cc1, gc1 = _plan(f, tm.id2pdf[f.arcs["ilabel"]], pdf_class, span, groups)
            s0, a0, k = int(state_off[u]), int(arc_base[u]), int(ncol[u])
            assert k == len(cp1)
            assert np.array_equal(sd[s0: s0 + f.num_states], sd1) and np.array_equal(col[a0: a0 + f.num_arcs], col1)
            assert np.array_equal(cp[a0: a0 + k], cp1) and np.array_equal(cf[a0: a0 + k], cf1) and np.array_equal(cl[a0: a0 + k], cl1)
            assert np.array_equal(cc[u], cc1)
            if groups > 1:
                assert np.array_equal(gc[u], gc1)
    pdf_bad = pdf.copy()
    pdf_bad[int(arc_base[5])] = tm.num_pdfs + 3          # utterance 5 names a pdf the model does not have
    rc = lib.mfa_build_score_plans_batch(len(fsts), state_off.ctypes.data, arc_base.ctypes.data, arc_off.ctypes.data,
                                         nxt.ctypes.data, pdf_bad.ctypes.data, starts.ctypes.data, len(pdf_class),
                                         pdf_class.ctypes.data, span, groups, 3, sd.ctypes.data, col.ctypes.data,
                                         cp.ctypes.data, cf.ctypes.data, cl.ctypes.data, cc.ctypes.data, gc.ctypes.data,
                                         ncol.ctypes.data, C.byref(bad))
    assert rc == -2 and bad.value == 5
