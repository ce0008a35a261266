"""Pure-numpy emulation of the device Viterbi kernel's PARALLEL formulation (montreal_forced_aligner_amd/csrc/viterbi.hip):
prefix-min cutoffs, per-state winners, Kaldi list order via first-creator ordinals and hash buckets.  Lets the
formulation be checked against the sequential oracle on a machine without a GPU.  Test infrastructure only."""
import numpy as np

MIN_ACTIVE, BEAM_DELTA, HASH_RATIO = 20, np.float32(0.5), np.float32(2.0)


def decode(S, start, arc_off, arcs, final, ll, tid2col, scale, beam, H0=1000):
    """Returns dict(status, ali, words, like, H).  One beam, no retry."""
    T = ll.shape[0]
    scale = np.float32(scale)
    beam = np.float32(beam)
    l_state = np.array([start], dtype=np.int64)
    l_cost = np.array([0.0], dtype=np.float64)
    H = H0
    bp = []  # per frame: (arc index, prev pos) arrays
    a_w = arcs["weight"].astype(np.float32)
    a_next = arcs["nextstate"].astype(np.int64)
    a_col = tid2col[arcs["ilabel"]]
    for t in range(T):
        n = l_state.shape[0]
        if n == 0:
            return dict(status=2, H=H)
        best_i = int(np.argmin(l_cost))  # first index of the minimum
        best = l_cost[best_i]
        if n <= MIN_ACTIVE:
            wcut, abeam = np.inf, np.float32(np.inf)
        else:
            beam_cut = best + np.float64(beam)
            kle = int((l_cost <= beam_cut).sum())
            if kle > MIN_ACTIVE:
                wcut, abeam = beam_cut, beam
            else:
                v = np.sort(l_cost)[MIN_ACTIVE]
                wcut = v
                abeam = np.float32(v - best + np.float64(BEAM_DELTA))
        want = int(np.float32(n) * HASH_RATIO)
        if want > H:
            H = want

        def cand(i, a):
            ac = np.float32(-(scale * ll[t, a_col[a]]))
            return (np.float64(a_w[a]) + l_cost[i]) + np.float64(ac)

        run = np.inf
        bs = l_state[best_i]
        for a in range(arc_off[bs], arc_off[bs + 1]):
            run = min(run, cand(best_i, a))
        # candidates in list x arc order
        created = []  # (cidx_pos, k, dest, nw, ordinal)
        ordinal = 0
        cbase = np.zeros(n, dtype=np.int64)
        local = run
        for i in range(n):
            cbase[i] = ordinal
            if not (l_cost[i] < wcut):
                continue
            st = l_state[i]
            for k, a in enumerate(range(arc_off[st], arc_off[st + 1])):
                nw = cand(i, a)
                if nw < local + np.float64(abeam):
                    created.append((i, k, int(a_next[a]), nw, ordinal + k, a))
                local = min(local, nw)
            ordinal += arc_off[st + 1] - arc_off[st]
        if not created:
            return dict(status=2, H=H)
        # per destination: best cost, first creator F, winner W
        slots = {}
        for (i, k, d, nw, o, a) in created:
            s = slots.setdefault(d, dict(cost=np.inf, F=None, W=None))
            if s["F"] is None or (i, k) < s["F"][:2]:
                s["F"] = (i, k, o)
            s["cost"] = min(s["cost"], nw)
        for (i, k, d, nw, o, a) in created:
            s = slots[d]
            if nw == s["cost"] and (s["W"] is None or (i, k) < s["W"][:2]):
                s["W"] = (i, k, a)
        # order: (first-creator ordinal of the bucket, first-creator ordinal of the state)
        keys = []
        for d, s in slots.items():
            if S > H:
                Fb = min(slots[m]["F"][2] for m in range(d % H, S, H) if m in slots)
            else:
                Fb = s["F"][2]
            keys.append((Fb, s["F"][2], d))
        keys.sort()
        new_state = np.array([d for _, _, d in keys], dtype=np.int64)
        new_cost = np.array([slots[d]["cost"] for d in new_state], dtype=np.float64)
        bp.append((np.array([slots[d]["W"][2] for d in new_state]), np.array([slots[d]["W"][0] for d in new_state])))
        l_state, l_cost = new_state, new_cost
    fw = final[l_state].astype(np.float64)
    tot = np.where(np.isinf(fw), np.inf, l_cost + fw)
    if not np.isfinite(tot).any():
        return dict(status=2, H=H)
    pos = int(np.argmin(tot))
    fstate = l_state[pos]
    path = np.zeros(T, dtype=np.int64)
    for t in range(T - 1, -1, -1):
        path[t] = bp[t][0][pos]
        pos = int(bp[t][1][pos])
    ali = arcs["ilabel"][path].astype(np.int32)
    words = arcs["olabel"][path]
    words = words[words != 0].astype(np.int32)
    cost, w1, w2 = 0.0, np.float32(0), np.float32(0)
    for t in range(T):
        a = path[t]
        ac = np.float32(-(scale * ll[t, a_col[a]]))
        nc = (np.float64(a_w[a]) + cost) + np.float64(ac)
        tot_c = np.float32(nc - cost)
        w1 = np.float32(w1 + a_w[a])
        w2 = np.float32(w2 + np.float32(tot_c - a_w[a]))
        cost = nc
    w1 = np.float32(w1 + final[fstate])
    like = np.float32(-np.float32(w1 + w2) / scale)
    return dict(status=0, ali=ali, words=words, like=float(like), H=H)
