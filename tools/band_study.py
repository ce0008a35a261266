"""Design study of the lazy-scoring band (CPU only: the oracle's decoder with per-frame token depths + the host's score
plan): for configs[2]-shaped utterances, how many (window, column) blocks and cells would a band policy score, against
what the decoder asks for?  python tools/band_study.py [n_utt]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import synth_workload as S                                  # noqa: E402
from montreal_forced_aligner_amd import graph as G          # noqa: E402
from oracle import oracle as O                              # noqa: E402
from tests.test_score_plan_cpu import _plan                 # noqa: E402


def main():
    n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    w = S.SynthWorld.build()
    lda, fm = S.seeded_lda(), S.seeded_fmllr(8)

    def ff(pcm, spk):
        mf = O.mfcc(pcm.astype(np.float32), O.default_mfcc_opts())
        return O.affine(O.affine(O.splice(O.cmvn_apply(O.cmvn_stats([mf]), mf)), lda), fm[spk % 8])

    m = S.train_triphone(w, ff, n_train=60)
    gc = G.TrainingGraphCompiler(m.tm, m.tree, w.lexicon)
    sc = m.tm.scaled_log_probs(1.0, 0.1)
    pdf_class = np.zeros(m.am.num_pdfs, np.int32)
    rows = []
    for i in range(n_utt):
        pcm, text, segs, spk = w.utterance(i)
        f = G.add_transition_probs(gc.compile_fst(text), sc)
        pdf_of_arc = m.tm.id2pdf[f.arcs["ilabel"]].astype(np.int32)
        for span in (32, 8):
            sd, col, cp, cf, cl, cc, gcnt = _plan(f, pdf_of_arc, pdf_class, span)
            # per column: [first depth, last depth (own, not the running max)]
            src = np.repeat(np.arange(f.num_states), np.diff(f.arc_offsets))
            own_last = np.zeros(len(cp), np.int64)
            np.maximum.at(own_last, col, sd[src, 0])
            r = O.align_feats(f.num_states, f.start, f.arc_offsets, f.arcs, f.final, ff(pcm, spk), m.am.gconsts, m.am.means_invvars,
                              m.am.inv_vars, m.am.pdf_offsets, m.tm.id2pdf, 0.1, 10.0, 40.0, state_depth=sd)
            if r["status"] != 0:
                continue
            fb = r["frame_band"]
            T = fb.shape[0]
            for K in (64, 128):
                for policy in ("proven", "look48", "look32", "adaptive", "oracle_reach"):
                    blocks = 0
                    fails = 0
                    prev_adv = None
                    for t0 in range(0, T, K):
                        lo, dmax = int(fb[t0, 0]), int(fb[t0, 1])
                        t1 = min(T, t0 + K)
                        reach_hi = int(fb[t0:t1, 1].max())          # what the window's frames actually reach (max BFS depth of live tokens)
                        adv = reach_hi - dmax
                        if policy == "proven":
                            look = K - 1
                        elif policy == "look48":
                            look = K * 3 // 4
                        elif policy == "look32":
                            look = K // 2
                        elif policy == "adaptive":
                            look = K * 3 // 4 if prev_adv is None else min(K - 1, max(12, int(prev_adv * 1.5) + 6))
                        else:
                            look = adv + 1
                        hi = dmax + look
                        if reach_hi + 1 > hi:                      # a token reads an arc leaving a state deeper than the band: redo
                            fails += 1
                            hi = dmax + K - 1
                            blocks += int(((cl >= lo) & (cf <= dmax + look)).sum())      # the failed attempt was scored too
                        blocks += int(((cl >= lo) & (cf <= hi)).sum())
                        prev_adv = adv
                    n_win = (T + K - 1) // K
                    rows.append((span, K, policy, blocks / n_win, blocks * K / (T * len(cp)), fails / n_win, len(cp), r["cells"] / T))
    import collections
    agg = collections.defaultdict(list)
    for span, K, policy, bpw, frac, fail, ncols, cpf in rows:
        agg[(span, K, policy)].append((bpw, frac, fail, ncols, cpf))
    print("span  K   policy        blocks/window  cells-frac  fail/window  cols/utt  kaldi-cells/frame   gathers per 64 frames")
    for key in sorted(agg):
        a = np.array(agg[key]).mean(axis=0)
        print(f"{key[0]:4d} {key[1]:4d} {key[2]:13s} {a[0]:10.1f} {a[1]:12.3f} {a[2]:10.3f} {a[3]:10.0f} {a[4]:10.1f} {a[0] * 64 / key[1]:14.1f}")
    # token spread
    print("done")


if __name__ == "__main__":
    main()
