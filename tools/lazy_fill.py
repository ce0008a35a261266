#!/usr/bin/env python3
"""Diagnostic (GPU): how much of the score matrix does the lazy path write, per window size and column clustering gap,
on the bench's own workload (BASELINE configs[2] model, full pdf inventory)?  Prints one line per setting."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import synth_workload as synth  # noqa: E402
from montreal_forced_aligner_amd import graph as G  # noqa: E402
from montreal_forced_aligner_amd.engine import AlignmentEngine  # noqa: E402


def main(n_utt=48):
    eng = AlignmentEngine(0)
    eng.configure_mfcc()
    dev = eng.device
    world = synth.SynthWorld.build()
    lda = synth.seeded_lda()
    fm = synth.seeded_fmllr(1000)
    d_lda = torch.from_numpy(lda).to(dev)

    def feats_of(pcm_list, spks):
        so = np.concatenate([[0], np.cumsum([len(p) for p in pcm_list])]).astype(np.int64)
        mfcc, fo = eng.mfcc(torch.from_numpy(np.concatenate(pcm_list)).to(dev), so)
        own = np.arange(len(pcm_list), dtype=np.int32)
        stats = eng.cmvn_stats(mfcc, fo, own, len(pcm_list))
        per = torch.from_numpy(fm[np.asarray(spks) % 1000]).to(dev)
        return eng.features(mfcc, fo, own, stats, lda=d_lda, fmllr=per), fo

    model = synth.train_triphone(world, lambda pcm, spk: feats_of([pcm], [spk])[0].cpu().numpy(), n_train=120)
    eng.load_gmm(model.am)
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    utts = [world.utterance(5000 + i) for i in range(n_utt)]
    fsts = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    feats, fo = feats_of([u[0] for u in utts], [u[3] for u in utts])
    for gap in (64, 32, 16, 8):
        graphs = eng.pack_graphs(fsts, model.tm, cluster_gap=gap)
        P = np.diff(graphs.pdf_off_host)
        sd = graphs.state_depth.cpu().numpy()
        for window in (64, 128):
            t0 = time.time()
            r = eng.align_features(graphs, feats, fo, beam=10.0, retry_beam=40.0, max_tokens=256, bp_tokens_per_frame=128,
                                   window=window)
            torch.cuda.synchronize()
            ll = r["loglikes"].cpu().numpy()
            st = r["status"].cpu().numpy()
            fill = float((ll != 0).mean())
            # per-window band width (columns written in the window's first frame), utterance 0
            T0, P0 = int(fo[1] - fo[0]), int(P[0])
            m0 = ll[: T0 * P0].reshape(T0, P0) != 0
            widths = [int(m0[t].sum()) for t in range(0, T0, window)]
            print(f"gap {gap} window {window}: fill {fill:.3f}; columns/utt {P.mean():.0f} (max {P.max()}); "
                  f"max BFS depth {sd[:, 0].max()}, max longest depth {sd[:, 1].max()}; ok {(st == 0).sum()}/{len(st)}; "
                  f"utt0 band widths {widths}; {time.time() - t0:.2f}s", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
