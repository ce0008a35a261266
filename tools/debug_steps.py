#!/usr/bin/env python3
"""Step-by-step replay of tests/test_gpu_lazy.py::test_lazy_single_gaussian_model_and_real_audio with a device
synchronisation and a progress line after every stage (debugging aid)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from montreal_forced_aligner_amd.engine import AlignmentEngine  # noqa: E402
from tests import helpers  # noqa: E402


def say(*a):
    torch.cuda.synchronize()
    time.sleep(0.3)
    print("[step]", *a, flush=True)


def main():
    fx = helpers.Fixtures()
    e = AlignmentEngine(0)
    say("engine")
    tm, am = fx.mono_tm, fx.mono_am
    sr = 16000
    cuts = [(0.0, 4.2), (4.0, 6.5), (0.0, 26.72), (23.5, 26.72)]
    texts = ["this is the acoustic corpus i'm talking pretty fast here", "there's nothing going else going on", fx.text,
             "um and that should be all thanks"]
    segs = [fx.pcm[int(a * sr): int(b * sr)] for a, b in cuts]
    e.configure_mfcc()
    e.load_gmm(am)
    say("model loaded")
    so = np.concatenate([[0], np.cumsum([len(s) for s in segs])]).astype(np.int64)
    mfcc, fo = e.mfcc(torch.from_numpy(np.concatenate(segs)).to(e.device), so)
    say("mfcc", fo)
    u2s = np.arange(len(segs), dtype=np.int32)
    st = e.cmvn_stats(mfcc, fo, u2s, len(segs))
    say("cmvn")
    feats = e.features(mfcc, fo, u2s, st)
    say("feats", tuple(feats.shape))
    fsts = [fx.mono_graph(t) for t in texts]
    graphs = e.pack_graphs(fsts, tm)
    say("packed", np.diff(graphs.pdf_off_host), graphs.class_counts.cpu().numpy().tolist())
    for beam, retry in ((100.0, 400.0), (10.0, 40.0)):
        ll, ll_off, ll_cols = e.score(feats, fo, graphs.pdf_list, graphs.pdf_off_host, graphs.class_counts)
        say("dense score")
        dense = e.align(graphs, ll, ll_off, ll_cols, fo, want_frame_likes=True, beam=beam, retry_beam=retry, max_tokens=2048,
                        bp_tokens_per_frame=1024)
        say("dense align", dense["status"].cpu().numpy())
        lazy = e.align_features(graphs, feats, fo, want_frame_likes=True, beam=beam, retry_beam=retry, max_tokens=2048,
                                bp_tokens_per_frame=1024)
        say("lazy align", lazy["status"].cpu().numpy())
        for k in ("status", "ali", "words", "n_words", "like", "frame_like"):
            print(k, bool(torch.equal(dense[k], lazy[k])), flush=True)
        d, s = ll.cpu().numpy(), lazy["loglikes"].cpu().numpy()
        w = s != 0
        print("cells equal", bool(np.array_equal(d[w], s[w])), "fill", float(w.mean()), flush=True)
    e.close()


if __name__ == "__main__":
    main()
