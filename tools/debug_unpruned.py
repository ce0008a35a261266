#!/usr/bin/env python3
"""Debug aid: beam 1e4 (no pruning) through the dense and the lazy path on a few headline-shape utterances."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import synth_workload as synth  # noqa: E402
from montreal_forced_aligner_amd import graph as G  # noqa: E402
from montreal_forced_aligner_amd.engine import AlignmentEngine  # noqa: E402

e = AlignmentEngine(0)
e.configure_mfcc()
world = synth.SynthWorld.build()
lda = synth.seeded_lda(); fm = synth.seeded_fmllr(16); d_lda = torch.from_numpy(lda).to(e.device)


def feats_of(pcm_list, spks):
    so = np.concatenate([[0], np.cumsum([len(p) for p in pcm_list])]).astype(np.int64)
    mfcc, fo = e.mfcc(torch.from_numpy(np.concatenate(pcm_list)).to(e.device), so)
    own = np.arange(len(pcm_list), dtype=np.int32)
    st = e.cmvn_stats(mfcc, fo, own, len(pcm_list))
    return e.features(mfcc, fo, own, st, lda=d_lda, fmllr=torch.from_numpy(fm[np.asarray(spks) % 16]).to(e.device)), fo


model = synth.train_triphone(world, lambda pcm, spk: feats_of([pcm], [spk])[0].cpu().numpy(), n_train=40, n_gauss=32, n_classes=2)
e.load_gmm(model.am)
utts = [world.utterance(7500 + i) for i in range(4)]
gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
fsts = [G.add_transition_probs(gc.compile_fst(u[1]), model.tm.scaled_log_probs(1.0, 0.1)) for u in utts]
feats, fo = feats_of([u[0] for u in utts], [u[3] for u in utts])
g = e.pack_graphs(fsts, model.tm)
print("max_states", g.max_states, "max_arcs", g.max_arcs, "S", [f.num_states for f in fsts], "A", [f.num_arcs for f in fsts], flush=True)
ll, ll_off, ll_cols = e.score(feats, fo, g.pdf_list, g.pdf_off_host, g.class_counts)
for mt in (g.max_states, 4096):
    d = e.align(g, ll, ll_off, ll_cols, fo, beam=1.0e4, retry_beam=0.0, max_tokens=mt, bp_tokens_per_frame=g.max_states)
    print("dense max_tokens", mt, d["status"].cpu().tolist(), flush=True)
    z = e.align_features(g, feats, fo, beam=1.0e4, retry_beam=0.0, max_tokens=mt, bp_tokens_per_frame=g.max_states)
    print("lazy  max_tokens", mt, z["status"].cpu().tolist(), flush=True)
e.close()
