"""Rate of the general-graph decoder (mfa_align_general_batch: ε input arcs, one thread per utterance) on configs[2]-shaped
utterances whose training graphs were rewritten with ε arcs on a third of their arcs — next to the fast path on the
equivalent ε-free graphs.  GPU box only:  python tools/general_rate.py [n_utt] [eps_fraction]
(eps_fraction: share of the arcs split by an ε arc, default 0.3 — far more than a compiled training graph carries.)
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from montreal_forced_aligner_amd import graph as G                      # noqa: E402
from montreal_forced_aligner_amd.engine import AlignmentEngine          # noqa: E402
from tests import synth                                                 # noqa: E402
from tests.test_gpu_general import _with_eps                            # noqa: E402


def main():
    n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    frac = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else 0.3
    engine = AlignmentEngine(0)
    world = synth.SynthWorld.build()
    engine.configure_mfcc()
    lda = synth.seeded_lda()
    fm = synth.seeded_fmllr(16)
    d_lda = torch.from_numpy(lda).to(engine.device)

    def feats_of(pcm_list, spks):
        so = np.concatenate([[0], np.cumsum([len(p) for p in pcm_list])]).astype(np.int64)
        mfcc, fo = engine.mfcc(torch.from_numpy(np.concatenate(pcm_list)).to(engine.device), so)
        own = np.arange(len(pcm_list), dtype=np.int32)
        stats = engine.cmvn_stats(mfcc, fo, own, len(pcm_list))
        per_utt = torch.from_numpy(fm[np.asarray(spks) % 16]).to(engine.device)
        return engine.features(mfcc, fo, own, stats, lda=d_lda, fmllr=per_utt), fo

    model = synth.train_triphone(world, lambda pcm, spk: feats_of([pcm], [spk])[0].cpu().numpy(), n_train=40, n_gauss=32,
                                 n_classes=2)
    engine.load_gmm(model.am)
    pool = 64                                                           # distinct utterances, repeated to n_utt
    utts = [world.utterance(9000 + i, n_words=30, samples=160000) for i in range(pool)]
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    plain = [G.add_transition_probs(gc.compile_fst(u[1]), scaled) for u in utts]
    rng = np.random.default_rng(5)
    eps = [_with_eps(rng, f, frac) for f in plain]
    if "--no-start-eps" in sys.argv:      # (measurement aid: graphs whose start state has no epsilon arc)
        def start_has_eps(f):
            a0, a1 = int(f.arc_offsets[f.start]), int(f.arc_offsets[f.start + 1])
            return bool((f.arcs["ilabel"][a0:a1] == 0).any())
        for k in range(len(eps)):
            while start_has_eps(eps[k]):
                eps[k] = _with_eps(rng, plain[k], frac)
    print(f"epsilon arcs: {sum(int((f.arcs['ilabel'] == 0).sum()) for f in eps)} of {sum(f.num_arcs for f in eps)} arcs", flush=True)
    rep = (n_utt + pool - 1) // pool
    pcm = [u[0] for u in utts] * rep
    spk = [u[3] for u in utts] * rep
    feats, fo = feats_of(pcm[:n_utt], spk[:n_utt])
    g_fast = engine.pack_graphs((plain * rep)[:n_utt], model.tm)
    g_gen = engine.pack_graphs_general((eps * rep)[:n_utt], model.tm)
    g_eps = engine.pack_graphs((eps * rep)[:n_utt], model.tm)            # round 3: ε graphs on the wavefront-parallel decoder
    kw = dict(beam=10.0, retry_beam=40.0)
    out = {}
    for name, fn in (("fast path, ε-free graphs", lambda: engine.align_features(g_fast, feats, fo, max_tokens=1024,
                                                                                 bp_tokens_per_frame=256, **kw)),
                     ("wavefront decoder, ε graphs", lambda: engine.align_features(g_eps, feats, fo, max_tokens=1024,
                                                                                   bp_tokens_per_frame=256, **kw)),
                     ("general decoder, ε graphs", lambda: engine.align_general(g_gen, feats, fo, **kw))):
        r = fn()
        torch.cuda.synchronize()
        t0 = time.time()
        n = 3
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        st = r["status"].cpu().numpy()
        out[name] = r
        print(f"{name}: {n_utt} utterances in {dt * 1e3:.1f} ms = {n_utt / dt:.0f} utterances/s; "
              f"aligned {(st <= 1).mean():.3f}", flush=True)
    a, b, c = out["fast path, ε-free graphs"], out["general decoder, ε graphs"], out["wavefront decoder, ε graphs"]
    print("same likelihoods (|Δ| per frame):", float((a["like"] - b["like"]).abs().max()))
    print("wavefront ε decoder == general decoder:", bool(torch.equal(b["ali"], c["ali"]) and torch.equal(b["like"], c["like"])
                                                          and torch.equal(b["status"], c["status"])))


if __name__ == "__main__":
    main()
