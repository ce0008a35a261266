"""Host-side rates around the device path for configs[2]-shaped utterances: transcript → training graph (native batch
compiler vs graph.py), graph → device layout + score plan (engine.pack_graphs), alignment → phone/word intervals
(ctm).  GPU box (pack_graphs uploads its tables):  python tools/host_pipeline_rate.py [n_utt]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import synth_workload as synth                                          # noqa: E402
from montreal_forced_aligner_amd import ctm as CTM                      # noqa: E402
from montreal_forced_aligner_amd import graph as G                      # noqa: E402
from montreal_forced_aligner_amd.engine import AlignmentEngine          # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    world = synth.SynthWorld.build()
    rng = np.random.default_rng(0)
    model = synth.train_triphone(world, lambda pcm, spk: rng.normal(size=(len(pcm) // 160, 40)).astype(np.float32),
                                 n_train=12, n_gauss=32)
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    texts = [world.utterance(500 + i, n_words=30)[1] for i in range(n)]
    t0 = time.time()
    ref = [G.add_transition_probs(gc.compile_fst(t), scaled) for t in texts[:64]]
    t_py = (time.time() - t0) / 64
    gc.compile_fsts(texts, scaled)                                       # registers the context windows
    t0 = time.time()
    fsts = gc.compile_fsts(texts, scaled)
    t_nat = (time.time() - t0) / n
    print(f"graph compile: graph.py {1e3 * t_py:.2f} ms/utterance/core; native {1e3 * t_nat:.3f} ms/utterance "
          f"({gc._native.n_threads} threads) = {1 / t_nat:.0f} utterances/s; states {fsts[0].num_states}", flush=True)
    eng = AlignmentEngine(0)
    eng.load_gmm(model.am)
    eng.pack_graphs(fsts[:64], model.tm)
    t0 = time.time()
    eng.pack_graphs(fsts, model.tm)
    t_pack = (time.time() - t0) / n
    print(f"pack_graphs (depths, score columns, upload): {1e3 * t_pack:.3f} ms/utterance = {1 / t_pack:.0f} utterances/s", flush=True)
    pt = world.lexicon.phone_table
    # intervals: real alignments (of noise features: valid transition-id sequences is all that matters here)
    import torch
    m = 256
    fo = np.arange(m + 1, dtype=np.int64) * 1000
    feats = torch.from_numpy(rng.normal(size=(m * 1000, model.am.dim)).astype(np.float32)).to(eng.device)
    res = eng.align_features(eng.pack_graphs(fsts[:m], model.tm), feats, fo, beam=200.0, retry_beam=800.0, max_tokens=2048,
                             bp_tokens_per_frame=1024)
    st, ali = res["status"].cpu().numpy(), res["ali"].cpu().numpy()
    alis = [ali[k * 1000: (k + 1) * 1000] for k in range(m) if st[k] in (0, 1)]
    words = [res["words"].cpu().numpy()[k * 1000: k * 1000 + int(res["n_words"][k])] for k in range(m) if st[k] in (0, 1)]
    print(f"{len(alis)} of {m} noise utterances aligned", flush=True)
    t0 = time.time()
    for a in alis:
        CTM.generate_ctm(a, model.tm, pt, 0.01)
    t_ctm = (time.time() - t0) / len(alis)
    print(f"generate_ctm (SplitToPhones): {1e3 * t_ctm:.3f} ms/utterance/core", flush=True)
    t0 = time.time()
    for a, w, t in zip(alis, words, texts):
        iv = CTM.generate_ctm(a, model.tm, pt, 0.01)
        CTM.phones_to_pronunciations(world.lexicon, w, iv, transcription=False, text=t)
    t_w = (time.time() - t0) / len(alis)
    print(f"generate_ctm + phones_to_pronunciations: {1e3 * t_w:.3f} ms/utterance/core", flush=True)


if __name__ == "__main__":
    main()
