"""Randomised twin test of the native training-graph compiler (libmfa_graph.so) against graph.py: random transcripts (with
out-of-vocabulary words, repeated words, single words, empty text) over the synthetic triphone model and the reference's
monophone fixture; state numbers, arc order and float32 weights must be identical.  CPU only.  python tools/graph_fuzz.py [n] [seed]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np                                                       # noqa: E402

import synth_workload as synth                                           # noqa: E402
from montreal_forced_aligner_amd import graph as G, graph_native as GN   # noqa: E402
from tests import helpers                                                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(77000 + seed)


def same(a, b):
    return (a.start == b.start and np.array_equal(a.arc_offsets, b.arc_offsets) and np.array_equal(a.arcs, b.arcs)
            and np.array_equal(a.final, b.final))


bad = 0
world = synth.SynthWorld.build()
model = synth.train_triphone(world, lambda pcm, spk: rng.normal(size=(len(pcm) // 160, 40)).astype(np.float32), n_train=12, n_gauss=1)
fx = helpers.Fixtures()
for name, gc, words, scaled in (("triphone", G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon), list(world.lexicon._by_word.keys()),
                                 model.tm.scaled_log_probs(1.0, 0.1)),
                                ("triphone-undeterminized", G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon, determinize=False),
                                 list(world.lexicon._by_word.keys()), model.tm.scaled_log_probs(1.0, 0.1)),
                                ("mono fixture", fx.mono_gc, list(fx.mono_lex._by_word.keys()), fx.mono_tm.scaled_log_probs(1.0, 0.1))):
    texts = []
    for _ in range(n):
        k = int(rng.choice([0, 1, 1, 2, 3, 5, 8, 13, 21, 34]))
        ws = [str(rng.choice(words)) if rng.random() > 0.1 else "zzqx" for _ in range(k)]
        if k > 2 and rng.random() < 0.3:
            ws[1] = ws[0]
        texts.append(" ".join(ws))
    nat = GN.NativeGraphCompiler(gc, n_threads=int(rng.choice([1, 3, 8])))
    t0 = time.time()
    got = nat.compile_batch(texts, scaled)
    t_nat = time.time() - t0
    t0 = time.time()
    n_bad = 0
    for t, g in zip(texts, got):
        ref = G.add_transition_probs(gc.compile_fst(t), scaled)
        if not same(g, ref):
            n_bad += 1
            print("MISMATCH", name, repr(t)[:120], flush=True)
    print(f"{name}: {len(texts)} transcripts, {n_bad} mismatches; native {t_nat:.2f} s, graph.py {time.time() - t0:.1f} s", flush=True)
    bad += n_bad
print("mismatches:", bad)
sys.exit(1 if bad else 0)
