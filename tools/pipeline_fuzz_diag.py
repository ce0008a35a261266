"""Diagnosis of one tools/pipeline_fuzz.py mismatch: is the difference the decoder's (the oracle fed the DEVICE's features must
then disagree with the device too) or the features' (a near-tie decided differently by features 1e-3 apart)?
python tools/pipeline_fuzz_diag.py SEED UTT"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import synth_workload as synth
from montreal_forced_aligner_amd import graph as G
from montreal_forced_aligner_amd.engine import AlignmentEngine
from oracle import oracle as O
from tests import helpers

seed, k = int(sys.argv[1]), int(sys.argv[2])
eng = AlignmentEngine(0)
eng.configure_mfcc()
world = synth.SynthWorld.build()
lda = synth.seeded_lda()
fm = synth.seeded_fmllr(16)
d_lda = torch.from_numpy(lda).to(eng.device)


def feats_of(pcm, spk):
    so = np.array([0, len(pcm)], dtype=np.int64)
    mfcc, fo = eng.mfcc(torch.from_numpy(pcm).to(eng.device), so)
    own = np.zeros(1, dtype=np.int32)
    return eng.features(mfcc, fo, own, eng.cmvn_stats(mfcc, fo, own, 1), lda=d_lda,
                        fmllr=torch.from_numpy(fm[[spk % 16]]).to(eng.device)).cpu().numpy()


model = synth.train_triphone(world, feats_of, n_train=40, n_gauss=32, n_classes=2)
gc = G.TrainingGraphCompiler(model.tm, model.tree, world.lexicon)
scaled = model.tm.scaled_log_probs(1.0, 0.1)
rng = np.random.default_rng(52000 + seed)
n_utt = int(rng.integers(2, 40)); n_spk = int(rng.integers(1, 5))
raw = []
for i in range(n_utt):
    nw = int(rng.choice([1, 2, 3, 5, 8, 15, 30, 60]))
    ns = int(nw * rng.integers(4000, 9000)) + int(rng.integers(0, 160))
    if rng.random() < 0.1:
        ns = int(rng.integers(500, 4000))
    spk = int(rng.integers(0, n_spk))
    pcm, text, _segs, _ = world.utterance(60000 + 100 * seed + i, n_words=nw, samples=ns, speaker=spk)
    raw.append((pcm, text, spk))
pcm, text, spk = raw[k]
mates = [j for j in range(n_utt) if raw[j][2] == spk]
# device features with the speaker's pooled CMVN
so = np.concatenate([[0], np.cumsum([len(raw[j][0]) for j in mates])]).astype(np.int64)
mfcc, fo = eng.mfcc(torch.from_numpy(np.concatenate([raw[j][0] for j in mates])).to(eng.device), so)
rows = np.zeros(len(mates), dtype=np.int32)
stats = eng.cmvn_stats(mfcc, fo, rows, 1)
feats = eng.features(mfcc, fo, rows, stats, lda=d_lda, fmllr=torch.from_numpy(fm[[spk % 16]]).to(eng.device))
m = mates.index(k)
x_dev = feats[int(fo[m]): int(fo[m + 1])].cpu().numpy()
mf = [O.mfcc(raw[j][0].astype(np.float32), O.default_mfcc_opts()) for j in mates]
cm = O.cmvn_stats([a for a in mf if a.shape[0] > 0])
x_orc = O.affine(O.affine(O.splice(O.cmvn_apply(cm, mf[m])), lda), fm[spk % 16])
print("feature difference device vs oracle: max abs", float(np.abs(x_dev - x_orc).max()), "on values of", float(np.abs(x_orc).max()))
fst = G.add_transition_probs(gc.compile_fst(text), scaled)
ref_o = helpers.oracle_align_feats(model.tm, fst, x_orc, model.am, beam=10.0, retry_beam=40.0)
ref_d = helpers.oracle_align_feats(model.tm, fst, x_dev, model.am, beam=10.0, retry_beam=40.0)
eng.load_gmm(model.am)
g = eng.pack_graphs([fst], model.tm)
fo1 = np.array([0, x_dev.shape[0]], dtype=np.int64)
r = eng.align_features(g, torch.from_numpy(x_dev).to(eng.device), fo1, beam=10.0, retry_beam=40.0)
ali = r["ali"].cpu().numpy()
print("oracle(oracle features) vs oracle(device features): frames differing", int((ref_o["ali"] != ref_d["ali"]).sum()),
      "likes", ref_o["like"], ref_d["like"])
print("device vs oracle(device features): frames differing", int((ali != ref_d["ali"]).sum()), "status", int(r["status"].cpu()[0]), ref_d["status"],
      "like", float(r["like"].cpu()[0]), ref_d["like"])
bad = np.flatnonzero(ref_o["ali"] != ref_d["ali"])
print("frames", bad.tolist()[:10])
