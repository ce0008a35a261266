"""Randomised end-to-end check of the corpus driver against the oracle from PCM: random numbers of utterances, lengths from a
fraction of a second to 25 s, a few speakers (per-speaker CMVN pooled over the speaker's utterances), LDA + per-speaker fMLLR,
small `batch_frames` (several length-bucketed batches, the software pipeline, the staging pools), beam 10 / 40, through
CorpusAligner.align; every utterance's alignment, words and likelihood against the oracle's whole path.  GPU.
python tools/pipeline_fuzz.py [n_seeds] [first_seed]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np                                                       # noqa: E402
import torch                                                             # noqa: E402

import synth_workload as synth                                           # noqa: E402
from montreal_forced_aligner_amd import graph as G                       # noqa: E402
from montreal_forced_aligner_amd.aligner import AlignOptions, CorpusAligner, CorpusUtterance   # noqa: E402
from montreal_forced_aligner_amd.engine import AlignmentEngine           # noqa: E402
from oracle import oracle as O                                           # noqa: E402
from tests import helpers                                                # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
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
pt = world.lexicon.phone_table
bad = 0
for seed in range(seed0, seed0 + n_seeds):
    rng = np.random.default_rng(52000 + seed)
    n_utt = int(rng.integers(2, 40))
    n_spk = int(rng.integers(1, 5))
    utts, raw = [], []
    for i in range(n_utt):
        nw = int(rng.choice([1, 2, 3, 5, 8, 15, 30, 60]))
        ns = int(nw * rng.integers(4000, 9000)) + int(rng.integers(0, 160))
        if rng.random() < 0.1:
            ns = int(rng.integers(500, 4000))                      # shorter than the transcript can be spoken in: fails, alone
        spk = int(rng.integers(0, n_spk))
        pcm, text, _segs, _ = world.utterance(60000 + 100 * seed + i, n_words=nw, samples=ns, speaker=spk)
        raw.append((pcm, text, spk))
        utts.append(CorpusUtterance(f"s{spk}-{i}", f"s{spk}", pcm, text))
    spk_order = list(dict.fromkeys(u.speaker for u in utts))
    prev = fm[np.array([int(s[1:]) for s in spk_order]) % 16]
    al = CorpusAligner(model.tm, model.am, model.tree, world.lexicon, lda=lda, engine=eng,
                       options=AlignOptions(beam=10.0, retry_beam=40.0, batch_frames=int(rng.choice([800, 3000, 20000, 200000]))),
                       silence_phones=[pt.find("sil"), pt.find("spn")])
    t0 = time.time()
    res = al.align(utts, make_ctm=bool(rng.random() < 0.5), previous_transforms=prev)
    t_dev = time.time() - t0
    # oracle: MFCC of every utterance, CMVN per speaker over all of the speaker's utterances, splice + LDA + fMLLR, lazy decodable
    mf = [O.mfcc(p.astype(np.float32), O.default_mfcc_opts()) for p, _t, _s in raw]
    n_bad = n_fail = n_tie = 0
    for k, (pcm, text, spk) in enumerate(raw):
        if mf[k].shape[0] == 0:
            ok = res[k] is None
        else:
            cm = O.cmvn_stats([mf[j] for j in range(n_utt) if raw[j][2] == spk and mf[j].shape[0] > 0])
            x = O.affine(O.affine(O.splice(O.cmvn_apply(cm, mf[k])), lda), fm[spk % 16])
            fst = G.add_transition_probs(gc.compile_fst(text), scaled)
            ref = helpers.oracle_align_feats(model.tm, fst, x, model.am, beam=10.0, retry_beam=40.0)
            if ref["status"] not in (0, 1):
                ok = res[k] is None
                n_fail += 1
            else:
                r = res[k]
                ok = (r is not None and np.array_equal(r.alignment, ref["ali"]) and np.array_equal(r.words, ref["words"])
                      and abs(r.per_frame_likelihood - ref["like"] / len(ref["ali"])) < 1e-3)
                if (not ok and r is not None and len(r.alignment) == len(ref["ali"]) and np.array_equal(r.words, ref["words"])
                        and abs(r.per_frame_likelihood - ref["like"] / len(ref["ali"])) < 1e-4
                        and int((np.asarray(r.alignment) != ref["ali"]).sum()) <= max(3, len(ref["ali"]) // 200)):
                    # a near-tie decided by features that differ in the fifth digit (device MFCC vs the oracle's): the same
                    # words, the same likelihood to 1e-4 per frame, a boundary or two a frame apart.  tools/pipeline_fuzz_diag.py
                    # tells the two apart (the oracle fed the device's features agrees with the device).
                    n_tie += 1
                    ok = True
        if not ok:
            n_bad += 1
            r = res[k]
            why = "device failed" if r is None else ("ali differs at %d frames" % int((np.asarray(r.alignment) != ref["ali"]).sum())
                                                     if ref["status"] in (0, 1) and len(r.alignment) == len(ref["ali"]) else "other")
            print(f"  seed {seed} utterance {k} ({len(pcm)} samples, {len(text.split())} words, speaker {spk}): {why}; "
                  f"reason {al.failure_reasons.get(utts[k].utt_id)}", flush=True)
    bad += n_bad
    print(f"{seed}: {n_utt} utterances, {n_spk} speakers, batch_frames {al.opt.batch_frames}, {n_fail} unalignable, "
          f"{n_tie} near-ties, {n_bad} mismatches, device {t_dev:.2f} s", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
