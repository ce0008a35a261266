"""End-to-end rate of CorpusAligner on configs[2]-shaped utterances: int16 PCM + transcripts in host memory → alignments
(and, with --ctm, phone/word intervals) on the host; stage times of the host loop printed beside it.
GPU box:  python tools/corpus_rate.py [n_utt] [--ctm] [--full] [--textgrid] [--ragged]
--ragged: BASELINE configs[4]'s shape — utterance lengths log-uniform between 1 s and 30 s (three words per second) instead of 10 s each.
--full: the bench's model (4 960 pdfs x 32 Gaussians) and DISTINCT utterances, as bench.py's value_end_to_end loop."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import synth_workload as synth                                              # noqa: E402
from montreal_forced_aligner_amd.aligner import AlignOptions, CorpusAligner, CorpusUtterance   # noqa: E402
from montreal_forced_aligner_amd.engine import AlignmentEngine              # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2048
    want_ctm = "--ctm" in sys.argv or "--textgrid" in sys.argv
    full = "--full" in sys.argv
    import torch

    world = synth.SynthWorld.build()
    eng = AlignmentEngine(0)
    eng.configure_mfcc()
    lda = synth.seeded_lda()
    d_lda = torch.from_numpy(lda).to(eng.device)

    def feats_of(pcm, spk):
        so = np.array([0, len(pcm)], dtype=np.int64)
        mfcc, fo = eng.mfcc(torch.from_numpy(pcm).to(eng.device), so)
        own = np.zeros(1, dtype=np.int32)
        return eng.features(mfcc, fo, own, eng.cmvn_stats(mfcc, fo, own, 1), lda=d_lda).cpu().numpy()

    model = synth.train_triphone(world, feats_of, n_train=60, n_gauss=32, n_classes=5 if full else 2)
    pool = n if full else 128
    if "--ragged" in sys.argv:
        rng = np.random.default_rng(4)
        secs = np.exp(rng.uniform(np.log(1.0), np.log(30.0), size=pool))
        base = [world.utterance(20000 + i, n_words=max(1, int(round(3 * s_))), samples=int(16000 * s_)) for i, s_ in enumerate(secs)]
    else:
        base = [world.utterance(20000 + i, n_words=30) for i in range(pool)]
    utts = [CorpusUtterance(f"{base[i % pool][3]}-{i}", str(base[i % pool][3]), base[i % pool][0], base[i % pool][1]) for i in range(n)]
    al = CorpusAligner(model.tm, model.am, model.tree, world.lexicon, lda=lda, engine=eng,
                       options=AlignOptions(batch_frames=1_025_000))
    for _ in range(3):                                                      # warm-up at full size, twice: context windows, both sets of pinned staging buffers
        al.align(utts, make_ctm=want_ctm)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    res = al.align(utts, make_ctm=want_ctm)
    pr.disable()
    torch.cuda.synchronize()
    dt = time.time() - t0
    ok = sum(r is not None for r in res)
    audio = sum(len(u.pcm) for u in utts) / 16000.0
    print(f"CorpusAligner.align(make_ctm={want_ctm}): {n} utterances in {dt:.2f} s = {n / dt:.0f} utterances/s; {ok} aligned; "
          f"{audio / 3600:.2f} h of audio, real-time factor {dt / audio:.2e}", flush=True)
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    if "--textgrid" in sys.argv:
        import shutil
        import tempfile
        d = tempfile.mkdtemp()
        for u in utts:
            u.file_name = u.utt_id
        pr = cProfile.Profile()
        t0 = time.time()
        pr.enable()
        files = al.export_textgrids(utts, res, d)
        pr.disable()
        dt2 = time.time() - t0
        print(f"export_textgrids: {len(files)} files in {dt2:.2f} s = {n / dt2:.0f} utterances/s; with align {n / (dt + dt2):.0f}/s", flush=True)
        pstats.Stats(pr).sort_stats("cumulative").print_stats(15)
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
