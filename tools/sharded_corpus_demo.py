"""BASELINE configs[3] at rehearsal scale: a multi-speaker corpus sharded by speaker over the ranks of a
``torch.distributed`` job (one process per GPU; no data-path collective), aligned, gathered on the host, written as TextGrid
files by rank 0 — and, on rank 0, compared with the same corpus aligned by one rank alone.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
        tools/sharded_corpus_demo.py --out DIR [--backend gloo]

With fewer GPUs than ranks (the one-GPU box) the ranks share device 0 — a rehearsal of the control flow, which is all that
differs from the one-rank run (every kernel is rank-local).  Corpus: the reference's fixture recording cut into utterances of
six "speakers" in four "files" (tests/golden/ref_fixtures)."""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="gloo")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist

    from montreal_forced_aligner_amd.aligner import AlignOptions, CorpusAligner, CorpusUtterance, align_sharded
    from montreal_forced_aligner_amd.engine import AlignmentEngine
    from tests import helpers

    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    dev = local_rank % max(1, torch.cuda.device_count())
    fx = helpers.Fixtures()
    sr = 16000
    segs = [(0.0, 4.2, "this is the acoustic corpus i'm talking pretty fast here"),
            (4.0, 6.5, "there's nothing going else going on"),
            (23.5, 26.72, "um and that should be all thanks")]
    utts = []
    for spk in range(6):                       # six speakers with 1–3 utterances each, in four files
        for k in range(1 + spk % 3):
            a, b, t = segs[(spk + k) % 3]
            gain = 0.5 + 0.1 * spk             # (distinct audio per speaker: per-speaker CMVN then differs too)
            pcm = np.clip(fx.pcm[int(a * sr): int(b * sr)].astype(np.float32) * gain, -32768, 32767).astype(np.int16)
            utts.append(CorpusUtterance(f"spk{spk}-{len(utts)}", f"spk{spk}", pcm, t, begin=10.0 * k, file_name=f"file{spk % 4}",
                                        file_duration=40.0))
    opts = AlignOptions(beam=100.0, retry_beam=400.0)

    def factory():
        return CorpusAligner(fx.mono_tm, fx.mono_am, fx.mono_tree, fx.mono_lex, options=opts, engine=AlignmentEngine(dev))

    res = align_sharded(factory, utts, rank=rank, world_size=world)
    ok = all(r is not None for r in res)
    report = {"world_size": world, "utterances": len(utts), "all_aligned": ok}
    if rank == 0:
        al = factory()
        paths = al.export_textgrids(utts, res, Path(args.out) / "textgrids")
        alone = al.align(utts)                 # the whole corpus on one rank
        same = all(a is not None and b is not None and np.array_equal(a.alignment, b.alignment) and
                   np.array_equal(a.words, b.words) and a.likelihood == b.likelihood for a, b in zip(res, alone))
        paths1 = al.export_textgrids(utts, alone, Path(args.out) / "textgrids_one_rank")
        same_files = [p.name for p in paths] == [p.name for p in paths1] and \
            all(p.read_bytes() == q.read_bytes() for p, q in zip(paths, paths1))
        report.update(textgrids=[p.name for p in paths], identical_to_one_rank=bool(same), identical_files=bool(same_files))
        Path(args.out).mkdir(parents=True, exist_ok=True)
        (Path(args.out) / "report.json").write_text(json.dumps(report))
        print(json.dumps(report), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
