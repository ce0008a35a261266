#!/usr/bin/env python3
"""bench.py — aligned utterances/s of the MI355X alignment hot path (BASELINE.json metric).

A step = one pass of the whole device path (MFCC → CMVN → splice+LDA+fMLLR → diagonal-GMM scores → beam Viterbi) over
one batch of synthetic 10 s / 16 kHz utterances already resident in HBM.  Workload = BASELINE.json configs[2]:
context-dependent SAT-style model (~5k pdfs × 32 Gaussians, D = 40, per-speaker fMLLR), beam 10 / retry 40.
`--workload mono` runs configs[1] (monophone, 1 Gaussian/state, Δ+ΔΔ features) instead.

N > 1: one process per GPU (torch.distributed.run); utterances are sharded by speaker, every rank runs the same
per-GPU batch (weak scaling), no data-path collective; ranks meet only at the timing barriers.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     — dominant kernel (GMM scoring on the matrix pipe: f16x2 split by default, bf16x3 with MFA_GMM_F16=0, f32 with
                 MFA_GMM_BF16=0): executed MFMA TFLOP/s measured with HIP events on the launch stream
  cpu_baseline — the CPU oracle (a port, not stock Kaldi) timed on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="utterances per step per GPU")
    ap.add_argument("--pool", type=int, default=128, help="distinct synthetic utterances generated per rank")
    ap.add_argument("--workload", choices=["triphone", "mono"], default="triphone")
    ap.add_argument("--train-utts", type=int, default=120)
    ap.add_argument("--gauss-per-pdf", type=int, default=32,  # 0 = mixture sizes as in a trained model (1..48, median 11)
                    help="triphone workload: Gaussians per pdf (BASELINE configs[2] = 32; other values exercise the other "
                         "slot classes of the scoring kernels and are NOT the headline workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="utterances for the CPU baseline (0 = 2 per core)")
    ap.add_argument("--max-tokens", type=int, default=256)
    ap.add_argument("--bp-tokens", type=int, default=128)
    ap.add_argument("--reachability", type=int, default=1,
                    help="1: score a pdf only from the first frame the decoder can ask for it (default); 0: dense matrix")
    ap.add_argument("--streams", type=int, default=1, help="split the batch over this many HIP streams (stage overlap)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse "
                                                           "several ranks on one GPU)")
    ap.add_argument("--verbose", action="store_true")
    return ap.parse_args()


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------ CPU baseline (oracle)
def _cpu_one(args):
    """Whole oracle path for one utterance (runs in a worker process)."""
    (pcm, spk_fm, lda, graph, pdf_list, tid2col, am, mono) = args
    from oracle import oracle as O

    mf = O.mfcc(pcm.astype(np.float32), O.default_mfcc_opts())
    x = O.cmvn_apply(O.cmvn_stats([mf]), mf)
    if mono:
        feats = O.deltas(x)
    else:
        feats = O.affine(O.affine(O.splice(x), lda), spk_fm)
    ll = O.gmm_loglikes(feats, am[0], am[1], am[2], am[3], pdf_list)
    r = O.align(graph[0], graph[1], graph[2], graph[3], graph[4], ll, tid2col, 0.1, 10.0, 40.0)
    return r["status"]


def cpu_baseline(sample, cores):
    import multiprocessing as mp

    t0 = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        st = pool.map(_cpu_one, sample, chunksize=1)
    dt = time.time() - t0
    return len(sample) / dt, dt, st


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    import torch
    import torch.distributed as dist

    from montreal_forced_aligner_amd import graph as G
    from montreal_forced_aligner_amd import sharding
    from montreal_forced_aligner_amd.engine import AlignmentEngine, Pipeline
    from tests import synth

    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs a GPU: the alignment engine has no CPU fallback")
    if local_rank >= n_dev:
        if args.dist_backend == "nccl":
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible")
        local_rank %= n_dev  # rehearsal: several gloo ranks share one GPU
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    mono = args.workload == "mono"
    t_setup = time.time()
    eng = AlignmentEngine(local_rank)
    eng.configure_mfcc()  # MFA defaults: 25/10 ms, 23 mel bins, 13 ceps, snip_edges False, dither 0
    dev = eng.device
    world_ = synth.SynthWorld.build()
    log(rank, f"synthetic world built in {time.time() - t_setup:.1f}s")

    # ---- features through the device path (used to estimate the synthetic acoustic model and nothing else)
    lda_np = None if mono else synth.seeded_lda()
    n_spk_total = 1000
    fm_np = None if mono else synth.seeded_fmllr(n_spk_total)
    d_lda = None if mono else torch.from_numpy(lda_np).to(dev)

    def device_features(pcm_list, spks):
        sample_off = np.concatenate([[0], np.cumsum([len(p) for p in pcm_list])]).astype(np.int64)
        mfcc, frame_off = eng.mfcc(torch.from_numpy(np.concatenate(pcm_list)).to(dev), sample_off)
        u2s = np.arange(len(pcm_list), dtype=np.int32)
        stats = eng.cmvn_stats(mfcc, frame_off, u2s, len(pcm_list))
        if mono:
            f = eng.features(mfcc, frame_off, u2s, stats)
        else:
            fm = torch.from_numpy(fm_np[np.asarray(spks) % n_spk_total]).to(dev)
            f = eng.features(mfcc, frame_off, u2s, stats, lda=d_lda, fmllr=fm)
        f = f.cpu().numpy()
        return [f[frame_off[i]: frame_off[i + 1]] for i in range(len(pcm_list))]

    cache = {}

    def feature_fn(pcm, spk):  # called per training utterance by the synthetic trainer; batched lazily
        key = pcm.tobytes()[:64]
        if key not in cache:
            cache[key] = device_features([pcm], [spk])[0]
        return cache.pop(key)

    t0 = time.time()
    trainer = synth.train_monophone if mono else synth.train_triphone
    model = trainer(world_, feature_fn, n_train=args.train_utts) if mono or args.gauss_per_pdf == 32 else \
        trainer(world_, feature_fn, n_train=args.train_utts, n_gauss=args.gauss_per_pdf)
    log(rank, f"synthetic {'monophone' if mono else 'triphone'} model: {model.am.num_pdfs} pdfs, {model.am.num_gauss} Gaussians, "
              f"dim {model.am.dim}, {model.tm.num_transition_ids} transition-ids ({time.time() - t0:.1f}s)")
    eng.load_gmm(model.am)

    # ---- the rank's utterance pool → batch (weak scaling: every rank aligns `batch` utterances per step)
    t0 = time.time()
    gc = G.TrainingGraphCompiler(model.tm, model.tree, world_.lexicon)
    scaled = model.tm.scaled_log_probs(1.0, 0.1)
    pool = []
    for i in range(args.pool):
        pcm, text, _segs, spk = world_.utterance(rank * 1_000_003 + i)
        fst = G.add_transition_probs(gc.compile_fst(text), scaled)
        pool.append((pcm, fst, spk))
    log(rank, f"pool of {args.pool} utterances + graphs in {time.time() - t0:.1f}s; graph states "
              f"{np.mean([p[1].num_states for p in pool]):.0f} avg / {max(p[1].num_states for p in pool)} max, arcs "
              f"{np.mean([p[1].num_arcs for p in pool]):.0f} avg")
    B = args.batch
    idx = np.arange(B) % args.pool
    copy = np.arange(B) // args.pool
    utt_spk = np.array([pool[i][2] for i in idx], dtype=np.int64) + 1000 * copy  # copies are distinct speakers
    # (sharding across ranks is by speaker, as the reference's jobs: each rank owns whole speakers — here each rank
    #  generates its own speakers' utterances, so the assignment is the identity; the rule itself is unit-tested)
    _ = sharding.assign_speakers(utt_spk, 1)
    pcm_all = torch.from_numpy(np.concatenate([pool[i][0] for i in idx])).to(dev)
    sample_off = np.concatenate([[0], np.cumsum([len(pool[i][0]) for i in idx])]).astype(np.int64)
    t0 = time.time()
    packed_pool = eng.pack_graphs([p[1] for p in pool], model.tm)
    graphs = tile_graphs(eng, packed_pool, [p[1] for p in pool], idx)
    log(rank, f"batch of {B} graphs packed in {time.time() - t0:.1f}s")
    # one Pipeline per stream: sub-batches are independent (whole speakers each), so the latency-bound Viterbi of one
    # sub-batch overlaps the MFMA-bound scoring of the other.  Each stream has its own engine context and workspace.
    n_streams = max(1, args.streams)
    engines, pipes, streams = [eng], [], [torch.cuda.current_stream(dev)]
    for k in range(1, n_streams):
        st = torch.cuda.Stream(dev)
        streams.append(st)
        with torch.cuda.stream(st):
            e2 = AlignmentEngine(local_rank)
            e2.configure_mfcc()
            e2.load_gmm(model.am)
        engines.append(e2)
    bounds = np.linspace(0, B, n_streams + 1).astype(int)
    bounds = np.array([(b // args.pool) * args.pool if 0 < b < B and B % args.pool == 0 and B // args.pool >= n_streams else b
                       for b in bounds])
    for k in range(n_streams):
        lo, hi = int(bounds[k]), int(bounds[k + 1])
        sub = idx[lo:hi]
        with torch.cuda.stream(streams[k]):
            g_k = graphs if n_streams == 1 else tile_graphs(engines[k], packed_pool, [p[1] for p in pool], sub)
            so_k = np.concatenate([[0], np.cumsum([len(pool[i][0]) for i in sub])]).astype(np.int64)
            pcm_k = pcm_all[int(sample_off[lo]): int(sample_off[hi])]
            ids_k, inv_k = np.unique(utt_spk[lo:hi], return_inverse=True)
            fm_k = None if mono else torch.from_numpy(fm_np[ids_k % n_spk_total]).to(dev)
            pipes.append(Pipeline(engines[k], pcm_k, so_k, inv_k.astype(np.int32), g_k, lda=d_lda, fmllr=fm_k,
                                  max_tokens=args.max_tokens, bp_tokens_per_frame=args.bp_tokens,
                                  reachability=bool(args.reachability)))
    torch.cuda.synchronize()

    class _Multi:
        audio_seconds = sum(p.audio_seconds for p in pipes)
        gmm_flops = sum(p.gmm_flops for p in pipes)
        max_frames = max(p.max_frames for p in pipes)

        @staticmethod
        def step():
            for p in pipes:
                p.step()

        @property
        def status(self):
            return torch.cat([p.status for p in pipes])

    pipe = _Multi()
    log(rank, f"setup {time.time() - t_setup:.1f}s; HBM in use {torch.cuda.memory_allocated(dev) / 2**30:.1f} GiB "
              f"(+ Viterbi workspace); audio per step {pipe.audio_seconds:.0f}s; {n_streams} stream(s)")

    def barrier():
        if world > 1:
            dist.barrier()

    if os.environ.get("MFA_BENCH_OVERLAP_PROBE") and n_streams == 2:  # diagnostic: can the decoder of one half-batch
        pA, pB = pipes                                                   # share the chip with the front end of the other?
        pipe.step()
        torch.cuda.synchronize()

        def timed(fn):
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t_) * 1e3

        for _ in range(2):
            t_v, t_f, t_g = timed(pA.decode), timed(pB.front), timed(pB.score)
            t_vf = timed(lambda: (pA.decode(), pB.front()))
            t_vg = timed(lambda: (pA.decode(), pB.score()))
            log(rank, f"overlap probe: decode {t_v:.2f} ms, front {t_f:.2f} ms, score {t_g:.2f} ms; decode||front {t_vf:.2f} ms, "
                      f"decode||score {t_vg:.2f} ms")
    if os.environ.get("MFA_BENCH_FILL"):  # diagnostic: how much of the score matrix does one step write?
        for p_ in pipes:
            p_.loglikes.zero_()
        pipe.step()
        torch.cuda.synchronize()
        for p_ in pipes:
            nz = int((p_.loglikes != 0).sum().item())
            log(rank, f"score cells written: {nz}/{p_.loglikes.numel()} = {nz / p_.loglikes.numel():.4f}")
    if os.environ.get("MFA_VIT_STAMPS"):  # diagnostic (library built with -DVIT_STAMPS): decoder phase cycles → .npy
        import ctypes as C
        p_ = pipes[0]
        stamps = torch.zeros(p_.n_utt * 12, dtype=torch.int64, device=dev)
        eng.lib.mfa_debug_viterbi_stamps(eng.ctx, C.c_void_p(stamps.data_ptr()))
        pipe.step()
        torch.cuda.synchronize()
        eng.lib.mfa_debug_viterbi_stamps(eng.ctx, None)
        np.save(os.environ["MFA_VIT_STAMPS"], stamps.cpu().numpy().reshape(p_.n_utt, 12))
    if os.environ.get("MFA_GMM_TRACE"):  # diagnostic: per-wavefront timeline of one scoring launch → .npy
        import ctypes as C
        p_ = pipes[0]
        tiles = (p_.max_frames + 255) // 256
        trace = torch.zeros(p_.n_utt * tiles * 4 * 4, dtype=torch.int64, device=dev)
        pipe.step()
        torch.cuda.synchronize()
        eng.lib.mfa_debug_gmm_trace(eng.ctx, C.c_void_p(trace.data_ptr()))
        pipe.step()
        torch.cuda.synchronize()
        eng.lib.mfa_debug_gmm_trace(eng.ctx, None)
        np.save(os.environ["MFA_GMM_TRACE"], trace.cpu().numpy().reshape(p_.n_utt, tiles * 4, 4))
    for _ in range(args.warmup):
        pipe.step()
    torch.cuda.synchronize()
    status = pipe.status.cpu().numpy()
    n_ok = int(((status == 0) | (status == 1)).sum())
    log(rank, f"warmup done: {n_ok}/{B} aligned, status counts {dict(zip(*np.unique(status, return_counts=True)))}")
    diagnostic = os.environ.get("MFA_GMM_DIAG", "0") != "0"  # timing-only kernel variants produce no valid scores
    if n_ok < 0.98 * B and not diagnostic:
        raise SystemExit(f"benchmark invalid: only {n_ok}/{B} utterances aligned")

    for e_ in engines:
        e_.kernel_timing(True)
        e_.reset_kernel_times()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ktimes = {}
    for e_ in engines:
        for k_, v_ in e_.kernel_times().items():
            acc_ = ktimes.setdefault(k_, dict(ms=0.0, launches=0))
            acc_["ms"] += v_["ms"]
            acc_["launches"] += v_["launches"]
        e_.kernel_timing(False)

    total_utts = B * args.steps * world
    value = total_utts / dt
    gmm_ms = ktimes["gmm"]["ms"] / max(1, ktimes["gmm"]["launches"])
    flops_per_launch = pipe.gmm_flops / n_streams  # one scoring launch per stream per step
    achieved = flops_per_launch / (gmm_ms * 1e-3) / 1e12 if gmm_ms > 0 else 0.0
    # fabric-side bytes of one scoring launch: PMC counters cannot be collected from inside this process, so the figure
    # is the one measured by tools/profile_round.sh (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very
    # command) and committed under profiles/; it applies to the triphone workload at the batch size in the file's name
    traffic, traffic_src = None, None
    prof_name = f"r01_profile_summary_triphone_b{B}.json"
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", prof_name)
    if not mono and n_streams == 1 and args.reachability and os.path.exists(prof):
        try:
            with open(prof) as fh:
                ks_ = json.load(fh)["kernels"]
                if os.environ.get("MFA_GMM_BF16", "1") == "0":
                    d_ = ks_["gmm_kernel"]["derived"]
                elif os.environ.get("MFA_GMM_F16", "1") == "0":
                    d_ = (ks_.get("gmm_split_single_kernel_bf16") or ks_.get("gmm_bf16_single_kernel") or ks_["gmm_bf16_kernel"])["derived"]
                else:
                    d_ = ks_["gmm_split_single_kernel_f16"]["derived"]
            traffic = float(d_["fetch_bytes_per_dispatch_raw"] + d_["write_bytes_per_dispatch"])
            traffic_src = (f"profiles/{prof_name}: (FETCH_SIZE + WRITE_SIZE) KiB x 1024 per "
                           "launch, FETCH_SIZE uncorrected (gfx950 may tally 128-B reads at 64 B: up to 2x more)")
        except (KeyError, ValueError):
            pass
    # The scoring kernel of the headline workload.  Default: the f16×2 kernel — every float32 product is formed as three
    # f16 MFMA products (3·2^-22 per term worst case), so the matrix pipe executes 3× the algorithmic flops and is priced
    # against the dense f16 peak (same as bf16).  MFA_GMM_F16=0: the bf16×3 kernel, six products (2^-24 per term).
    # MFA_GMM_BF16=0 (or a model whose pdfs are not single 32-row blocks, like the monophone one): the f32 MFMA kernel
    # against the f32 peak.
    bf16 = os.environ.get("MFA_GMM_BF16", "1") != "0" and not mono and (1 < args.gauss_per_pdf <= 32 or args.gauss_per_pdf == 0)
    f16 = bf16 and os.environ.get("MFA_GMM_F16", "1") != "0"
    if bf16:
        mult = 3.0 if f16 else 6.0
        roofline = {
            "kernel": ("gmm_split_single_kernel<5,2> (diagonal-GMM scoring, 2-way f16 split on v_mfma_f32_32x32x16_f16)" if f16 else
                       "gmm_split_single_kernel<5,3> (diagonal-GMM scoring, 3-way bf16 split on v_mfma_f32_32x32x16_bf16)"),
            "bound": "mfma",
            "achieved": round(mult * achieved, 3), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(mult * achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "algorithmic_flops_per_launch": flops_per_launch, "mfma_flops_per_algorithmic_flop": int(mult),
            "f32_equivalent_tflops": round(achieved, 3), "avg_launch_ms": round(gmm_ms, 4),
            **({"traffic_source": traffic_src} if traffic is not None else {}),
        }
    else:
        roofline = {
            "kernel": "gmm_kernel (diagonal-GMM scoring, v_mfma_f32_32x32x2_f32)", "bound": "mfma",
            "achieved": round(achieved, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "algorithmic_flops_per_launch": flops_per_launch, "avg_launch_ms": round(gmm_ms, 4),
            **({"traffic_source": traffic_src} if traffic is not None else {}),
        }
    out = {
        "metric": "aligned utterances/sec (whole node), 10 s utts, 5k-state triphone" if not mono
        else "aligned utterances/sec (whole node), 10 s utts, monophone",
        "value": round(value, 2), "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f32 scores from 2-way f16-split products (3*2^-22 per term worst case), f64 path costs" if bf16 and f16
                  else "f32 scores from 3-way bf16-split products (2^-24 per term), f64 path costs" if bf16
                  else "f32 (scores), f64 (path costs)"), "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[2]: synthetic 10 s 16 kHz utterances, context-dependent SAT-style GMM "
                         f"({model.am.num_pdfs} pdfs x {model.am.num_gauss // model.am.num_pdfs} Gaussians, D={model.am.dim}) "
                         "+ per-speaker fMLLR, beam 10 / retry 40") if not mono else
                        (f"BASELINE configs[1]: synthetic 10 s 16 kHz utterances, monophone GMM ({model.am.num_pdfs} pdfs, "
                         "1 Gaussian/state, D=39), beam 10 / retry 40"),
            "batch_per_gpu": B, "utterances_total": total_utts, "distinct_utterances_per_gpu": args.pool,
            "frames_per_utt": int(pipe.max_frames), "parallelism": f"utterance-sharded x{world}, no collective",
            "streams_per_gpu": n_streams,
            "scores": "reachable cells only (pdf j from its first possible frame on)" if args.reachability else "dense T x P",
        },
        "real_time_factor": dt / (pipe.audio_seconds * args.steps * world),
        "aligned_fraction": n_ok / B, **({"diagnostic_variant": os.environ["MFA_GMM_DIAG"]} if diagnostic else {}),
        "stage_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in ktimes.items()},
        "roofline": roofline,
    }

    if rank == 0:
        # Host-fed variant of the boundary (PCM arrives in pinned host memory): the copy a step would need, timed on its
        # own stream.  Reported beside `value`, never part of it.
        try:
            host_pcm = torch.empty(pcm_all.shape, dtype=pcm_all.dtype, pin_memory=True)
            host_pcm.copy_(pcm_all.cpu())
            side = torch.cuda.Stream(dev)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(side):
                dst = torch.empty_like(pcm_all)
                dst.copy_(host_pcm, non_blocking=True)   # warm-up
                ev0.record(side)
                for _ in range(3):
                    dst.copy_(host_pcm, non_blocking=True)
                ev1.record(side)
            side.synchronize()
            h2d_ms = ev0.elapsed_time(ev1) / 3.0
            out["host_fed"] = {"pcm_bytes_per_step": int(pcm_all.numel() * 2), "h2d_ms_per_step_pinned": round(h2d_ms, 3),
                               "h2d_GBps": round(pcm_all.numel() * 2 / h2d_ms / 1e6, 2),
                               "note": "copy of one step's PCM from pinned host memory on a side stream; shorter than the "
                                       "step, so a double-buffered host-fed pipeline keeps `value`" if h2d_ms < dt / args.steps * 1e3
                                       else "copy longer than the step: a host-fed pipeline would be PCIe-bound"}
            del dst, host_pcm
        except RuntimeError as e:   # pinned allocation can be refused in constrained containers
            out["host_fed"] = {"error": str(e)[:200]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # a 1-GPU box grants a 16-core CPU share (of a much larger host): never size the pool by os.cpu_count() alone
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = max(1, min(16, avail))
        n_s = args.cpu_sample or min(args.pool, (16 if mono else 8) * cores)
        am = model.am
        tid2pdf = np.maximum(model.tm.id2pdf, 0)
        sample = []
        for i in range(n_s):
            pcm, fst, spk = pool[i]
            pl = packed_pool.pdf_lists_host[i]
            lut = np.zeros(am.num_pdfs, np.int32)
            lut[pl] = np.arange(len(pl), dtype=np.int32)
            sample.append((pcm, None if mono else fm_np[spk % n_spk_total], lda_np,
                           (fst.num_states, fst.start, fst.arc_offsets, fst.arcs, fst.final), pl,
                           lut[tid2pdf].astype(np.int32), (am.gconsts, am.means_invvars, am.inv_vars, am.pdf_offsets), mono))
        log(rank, f"CPU baseline: oracle on {n_s} utterances over {cores} processes ...")
        rate, secs, st = cpu_baseline(sample, cores)
        out["cpu_baseline"] = {"value": round(rate, 3), "unit": "utterances/s", "cores": cores, "kind": "port",
                               "sample": f"{n_s} utterances of the same workload, full oracle path (MFCC..Viterbi), "
                                         f"{secs:.1f}s wall, one process per core; CPU restatement, not stock Kaldi"}
        out["gpu_over_cpu"] = round(value / rate, 1) if rate > 0 else None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    for e_ in engines:
        e_.close()


def tile_graphs(eng, packed_pool, fsts, idx):
    """Batch = pool graphs repeated: every utterance gets its own copy in HBM (no sharing between utterances)."""
    import torch

    from montreal_forced_aligner_amd.engine import PackedGraphs

    n = len(idx)
    S = np.array([fsts[i].num_states for i in idx], dtype=np.int64)
    A = np.array([fsts[i].num_arcs for i in idx], dtype=np.int64)
    t = packed_pool.tensors
    host = {k: v.cpu().numpy() for k, v in t.items()}
    pS = np.concatenate([[0], np.cumsum([f.num_states for f in fsts])])
    pA = np.concatenate([[0], np.cumsum([f.num_arcs for f in fsts])])

    def cat_states(name):
        return np.concatenate([host[name][pS[i]: pS[i + 1]] for i in idx])

    def cat_arcs(name):
        return np.concatenate([host[name][pA[i]: pA[i + 1]] for i in idx])

    arc_off = np.concatenate([host["arc_off"][pS[i] + i: pS[i + 1] + i + 1] for i in idx])
    tensors = dict(
        state_off=eng._dev(np.concatenate([[0], np.cumsum(S)]).astype(np.int64)),
        arc_base=eng._dev(np.concatenate([[0], np.cumsum(A)]).astype(np.int64)),
        start=eng._dev(host["start"][idx]), arc_off=eng._dev(arc_off), final=eng._dev(cat_states("final")),
        arc_next=eng._dev(cat_arcs("arc_next")), arc_weight=eng._dev(cat_arcs("arc_weight")), arc_col=eng._dev(cat_arcs("arc_col")),
        arc_ilabel=eng._dev(cat_arcs("arc_ilabel")), arc_olabel=eng._dev(cat_arcs("arc_olabel")),
    )
    lists = [packed_pool.pdf_lists_host[i] for i in idx]
    pdf_off = np.concatenate([[0], np.cumsum([len(p) for p in lists])]).astype(np.int64)
    cc = packed_pool.class_counts.cpu().numpy()[idx]
    ffs = [packed_pool.pdf_first_frame_host[i] for i in idx]
    return PackedGraphs(n, int(S.max()), int(A.max()), int(A.sum()), tensors, eng._dev(np.concatenate(lists).astype(np.int32)),
                        eng._dev(pdf_off), eng._dev(cc.astype(np.int32)), pdf_off, lists,
                        eng._dev(np.concatenate(ffs).astype(np.int32)), ffs)


if __name__ == "__main__":
    main()
