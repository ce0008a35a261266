#!/usr/bin/env python3
"""bench.py — aligned utterances/s of the MI355X alignment hot path (BASELINE.json metric).

A step = one pass of the whole device path (MFCC → CMVN → splice+LDA+fMLLR → diagonal-GMM scores → beam Viterbi) over
one batch of DISTINCT synthetic 10 s / 16 kHz utterances, alignments copied back to pinned host memory inside the step.
Workload = BASELINE.json configs[2]: context-dependent SAT-style model (~5k pdfs × 32 Gaussians, D = 40, per-speaker
fMLLR), beam 10 / retry 40.  `--workload mono` runs configs[1] (monophone, 1 Gaussian/state, Δ+ΔΔ features) instead.

`value` follows the bench contract: inputs (PCM, graphs) resident in HBM when the timed region starts.  The same JSON
line also carries, each from its own timed loop in this very run:
  value_host_fed — PCM in pinned host memory, double-buffered H2D on a copy stream under the previous step's kernels,
                   outputs back on the host (SURVEY §8d "PCM in host pinned memory → alignments in host memory");
  value_bf16x3 / value_f32 — the same resident-input loop with the stricter scoring arithmetic (MFA_GMM_F16=0 /
                   MFA_GMM_BF16=0).

N > 1: one process per GPU; `python bench.py --gpus N` started without a torch.distributed environment launches its own
N ranks (torch.distributed.run as a CHILD process, before anything touches the GPU) and relays rank 0's line.
Utterances are sharded by speaker, every rank runs the same per-GPU batch (weak scaling), no data-path collective;
ranks meet only at the timing barriers.

Extra objects in the line:
  roofline     — dominant kernel of the step, measured with HIP events on the launch stream
  cpu_baseline — the CPU oracle (a port, not stock Kaldi) timed on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak (no sparsity)
HBM_PEAK_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8192,
                    help="utterances per step per GPU (8 192 x 6 in flight hold about 160 GB of the 288 GB of HBM; measured 243 k "
                         "utterances/s against 230 k with 4 096 x 6 and 232 k with 4 096 x 8)")
    ap.add_argument("--pool", type=int, default=0,
                    help="distinct synthetic utterances generated per rank (0 = one per batch slot: all distinct)")
    ap.add_argument("--workload", choices=["triphone", "mono"], default="triphone")
    ap.add_argument("--train-utts", type=int, default=120)
    ap.add_argument("--gauss-per-pdf", type=int, default=32,  # 0 = mixture sizes as in a trained model (1..48, median 11)
                    help="triphone workload: Gaussians per pdf (BASELINE configs[2] = 32; other values exercise the other "
                         "slot classes of the scoring kernels and are NOT the headline workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-loops", action="store_true", help="skip the host-fed and strict-precision loops")
    ap.add_argument("--cpu-sample", type=int, default=0, help="utterances for the CPU baseline (0 = 8 per core)")
    ap.add_argument("--beam", type=float, default=10.0)
    ap.add_argument("--retry-beam", type=float, default=40.0, help="0 = no retry pass (diagnostic; not the headline config)")
    ap.add_argument("--max-tokens", type=int, default=256)
    ap.add_argument("--bp-tokens", type=int, default=128)
    ap.add_argument("--reachability", type=int, default=1,
                    help="dense-scoring path only: 1 = score a pdf only from the first frame the decoder can ask for it")
    ap.add_argument("--lazy", type=int, default=1,
                    help="1: windowed scoring of the cells live decoder tokens can ask for (mfa_align_features_batch); "
                         "0: score the whole (reachability-bounded) matrix, then decode")
    ap.add_argument("--window", type=int, default=64, help="frames per scoring/decoding window of the lazy path")
    ap.add_argument("--stream-priorities", default="", help="experiment: comma-separated HIP stream priorities, cycled over the "
                                                            "batches in flight (e.g. \"-1,0\"); empty = all default")
    ap.add_argument("--inflight", type=int, default=6,
                    help="batches in flight per GPU: steps alternate between this many pipelines (own HIP stream, engine "
                         "context and buffers each), so that the short latency-bound tails of a step — retry-beam and "
                         "table-growth passes over a handful of utterances — run under the next steps' kernels, and "
                         "latency-bound kernels of different steps share the chip (GPU_MAX_HW_QUEUES is raised to 8: with the "
                         "runtime's default of 4 hardware queues more than three streams queue up behind each other)")
    ap.add_argument("--hf-buffers", type=int, default=2,
                    help="device PCM buffers per pipeline in the host-fed loop (2: the H2D of a pipeline's next step "
                         "travels under its current step)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse "
                                                           "several ranks on one GPU)")
    ap.add_argument("--verbose", action="store_true")
    return ap.parse_args(argv)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------ self-launch (N > 1)
def self_launch(args) -> int:
    """`python bench.py --gpus N` without a torch.distributed environment: start the N ranks as a child process
    (never exec: this parent stays GPU-free), relay rank 0's JSON line, return the child's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    print("[bench] launching", " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("[bench] child ranks exited 0 without a result line", file=sys.stderr, flush=True)
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------ worker pool (CPU side)
# Created before torch / HIP are imported in this process: forking a process that has initialised the GPU runtime is
# unsupported, so the workers are forked first and fed later (utterance generation, graph compilation, CPU baseline).
_WORLD = None


def _gen_chunk(task):
    """Generate utterances + their training graphs (worker process; numpy/scipy only)."""
    indices, tm, tree = task
    from montreal_forced_aligner_amd import graph as G

    gc = G.TrainingGraphCompiler(tm, tree, _WORLD.lexicon)
    scaled = tm.scaled_log_probs(1.0, 0.1)
    out = []
    for i in indices:
        pcm, text, _segs, spk = _WORLD.utterance(int(i))
        out.append((pcm, G.add_transition_probs(gc.compile_fst(text), scaled), spk, text))
    return out


_CPU_SHARED = {}


def _cpu_one(task):
    """Whole oracle path for one utterance (worker process) — the cpu_baseline leg, the only user of oracle/ here.
    Scores are evaluated as Kaldi's decodable evaluates them: lazily, a (frame, pdf) cell when a live token's arc first
    asks for it, cached for the frame (oracle/mfa_oracle.cpp `Decodable`, lazy form).  Returns (status, cells evaluated).
    The acoustic model (51 MB for configs[2]), LDA and transition-id table are shared through memory-mapped .npy files —
    pickling them into every task cost more than the alignment itself."""
    (pcm, spk_fm, graph, shared_dir, mono) = task
    from oracle import oracle as O

    sh = _CPU_SHARED.get(shared_dir)
    if sh is None:
        sh = _CPU_SHARED[shared_dir] = {k: np.load(os.path.join(shared_dir, k + ".npy"), mmap_mode="r")
                                        for k in ("gconsts", "means_invvars", "inv_vars", "pdf_offsets", "tid2pdf", "lda")
                                        if os.path.exists(os.path.join(shared_dir, k + ".npy"))}
    mf = O.mfcc(pcm.astype(np.float32), O.default_mfcc_opts())
    x = O.cmvn_apply(O.cmvn_stats([mf]), mf)
    if mono:
        feats = O.deltas(x)
    else:
        feats = O.affine(O.affine(O.splice(x), sh["lda"]), spk_fm)
    r = O.align_feats(graph[0], graph[1], graph[2], graph[3], graph[4], feats, sh["gconsts"], sh["means_invvars"], sh["inv_vars"],
                      sh["pdf_offsets"], sh["tid2pdf"], 0.1, 10.0, 40.0)
    return r["status"], r["cells"]


def main():
    args = parse_args()
    # One hardware queue per pipeline stream: read by the HIP runtime when it initialises (nothing has touched it yet, in
    # this process or — the environment is inherited — in the ranks self_launch starts).  Measured: 3 pipelines 185 k
    # utterances/s with 4 or 8 queues; 6 pipelines 184 k with 4 queues, 203 k with 8.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE {world} does not match --gpus {args.gpus}")

    # ---- CPU-only setup, before anything touches the GPU: synthetic world + forked worker pool
    global _WORLD
    import multiprocessing as mp

    import synth_workload as synth

    t_setup = time.time()
    _WORLD = world_ = synth.SynthWorld.build()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # a 1-GPU box grants a 16-core CPU share (of a much larger host): never size the pool by os.cpu_count() alone
    cores = max(1, min(16, avail // max(1, world)))
    pool_proc = mp.get_context("fork").Pool(cores)
    log(rank, f"synthetic world built in {time.time() - t_setup:.1f}s; {cores} worker processes forked (GPU not yet touched)")

    import torch
    import torch.distributed as dist

    from montreal_forced_aligner_amd import sharding
    from montreal_forced_aligner_amd.engine import AlignmentEngine, Pipeline

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
    if os.environ.get("MFA_BENCH_FAIL_RANK") == str(rank):
        # test hook (tests/test_gpu_multirank_bench.py): a rank that dies must take the whole run down with a non-zero
        # exit code, never leave rank 0 printing a line for a job that did not finish
        raise SystemExit(f"rank {rank}: failing on request (MFA_BENCH_FAIL_RANK)")
    mono = args.workload == "mono"
    eng = AlignmentEngine(local_rank)
    eng.configure_mfcc()  # MFA defaults: 25/10 ms, 23 mel bins, 13 ceps, snip_edges False, dither 0
    dev = eng.device

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

    def feature_fn(pcm, spk):  # called per training utterance by the synthetic trainer
        return device_features([pcm], [spk])[0]

    t0 = time.time()
    trainer = synth.train_monophone if mono else synth.train_triphone
    model = trainer(world_, feature_fn, n_train=args.train_utts) if mono or args.gauss_per_pdf == 32 else \
        trainer(world_, feature_fn, n_train=args.train_utts, n_gauss=args.gauss_per_pdf)
    log(rank, f"synthetic {'monophone' if mono else 'triphone'} model: {model.am.num_pdfs} pdfs, {model.am.num_gauss} Gaussians, "
              f"dim {model.am.dim}, {model.tm.num_transition_ids} transition-ids ({time.time() - t0:.1f}s)")
    eng.load_gmm(model.am)

    # ---- the rank's utterances (all distinct by default) and their graphs, generated by the worker pool
    t0 = time.time()
    B = args.batch
    n_pool = args.pool if args.pool > 0 else B
    n_pool = min(n_pool, B)
    ids = rank * 1_000_003 + np.arange(n_pool)
    chunks = np.array_split(ids, max(1, min(len(ids), cores * 4)))
    pool = [u for part in pool_proc.map(_gen_chunk, [(c, model.tm, model.tree) for c in chunks], chunksize=1) for u in part]
    log(rank, f"{n_pool} distinct utterances + graphs in {time.time() - t0:.1f}s; graph states "
              f"{np.mean([p[1].num_states for p in pool]):.0f} avg / {max(p[1].num_states for p in pool)} max, arcs "
              f"{np.mean([p[1].num_arcs for p in pool]):.0f} avg")
    idx = np.arange(B) % n_pool
    copy = np.arange(B) // n_pool
    utt_spk = np.array([pool[i][2] for i in idx], dtype=np.int64) + 1000 * copy  # copies are distinct speakers
    # (sharding across ranks is by speaker, as the reference's jobs: each rank owns whole speakers — here each rank
    #  generates its own speakers' utterances, so the assignment is the identity; the rule itself is unit-tested)
    _ = sharding.assign_speakers(utt_spk, 1)
    pcm_host = torch.from_numpy(np.concatenate([pool[i][0] for i in idx]))
    try:
        pcm_host = pcm_host.pin_memory()
        pinned = True
    except RuntimeError:   # pinned allocation can be refused in constrained containers
        pinned = False
    pcm_all = pcm_host.to(dev)
    sample_off = np.concatenate([[0], np.cumsum([len(pool[i][0]) for i in idx])]).astype(np.int64)
    t0 = time.time()
    fsts = [pool[i][1] for i in idx]
    graphs = eng.pack_graphs(fsts, model.tm)
    log(rank, f"batch of {B} graphs packed in {time.time() - t0:.1f}s")
    ids_k, inv_k = np.unique(utt_spk, return_inverse=True)
    fm_k = None if mono else torch.from_numpy(fm_np[ids_k % n_spk_total]).to(dev)
    # one Pipeline per batch in flight: own stream, own engine context (decoder workspace), own feature / score / output
    # buffers; the PCM, the graphs and the model arrays are read-only and shared
    n_inflight = max(1, args.inflight)
    engines, pipes, streams = [], [], []
    for k in range(n_inflight):
        prio = [int(x) for x in args.stream_priorities.split(",")] if args.stream_priorities else None
        st = torch.cuda.current_stream(dev) if (k == 0 and prio is None) else \
            torch.cuda.Stream(dev, priority=prio[k % len(prio)] if prio else 0)
        with torch.cuda.stream(st):
            e_k = eng
            if k > 0:
                e_k = AlignmentEngine(local_rank)
                e_k.configure_mfcc()
                e_k.load_gmm(model.am)
            pipes.append(Pipeline(e_k, pcm_all, sample_off, inv_k.astype(np.int32), graphs, lda=d_lda, fmllr=fm_k,
                                  beam=args.beam, retry_beam=args.retry_beam, max_tokens=args.max_tokens,
                                  bp_tokens_per_frame=args.bp_tokens, reachability=bool(args.reachability),
                                  lazy=bool(args.lazy), window=args.window))
        engines.append(e_k)
        streams.append(st)
    pipe = pipes[0]
    torch.cuda.synchronize()
    log(rank, f"setup {time.time() - t_setup:.1f}s; HBM in use {torch.cuda.memory_allocated(dev) / 2**30:.1f} GiB "
              f"(+ Viterbi workspace); audio per step {pipe.audio_seconds:.0f}s")

    def barrier():
        if world > 1:
            dist.barrier()

    # how much of the score matrix does one step write?  (lazy scoring: the cells inside the decoder's bands) — the work
    # the scoring stage executes, which is what the roofline prices
    sc = pipe.measure_scored_cells()
    for p_ in pipes[1:]:
        p_.cells_scored, p_.cells_scored_fraction, p_.scored_flops = pipe.cells_scored, pipe.cells_scored_fraction, pipe.scored_flops
    log(rank, f"score cells written: {int(sc['cells'])}/{pipe.loglikes.numel()} = {sc['fraction']:.4f}; "
              f"{sc['flops'] / 1e12:.3f} TFLOP executed per step ({pipe.gmm_flops / 1e12:.3f} for the whole reachable matrix)")
    if os.environ.get("MFA_GMM_STAMPS"):  # diagnostic (library built with -DGMM_BAND_STAMPS): band kernel phase ticks (100 MHz)
        import ctypes as C
        acc = torch.zeros(16, dtype=torch.int64, device=dev)
        eng.lib.mfa_debug_gmm_trace(eng.ctx, C.c_void_p(acc.data_ptr()))
        pipe.step()
        torch.cuda.synchronize()
        eng.lib.mfa_debug_gmm_trace(eng.ctx, None)
        a = acc.cpu().numpy().astype(np.float64)
        w = max(a[3], 1.0)
        log(rank, f"band kernel per wavefront: search {a[0] / w / 100:.2f} us, feature split {a[1] / w / 100:.2f} us, block loop "
                  f"{a[2] / w / 100:.2f} us over {a[4] / w:.1f} blocks ({a[2] / max(a[4], 1) / 100:.3f} us per block); {int(a[3])} wavefronts")
        nb = max(a[4], 1.0)
        log(rank, "band kernel block loop, shader-clock cycles per block: MFMA phase (issue + operand waits) %.0f, lookup wait %.0f, "
                  "log-sum-exp + stage %.0f, flush %.0f" % tuple(a[8 + k] / nb for k in range(4)))
    if os.environ.get("MFA_VIT_STAMPS"):  # diagnostic (library built with -DVIT_STAMPS): decoder phase cycles → .npy
        import ctypes as C
        stamps = torch.zeros(pipe.n_utt * 12, dtype=torch.int64, device=dev)
        eng.lib.mfa_debug_viterbi_stamps(eng.ctx, C.c_void_p(stamps.data_ptr()))
        pipe.step()
        torch.cuda.synchronize()
        eng.lib.mfa_debug_viterbi_stamps(eng.ctx, None)
        np.save(os.environ["MFA_VIT_STAMPS"], stamps.cpu().numpy().reshape(pipe.n_utt, 12))

    # pinned host buffers the alignments land in (the boundary's host side): ali, words, n_words, like, status
    host_outs = [p_.host_output_buffers(pinned) for p_ in pipes]
    host_out = host_outs[0]
    turn = {"i": 0}

    def step_resident():
        k = turn["i"] % n_inflight
        turn["i"] += 1
        with torch.cuda.stream(streams[k]):
            pipes[k].step()
            pipes[k].outputs_to_host(host_outs[k])

    for _ in range(max(args.warmup, n_inflight)):   # (every pipeline runs at least once before the clock starts)
        step_resident()
    torch.cuda.synchronize()
    status = host_out["status"].numpy().copy()
    n_ok = int(((status == 0) | (status == 1)).sum())
    log(rank, f"warmup done: {n_ok}/{B} aligned, status counts {dict(zip(*np.unique(status, return_counts=True)))}")
    diagnostic = os.environ.get("MFA_GMM_DIAG", "0") != "0"  # timing-only kernel variants produce no valid scores
    if n_ok < 0.98 * B and not diagnostic:
        raise SystemExit(f"benchmark invalid: only {n_ok}/{B} utterances aligned")

    def timed_loop(step_fn, steps):
        """EXACTLY `steps` steps between barrier + synchronize on both sides; max over ranks."""
        barrier()
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
        barrier()
        dt_ = time.perf_counter() - t_
        if world > 1:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_

    for e_ in engines:
        e_.kernel_timing(True)
        e_.reset_kernel_times()
    dt = timed_loop(step_resident, args.steps)
    ktimes = {}
    for e_ in engines:     # HIP-event times on each pipeline's own stream (kernels of two streams may overlap in wall time)
        for k_, v_ in e_.kernel_times().items():
            acc_ = ktimes.setdefault(k_, dict(ms=0.0, launches=0))
            acc_["ms"] += v_["ms"]
            acc_["launches"] += v_["launches"]
        e_.kernel_timing(False)
    total_utts = B * args.steps * world
    value = total_utts / dt

    extra = {}
    if not args.no_extra_loops:
        # ---- host-fed loop: PCM from pinned host memory; every pipeline in flight has its own device PCM buffer and copy
        # stream, so the H2D of one step travels under the kernels of the other pipeline's step (and, with one pipeline,
        # two buffers alternate: the copy of step i+1 under the kernels of step i)
        n_hf = max(3, min(args.steps, 10))
        n_buf = args.hf_buffers
        bufs = [[torch.empty_like(pcm_all) for _ in range(n_buf)] for _ in pipes]
        copy_streams = [torch.cuda.Stream(dev) for _ in pipes]
        ev_copied = [[torch.cuda.Event() for _ in range(n_buf)] for _ in pipes]
        ev_free = [[torch.cuda.Event() for _ in range(n_buf)] for _ in pipes]
        for k in range(n_inflight):
            for b_ in range(n_buf):
                ev_free[k][b_].record(streams[k])
        hf = {"i": 0, "queued": [0] * n_inflight}

        def enqueue_copy(k, j):
            b_ = j % n_buf
            with torch.cuda.stream(copy_streams[k]):
                copy_streams[k].wait_event(ev_free[k][b_])      # the step that last read this buffer has finished
                bufs[k][b_].copy_(pcm_host, non_blocking=True)
                ev_copied[k][b_].record(copy_streams[k])

        def step_host_fed():
            i = hf["i"]
            k, j = i % n_inflight, i // n_inflight
            hf["i"] = i + 1
            while hf["queued"][k] <= j + (n_buf - 1):            # this step's copy, and the next one's when double-buffered
                enqueue_copy(k, hf["queued"][k])
                hf["queued"][k] += 1
            b_ = j % n_buf
            with torch.cuda.stream(streams[k]):
                streams[k].wait_event(ev_copied[k][b_])
                pipes[k].pcm = bufs[k][b_]
                pipes[k].step()
                ev_free[k][b_].record(streams[k])
                pipes[k].outputs_to_host(host_outs[k])

        for _ in range(2 * n_inflight):
            step_host_fed()                                  # warm-up (first copies not overlapped)
        torch.cuda.synchronize()
        dt_hf = timed_loop(step_host_fed, n_hf)
        torch.cuda.synchronize()
        for p_ in pipes:
            p_.pcm = pcm_all
        extra["value_host_fed"] = round(B * n_hf * world / dt_hf, 2)
        extra["host_fed"] = {"steps": n_hf, "ms_per_step": round(dt_hf / n_hf * 1e3, 3), "pinned": pinned,
                             "pcm_bytes_per_step": int(pcm_all.numel() * 2),
                             "h2d_GBps_needed": round(pcm_all.numel() * 2 / (dt_hf / n_hf) / 1e9, 2),
                             "what": "PCM in pinned host memory -> H2D on a copy stream under other steps' kernels -> device "
                                     "path -> ali/words/n_words/like/status in pinned host memory"}
        del bufs
        # ---- strict-precision scoring, same resident-input loop (driver-verifiable): bf16x3 and bit-exact f32
        n_px = max(2, min(args.steps, 5))
        for name, env in (("value_bf16x3", {"MFA_GMM_F16": "0"}), ("value_f32", {"MFA_GMM_BF16": "0"})):
            if mono:
                break   # the monophone model is single-Gaussian: already on the bit-exact f32 kernel
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                for _ in range(n_inflight):
                    step_resident()
                torch.cuda.synchronize()
                st2 = host_out["status"].numpy()
                ok2 = int(((st2 == 0) | (st2 == 1)).sum())
                dt_p = timed_loop(step_resident, n_px)
                extra[name] = round(B * n_px * world / dt_p, 2)
                extra.setdefault("precision_loops", {})[name] = {"steps": n_px, "ms_per_step": round(dt_p / n_px * 1e3, 3),
                                                                 "aligned_fraction": ok2 / B, "env": env}
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        for _ in range(n_inflight):
            step_resident()
        torch.cuda.synchronize()

    # ---- end to end through the host side (SURVEY §8d restated for configs[3]: PCM + transcripts in host memory ->
    # alignments on the host -> interval arrays -> TextGrid text): CorpusAligner over the same utterances, every rank its
    # own, barrier + max over ranks like the device loops
    if not args.no_extra_loops:
        import shutil
        import tempfile

        from montreal_forced_aligner_amd.aligner import AlignOptions, CorpusAligner, CorpusUtterance

        pt = world_.lexicon.phone_table
        utts_e2e = [CorpusUtterance(f"s{int(utt_spk[b_])}-{b_}", f"s{int(utt_spk[b_])}", pool[idx[b_]][0], pool[idx[b_]][3],
                                    file_name=f"u{b_:05d}") for b_ in range(B)]
        spk_order = list(dict.fromkeys(u.speaker for u in utts_e2e))
        prev_tf = None if mono else fm_np[np.array([int(s_[1:]) for s_ in spk_order]) % n_spk_total]
        ca = CorpusAligner(model.tm, model.am, model.tree, world_.lexicon, lda=lda_np, engine=eng,
                           options=AlignOptions(beam=args.beam, retry_beam=args.retry_beam, max_tokens=args.max_tokens,
                                                bp_tokens_per_frame=args.bp_tokens, batch_frames=1024 * 1001),
                           silence_phones=[pt.find("sil"), pt.find("spn")])
        for _ in range(3):    # warm-up at full size: the pinned staging buffers come in rotating sets of two and three
            ca.align(utts_e2e, make_ctm=False, previous_transforms=prev_tf)
        e2e = {}
        holder = {}
        if os.environ.get("MFA_BENCH_PROFILE_E2E"):     # where the host loop's time goes (stderr)
            import cProfile
            import pstats
            pr_ = cProfile.Profile()
            pr_.enable()
            ca.align(utts_e2e, make_ctm=False, previous_transforms=prev_tf)
            pr_.disable()
            st_ = pstats.Stats(pr_, stream=sys.stderr).sort_stats("cumulative")
            st_.print_stats(40)
            st_.print_callers("wait|synchronize")
        # three timed passes, the median reported (a pass is 0.15-0.25 s of host threads next to 16 idle worker processes:
        # single passes scatter by a factor of 1.5; every pass is listed)
        dts_a = sorted(timed_loop(lambda: holder.__setitem__("res", ca.align(utts_e2e, make_ctm=False, previous_transforms=prev_tf)), 1)
                       for _ in range(3))
        dt_a = dts_a[1]
        ok_a = sum(r is not None for r in holder["res"])
        e2e["alignments"] = {"value": round(B * world / dt_a, 2), "seconds": round(dt_a, 3), "aligned_fraction": ok_a / B,
                             "passes_seconds": [round(x, 3) for x in dts_a]}
        out_dir = Path(tempfile.mkdtemp(prefix="mfa_bench_tg_"))

        def to_textgrids():
            res_ = ca.align(utts_e2e, make_ctm=True, previous_transforms=prev_tf)
            holder["files"] = ca.export_textgrids(utts_e2e, res_, out_dir)

        try:
            dts_t = sorted(timed_loop(to_textgrids, 1) for _ in range(3))
            dt_t = dts_t[1]
            n_files = len(holder["files"])
            tg_bytes = sum(f_.stat().st_size for f_ in holder["files"])
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
        e2e["textgrids"] = {"value": round(B * world / dt_t, 2), "seconds": round(dt_t, 3), "files": n_files, "bytes": tg_bytes,
                            "passes_seconds": [round(x, 3) for x in dts_t]}
        e2e["what"] = ("CorpusAligner, one process per GPU: int16 PCM + transcripts in host memory -> graphs compiled -> device "
                       "path -> alignments on the host ('alignments'); -> phone/word intervals -> one long-format TextGrid "
                       "file per utterance written to a temporary directory ('textgrids')")
        extra["value_end_to_end"] = e2e["alignments"]["value"]
        extra["value_end_to_end_textgrid"] = e2e["textgrids"]["value"]
        extra["end_to_end"] = e2e

    # ---- CPU baseline (rank 0, one GPU): the oracle with Kaldi's lazy decodable on a bounded sample.  Run before the
    # rooflines are written: its count of score cells the decoder asks for prices the Viterbi stage's score bytes.
    cpu_baseline, score_cells_read = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import shutil
        import tempfile

        n_s = args.cpu_sample or min(n_pool, (64 if mono else 48) * cores)
        am = model.am
        shared_dir = tempfile.mkdtemp(prefix="mfa_bench_cpu_")
        for k_, v_ in (("gconsts", am.gconsts), ("means_invvars", am.means_invvars), ("inv_vars", am.inv_vars),
                       ("pdf_offsets", am.pdf_offsets), ("tid2pdf", np.maximum(model.tm.id2pdf, 0).astype(np.int32))) + \
                (() if mono else (("lda", lda_np),)):
            np.save(os.path.join(shared_dir, k_ + ".npy"), np.ascontiguousarray(v_))
        sample = []
        for i in range(n_s):
            pcm, fst, spk, _text = pool[i]
            sample.append((pcm, None if mono else fm_np[spk % n_spk_total],
                           (fst.num_states, fst.start, fst.arc_offsets, fst.arcs, fst.final), shared_dir, mono))
        log(rank, f"CPU baseline: oracle (lazy decodable) on {n_s} utterances over {cores} worker processes ...")
        try:
            pool_proc.map(_cpu_one, sample[:cores], chunksize=1)          # workers map the shared model (untimed)
            t0 = time.time()
            st = pool_proc.map(_cpu_one, sample, chunksize=1)
            secs = time.time() - t0
        finally:
            shutil.rmtree(shared_dir, ignore_errors=True)
        rate = n_s / secs
        cells_per_utt = float(np.mean([c for _s, c in st]))
        score_cells_read = cells_per_utt * B
        cpu_baseline = {"value": round(rate, 3), "unit": "utterances/s", "cores": cores, "kind": "port",
                        "sample": f"{n_s} utterances of the same workload, full oracle path (MFCC .. Viterbi), scores evaluated "
                                  f"lazily and cached per frame as Kaldi's decodable does ({cells_per_utt:.0f} (frame, pdf) cells "
                                  f"per utterance), {secs:.1f}s wall, one process per core; CPU restatement, not stock Kaldi",
                        "score_cells_per_utterance": round(cells_per_utt, 1)}

    # ---- the same stages with ONE batch in flight (no other stream's kernels sharing the chip): clean per-kernel times
    single = None
    kt_ss, n_ss = None, 0
    if n_inflight > 1 and not args.no_extra_loops:
        n_ss = 2
        engines[0].kernel_timing(True)
        engines[0].reset_kernel_times()
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        for _ in range(n_ss):
            pipes[0].step()
            pipes[0].outputs_to_host(host_outs[0])
        torch.cuda.synchronize()
        dt_ss = time.perf_counter() - t_
        kt_ss = engines[0].kernel_times()
        engines[0].kernel_timing(False)
        single = {"steps": n_ss, "ms_per_step": round(dt_ss / n_ss * 1e3, 3),
                  "stage_ms_per_step": {k: round(v["ms"] / n_ss, 3) for k, v in kt_ss.items()},
                  "launches_per_step": {k: v["launches"] / n_ss for k, v in kt_ss.items()},
                  "roofline_scoring": pipe.roofline("gmm", kt_ss, n_ss, mono, args.gauss_per_pdf),
                  "roofline_viterbi": pipe.roofline("viterbi", kt_ss, n_ss, mono, args.gauss_per_pdf, score_cells_read)}

    # ---- roofline of the dominant kernel of the step (largest share of the per-stage HIP-event times; with several
    # batches in flight these are measured while other streams' kernels share the chip — what rocprofv3 sees too)
    stage_ms = {k: v["ms"] / args.steps for k, v in ktimes.items()}
    dominant = max(stage_ms, key=stage_ms.get)
    roofline_concurrent = pipe.roofline(dominant, ktimes, args.steps, mono, args.gauss_per_pdf, score_cells_read)
    roofline_concurrent["measured"] = (f"timed region, {n_inflight} batches in flight: launch durations include the time other "
                                       "streams' kernels share the chip")
    if kt_ss is not None:
        # the kernel's own roofline: its launch durations with the chip to itself (the one-batch-in-flight loop of this
        # very run, HIP events on the launch stream) — what tools/profile_round.sh's --inflight 1 stats pass and the PMC
        # passes see too.  The figure taken inside the timed region stays in the line as roofline_batches_in_flight.
        roofline = pipe.roofline(dominant, kt_ss, n_ss, mono, args.gauss_per_pdf, score_cells_read)   # (dominant: by the timed region's stage times)
        roofline["measured"] = "one batch in flight (same run, after the timed region): the kernel has the chip to itself"
    else:
        roofline = roofline_concurrent
    # fabric-side bytes of that kernel's launches: PMC counters cannot be collected from inside this process, so the
    # figure is the one tools/profile_round.sh measured (separate rocprofv3 --pmc passes of this very command), committed
    # under profiles/; it applies to the default workload at the batch size in the file's name
    prof_name = next((n_ for n_ in (f"r03_profile_summary_triphone_b{B}.json", f"r02_profile_summary_triphone_b{B}.json")
                      if (ROOT / "profiles" / n_).exists()), None)
    if not mono and prof_name is not None:
        try:
            with open(ROOT / "profiles" / prof_name) as fh:
                tr = json.load(fh).get("bench_roofline_traffic", {}).get(roofline.get("kernel_key", ""))
            if tr:
                roofline["traffic"] = float(tr["bytes_per_step"])
                roofline["traffic_source"] = f"profiles/{prof_name}: {tr['how']}"
                # the same launches against the HBM roof (north_star asks for that fraction too): fabric bytes per step
                # over the stage's time per step, of 8 TB/s
                if roofline.get("ms_per_step"):
                    gbps = roofline["traffic"] / (roofline["ms_per_step"] * 1e-3) / 1e9
                    roofline["hbm"] = {"achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 4)}
        except (KeyError, ValueError, OSError):
            pass

    gpp = model.am.num_gauss // model.am.num_pdfs
    out = {
        "metric": "aligned utterances/sec (whole node), 10 s utts, 5k-state triphone" if not mono
        else "aligned utterances/sec (whole node), 10 s utts, monophone",
        "value": round(value, 2), "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": pipe.dtype_string(mono, args.gauss_per_pdf), "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[2]: synthetic 10 s 16 kHz utterances, context-dependent SAT-style GMM "
                         f"({model.am.num_pdfs} pdfs x {gpp} Gaussians, D={model.am.dim}) "
                         "+ per-speaker fMLLR, beam 10 / retry 40") if not mono else
                        (f"BASELINE configs[1]: synthetic 10 s 16 kHz utterances, monophone GMM ({model.am.num_pdfs} pdfs, "
                         "1 Gaussian/state, D=39), beam 10 / retry 40"),
            "batch_per_gpu": B, "utterances_total": total_utts, "distinct_utterances_per_gpu": n_pool,
            "frames_per_utt": int(pipe.max_frames), "parallelism": f"utterance-sharded x{world}, no collective",
            "inputs": "PCM + graphs resident in HBM; alignments copied to pinned host memory inside every step",
            "batches_in_flight": n_inflight, "hip_hardware_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
            "scores": pipe.scores_string(),
        },
        "value_definition": ("value = the bench contract's rate: inputs (PCM, graphs) resident in HBM when the timed region "
                             "starts, alignments copied to pinned host memory inside every step.  value_host_fed = SURVEY 8(d)'s "
                             "wording (PCM in pinned host memory -> alignments in host memory, PCIe inside the timed region); "
                             "value_end_to_end[_textgrid] = from PCM + transcripts on the host through graph compilation to "
                             "alignments [and TextGrid files]"),
        "value_resident": round(value, 2),
        "real_time_factor": dt / (pipe.audio_seconds * args.steps * world),
        "aligned_fraction": n_ok / B, **({"diagnostic_variant": os.environ["MFA_GMM_DIAG"]} if diagnostic else {}),
        "stage_ms_per_step": {k: round(v, 3) for k, v in stage_ms.items()},
        **extra,
        "roofline": roofline,
        "roofline_batches_in_flight": roofline_concurrent,
        **({"single_batch_in_flight": single} if single is not None else {}),
    }
    if cpu_baseline is not None:
        out["cpu_baseline"] = cpu_baseline
        out["gpu_over_cpu"] = round(value / cpu_baseline["value"], 1) if cpu_baseline["value"] > 0 else None
    pool_proc.close()
    pool_proc.join()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    for e_ in engines:
        e_.close()


if __name__ == "__main__":
    main()
