"""The exact path the driver's N > 1 scaling run takes, rehearsed on one GPU: a child
``python bench.py --gpus 2`` self-launches two ``torch.distributed`` ranks (gloo here, so that both can share the one
device; ``nccl`` = RCCL on a multi-GPU node), each runs its own per-GPU batch, the ranks meet at the barriers, the step
time is the max over ranks and rank 0 prints ONE JSON line.  A rank that dies must fail the whole run.
Ref: SURVEY §8(e); the speaker → job rule of /root/reference/montreal_forced_aligner/corpus/base.py:994-1015 is what
``sharding.assign_speakers`` implements and tests/test_host_cpu.py checks."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
ARGS = ["--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "512", "--inflight", "2",
        "--train-utts", "40", "--no-extra-loops", "--no-cpu-baseline"]


def _run(extra_env=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)                      # the child must self-launch, not believe it already is a rank
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + ARGS, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_two_rank_bench_prints_one_line_for_the_whole_job():
    p = _run()
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["aligned_fraction"] == 1.0
    assert out["config"]["utterances_total"] == 2 * 2 * 512 and out["config"]["parallelism"].startswith("utterance-sharded x2")
    assert out["value"] > 0 and abs(out["value"] - out["config"]["utterances_total"] / (out["ms_per_step"] * 2 / 1e3)) < 0.01 * out["value"]
    assert "launching" in p.stderr and "torch.distributed.run" in p.stderr          # the self-launch happened
    assert out["roofline"]["frac"] > 0 and "cpu_baseline" not in out


def test_a_failing_rank_fails_the_run():
    p = _run({"MFA_BENCH_FAIL_RANK": "1"}, timeout=600)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
