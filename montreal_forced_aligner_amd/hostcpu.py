"""How many host threads the native host-side code (PCM gather, graph compiler, interval writer) may start.

``os.sched_getaffinity`` alone over-counts inside a container: a cgroup CPU quota (``cpu.max`` = "1600000 100000" → 16 CPUs)
leaves the affinity mask at every core of the machine (256 on the GPU box).  The budget here is twice the quota, capped by
the mask: the native calls are short bursts that wait on memory and on each other, and on the GPU box 32 threads against
the 16-CPU quota measured 30.5 k utterances/s sustained over 16 384 utterances against 28.5 k with 16 (tools/corpus_rate.py
16384 --full); a thread per core of the mask would be 256.
"""
from __future__ import annotations

import math
import os
from functools import lru_cache


def _cgroup_quota(root: str = "/sys/fs/cgroup") -> float:
    """CPUs the cgroup quota allows (inf when unlimited or unreadable)."""
    try:                                                    # cgroup v2
        with open(root + "/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max" and float(period) > 0:
            return float(quota) / float(period)
        return math.inf
    except (OSError, ValueError):
        pass
    try:                                                    # cgroup v1
        with open(root + "/cpu/cpu.cfs_quota_us") as fh:
            quota = float(fh.read())
        with open(root + "/cpu/cpu.cfs_period_us") as fh:
            period = float(fh.read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return math.inf


@lru_cache(maxsize=1)
def cpu_budget() -> int:
    """CPUs this process can actually run on at once: affinity mask ∩ cgroup quota (MFA_HOST_THREADS overrides)."""
    env = os.environ.get("MFA_HOST_THREADS")
    if env:
        return max(1, int(env))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = _cgroup_quota()
    if quota != math.inf:
        avail = min(avail, max(1, int(2 * quota)))
    # one process per GPU (torchrun / bench.py's self-launch export LOCAL_WORLD_SIZE): the ranks of a node share its cores
    try:
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        ranks = 1
    return max(1, avail // ranks)


def threads(share: float = 1.0, cap: int = 32) -> int:
    """Thread count for one native call: ``share`` of the budget, at most ``cap`` (the calls overlap in the corpus pipeline —
    the graph compiler runs on a worker thread under the PCM gather and the interval writer — so each takes a share)."""
    return max(1, min(cap, int(cpu_budget() * share)))
