"""Utterance → rank assignment for one-process-per-GPU runs.

The reference splits work over "jobs" by speaker with a greedy lightest-job rule
(MFA/corpus/base.py:994-1015), or by contiguous utterance ranges under --single_speaker (:979-993); speakers stay
together because CMVN and fMLLR statistics are per speaker.  The same rule assigns speakers to GPUs here, optionally
weighted by audio seconds instead of utterance counts.  The path needs no collective: a speaker's utterances live on
one rank, so every kernel is rank-local; ranks only meet at the bench's barrier and at the host-side gather of results.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np


def assign_speakers(utt2spk: Sequence[int], world_size: int, weights: Optional[Sequence[float]] = None) -> np.ndarray:
    """Returns rank_of_utt [n_utt].  Speakers sorted by total weight (descending; stable on speaker id, as the
    reference's sort), each given to the currently lightest rank (ties → lowest rank id, as ``min`` over job ids)."""
    utt2spk = np.asarray(utt2spk, dtype=np.int64)
    w = np.ones(utt2spk.shape[0]) if weights is None else np.asarray(weights, dtype=np.float64)
    spk_ids, inv = np.unique(utt2spk, return_inverse=True)
    load = np.bincount(inv, weights=w, minlength=spk_ids.shape[0])
    order = np.argsort(-load, kind="stable")
    rank_load = np.zeros(world_size)
    rank_of_spk = np.zeros(spk_ids.shape[0], dtype=np.int64)
    for s in order:
        if load[s] == 0:
            continue
        r = int(np.argmin(rank_load))
        rank_of_spk[s] = r
        rank_load[r] += load[s]
    return rank_of_spk[inv]


def assign_contiguous(n_utt: int, world_size: int) -> np.ndarray:
    """--single_speaker rule: contiguous ranges of int(n/world) utterances, remainder to the last rank."""
    per = max(1, n_utt // world_size)
    rank = np.minimum(np.arange(n_utt) // per, world_size - 1)
    return rank.astype(np.int64)


def local_indices(rank_of_utt: np.ndarray, rank: int) -> np.ndarray:
    return np.nonzero(rank_of_utt == rank)[0]


def gather_results(local: Dict[int, object], world_size: int, group=None) -> Dict[int, object]:
    """Host-side fan-in of per-utterance results {utt index: result} onto every rank (torch.distributed object gather;
    gloo on CPU, RCCL-backed process groups fall back to their CPU object path).  Not on the timed data path."""
    import torch.distributed as dist

    if world_size == 1 or not dist.is_initialized():
        return dict(local)
    parts: List[Optional[Dict[int, object]]] = [None] * world_size
    dist.all_gather_object(parts, local, group=group)
    out: Dict[int, object] = {}
    for p in parts:
        out.update(p)
    return out
