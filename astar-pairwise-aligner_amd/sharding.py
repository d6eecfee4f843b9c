"""Batched many-pair mode across the GPUs of one node.

Pairs are independent units (the reference aligns them one after another, pa-bin/src/main.rs:24-35), so the
path shards with NO data-path collective: every rank (one process per GPU) aligns its own subset and the
only exchange is one small gather of (index, cost) at the end -- `torch.distributed` over RCCL on GPUs,
gloo in the CPU tests.  The single-pair path stays on one GPU ("replicas only").
"""
from __future__ import annotations

from typing import Callable, Sequence


def work_estimate(a_len: int, b_len: int) -> int:
    """Word updates of the full DP: n * ceil(m/64)."""
    return a_len * ((b_len + 63) // 64) + 1


def plan_shards(work: Sequence[int], world: int) -> list[list[int]]:
    """Longest-processing-time-first assignment of pair indices to ranks (deterministic)."""
    order = sorted(range(len(work)), key=lambda i: (-work[i], i))
    load = [0] * world
    shards: list[list[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += work[i]
    for s in shards:
        s.sort()
    return shards


def default_compute(pairs):
    """Full-DP edit distances of `pairs` on this rank's GPU (HIP strip kernels)."""
    from . import capi

    if not pairs:
        return []
    batch = capi.Batch(list(pairs))
    try:
        costs, _ = batch.run()
    finally:
        batch.close()
    return [int(c) for c in costs]


def default_align(pairs):
    """(cost, CIGAR) of `pairs` on this rank's GPU: checkpointing forward pass + device-side traceback (pa_batch_align)."""
    from . import capi

    if not pairs:
        return []
    batch = capi.Batch(list(pairs), trace=True)
    try:
        costs, cigars, _, _ = batch.align()
    finally:
        batch.close()
    return [(int(c), g) for c, g in zip(costs, cigars)]


def sharded_align(pairs: Sequence[tuple[bytes, bytes]], compute: Callable | None = None, group=None) -> list[tuple[int, str]]:
    """(cost, CIGAR) of every pair; same sharding and the same single gather as `sharded_costs` (the variable-size
    CIGARs travel in that one object gather)."""
    return _sharded(pairs, compute or default_align, group, lambda x: (int(x[0]), str(x[1])), (0, ""))


def sharded_costs(pairs: Sequence[tuple[bytes, bytes]], compute: Callable | None = None, group=None) -> list[int]:
    """Edit distance of every pair, computed by the ranks of the default (or given) process group.
    Every rank passes the same `pairs`; every rank returns the full, ordered result."""
    return _sharded(pairs, compute or default_compute, group, int, 0)


def _sharded(pairs, compute, group, conv, empty):
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [conv(x) for x in compute(list(pairs))]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shards = plan_shards([work_estimate(len(a), len(b)) for a, b in pairs], world)
    mine = shards[rank]
    local = list(compute([pairs[i] for i in mine]))
    gathered: list = [None] * world
    dist.all_gather_object(gathered, list(zip(mine, local)), group=group)  # the one exchange step
    out = [empty] * len(pairs)
    for part in gathered:
        for i, c in part:
            out[i] = conv(c)
    return out
