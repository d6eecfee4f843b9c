"""Batched many-pair mode across the GPUs of one node.

Pairs are independent units (the reference aligns them one after another, pa-bin/src/main.rs:24-35), so the
path shards with NO data-path collective: every rank (one process per GPU) aligns its own subset and the
only exchange is one small tensor gather of the costs (plus one padded byte gather of the CIGAR text) at the end -- `torch.distributed` over RCCL on GPUs,
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
    """(cost, CIGAR) of every pair; same sharding as `sharded_costs`.  The exchange: one fixed-size gather of (cost, CIGAR length)
    and one padded byte gather of the CIGAR text -- tensors, not pickled objects."""
    return _sharded(pairs, compute or default_align, group, True)


def sharded_costs(pairs: Sequence[tuple[bytes, bytes]], compute: Callable | None = None, group=None) -> list[int]:
    """Edit distance of every pair, computed by the ranks of the default (or given) process group.
    Every rank passes the same `pairs`; every rank returns the full, ordered result."""
    return _sharded(pairs, compute or default_compute, group, False)


def _sharded(pairs, compute, group, with_cigar):
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        local = compute(list(pairs))
        return [(int(c), str(g)) for c, g in local] if with_cigar else [int(c) for c in local]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shards = plan_shards([work_estimate(len(a), len(b)) for a, b in pairs], world)  # the same plan on every rank
    mine = shards[rank]
    local = list(compute([pairs[i] for i in mine]))
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    cap = max(len(sh) for sh in shards)
    # (1) costs and CIGAR lengths: [cap, 2] int32 per rank, one all_gather
    head = torch.zeros((cap, 2), dtype=torch.int32)
    texts = []
    for k, x in enumerate(local):
        if with_cigar:
            t = str(x[1]).encode()
            texts.append(t)
            head[k, 0], head[k, 1] = int(x[0]), len(t)
        else:
            head[k, 0] = int(x)
    head = head.to(dev)
    heads = torch.empty((world * cap, 2), dtype=torch.int32, device=dev)  # (the concatenated form: gloo accepts no other)
    dist.all_gather_into_tensor(heads, head, group=group)
    heads = heads.cpu().view(world, cap, 2)
    costs = [0] * len(pairs)
    for r in range(world):
        for k, i in enumerate(shards[r]):
            costs[i] = int(heads[r, k, 0])
    if not with_cigar:
        return costs
    # (2) the CIGAR text of each rank as one padded byte row, one all_gather
    width = max(1, int(heads[:, :, 1].sum(dim=1).max()))
    row = torch.zeros(width, dtype=torch.uint8)
    blob = b"".join(texts)
    if blob:
        row[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    rows = torch.empty(world * width, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(rows, row.to(dev), group=group)
    rows = rows.cpu().view(world, width).numpy()
    out = [(0, "")] * len(pairs)
    for r in range(world):
        off = 0
        raw = rows[r].tobytes()
        for k, i in enumerate(shards[r]):
            ln = int(heads[r, k, 1])
            out[i] = (costs[i], raw[off:off + ln].decode())
            off += ln
    return out
