"""Batched many-pair mode across the GPUs of one node.

Pairs are independent units (the reference aligns them one after another, pa-bin/src/main.rs:24-35), so the path shards with NO
data-path collective: every rank (one process per GPU) aligns chunks of pairs on its own GPU, and the only exchange is the result
gather at the end -- `torch.distributed` over RCCL on GPUs, gloo in the CPU tests.  The single-pair path stays on one GPU
("replicas only").

Work distribution is a QUEUE, not a static plan: the pairs are sorted by estimated work (heaviest first) and cut into chunks; a rank
that is free takes the next chunk by bumping ONE counter in the process group's key-value store (the rendezvous TCPStore every
torch.distributed job already has: an atomic `add`).  The estimate only orders the queue; the balance is dynamic, which is what
band-limited alignment needs -- the work of a pair depends on its divergence, which nobody knows beforehand.  Without a store the
chunks are dealt out longest-processing-time-first (`plan_shards`).

Results: by default every rank ends with the full, ordered result (two tensor all-gathers: a fixed-size header per rank, then the
padded rows); with `all_ranks=False` they are gathered to rank 0 only and the other ranks return None.
"""
from __future__ import annotations

from typing import Callable, Sequence


def work_estimate(a_len: int, b_len: int, band_words: int | None = None) -> int:
    """Word updates of the full DP: n * ceil(m/64); with `band_words` the band's words per column instead."""
    w = (b_len + 63) // 64
    if band_words is not None:
        w = min(w, max(1, band_words))
    return a_len * w + 1


def band_words_hint(a_len: int, b_len: int, divergence: float) -> int:
    """Words per column of an A*PA2 band for an expected edit rate (band doubling ends at the first power-of-two multiple of 256
    that holds the cost; the band is about that many rows high)."""
    cost = abs(a_len - b_len) + divergence * max(a_len, b_len)
    f = 256
    while f < cost:
        f *= 2
    return f // 64 + 2


def plan_shards(work: Sequence[int], world: int) -> list[list[int]]:
    """Longest-processing-time-first assignment of indices to ranks (deterministic); the static fallback of the queue."""
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


def plan_chunks(work: Sequence[int], world: int, min_chunk: int = 256, per_rank: int = 8) -> list[list[int]]:
    """The queue: indices sorted by work (heaviest first, ties by index), cut into chunks of about len / (world * per_rank) pairs,
    at least `min_chunk` (a chunk must still fill a GPU), one chunk per rank when there are few pairs."""
    n = len(work)
    if n == 0:
        return []
    order = sorted(range(n), key=lambda i: (-work[i], i))
    chunk = max(min_chunk, n // (world * per_rank) + 1)
    if chunk * world > n:
        # few pairs: one chunk per rank.  Contiguous slices of the heaviest-first order would hand rank 0 all the heavy pairs (most of
        # the n * m work of a length-skewed input), and with one chunk each the queue cannot correct that: deal them out
        # longest-processing-time-first instead
        return [sh for sh in plan_shards(work, min(world, n)) if sh]
    chunk = max(chunk, 1)
    return [sorted(order[k:k + chunk]) for k in range(0, n, chunk)]


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


def astarpa2_align(params):
    """-> a compute function: (cost, CIGAR) of `pairs` by the batched A*PA2 of pa_batch_create_params (band-limited; the `simple`
    preset and its relatives)."""

    def run(pairs):
        from . import capi

        if not pairs:
            return []
        batch = capi.Batch(list(pairs), params=params)
        try:
            costs, cigars, _, _ = batch.align()
        finally:
            batch.close()
        return [(int(c), g) for c, g in zip(costs, cigars)]

    return run


def sharded_align(pairs: Sequence[tuple[bytes, bytes]], compute: Callable | None = None, group=None, all_ranks: bool = True,
                  work: Sequence[int] | None = None, min_chunk: int = 256):
    """(cost, CIGAR) of every pair.  Every rank passes the same `pairs` (any indexable sequence: only the pairs of the chunks a rank
    takes are read once `work` is given).  The exchange: a fixed-size header gather and one padded row gather -- tensors, not pickled
    objects."""
    return _sharded(pairs, compute or default_align, group, True, all_ranks, work, min_chunk)


def sharded_costs(pairs: Sequence[tuple[bytes, bytes]], compute: Callable | None = None, group=None, all_ranks: bool = True,
                  work: Sequence[int] | None = None, min_chunk: int = 256):
    """Edit distance of every pair, computed by the ranks of the default (or given) process group."""
    return _sharded(pairs, compute or default_compute, group, False, all_ranks, work, min_chunk)


def _queue_store(dist, group):
    """The key-value store behind the process group (atomic add), or None."""
    try:
        from torch.distributed import distributed_c10d as c10d

        if group is not None and group is not dist.group.WORLD:
            return None  # (sub-groups: the static plan; their ranks would need a key space of their own)
        return c10d._get_default_store()
    except Exception:
        return None


def _sharded(pairs, compute, group, with_cigar, all_ranks, work, min_chunk):
    import numpy as np
    import torch
    import torch.distributed as dist

    import time as _time

    t_call = _time.perf_counter()
    if not (dist.is_available() and dist.is_initialized()):
        # One process, one GPU: the same queue of chunks, taken in order.  A batch's device buffers grow with its pairs (the traceback
        # keeps 0.6 MB of re-fill scratch per 10 kbp pair: 100 000 pairs at once are 85 GB of hipMalloc, seconds of it), so a long list
        # goes through the GPU in the chunks a rank of a larger job would take.
        if work is None:
            work = [work_estimate(len(a), len(b)) for a, b in pairs]
        chunks = plan_chunks(work, 1, min_chunk=min_chunk)
        tim = {"plan_s": _time.perf_counter() - t_call, "queue_s": 0.0, "compute_s": 0.0, "gather_s": 0.0, "unpack_s": 0.0,
               "chunks": len(chunks), "pairs": len(pairs)}
        n = len(pairs)
        res_all: list = [None] * n
        for ch in chunks:
            tc = _time.perf_counter()
            res = list(compute([pairs[i] for i in ch]))
            tim["compute_s"] += _time.perf_counter() - tc
            for i, x in zip(ch, res):
                res_all[i] = (int(x[0]), str(x[1])) if with_cigar else int(x)
        sharded_last_chunks[:] = list(range(len(chunks)))
        tim["total_s"] = _time.perf_counter() - t_call
        sharded_last_timing.clear()
        sharded_last_timing.update(tim)
        return res_all
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if work is None:
        work = [work_estimate(len(a), len(b)) for a, b in pairs]
    tim = {"plan_s": 0.0, "queue_s": 0.0, "compute_s": 0.0, "gather_s": 0.0, "unpack_s": 0.0, "chunks": 0, "pairs": 0}
    chunks = plan_chunks(work, world, min_chunk=min_chunk)  # the same queue on every rank
    tim["plan_s"] = _time.perf_counter() - t_call
    store = _queue_store(dist, group)
    # ---- pull chunks until the queue is empty ----
    mine: list[int] = []      # pair indices in the order computed
    local: list = []
    taken: list[int] = []
    key = None
    if store is not None:
        # The queue's key must be the same on every rank of THIS call whatever else the ranks did before (calls on sub-groups, calls
        # without a store): rank 0 draws a fresh id from the store and broadcasts it.  (A per-process call counter drifts apart as soon
        # as one rank takes part in a call another does not, and then every rank drains a queue of its own.)
        qdev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        qid = torch.zeros(1, dtype=torch.int64, device=qdev)
        if rank == 0:
            qid[0] = int(store.add("pa_work_queue_ids", 1))
        # (`src` is a GLOBAL rank: on a sub-group that does not hold global rank 0 the drawer is the group's first member)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(qid, src=src, group=group)
        key = f"pa_work_queue_{int(qid.item())}"
        while True:
            tq = _time.perf_counter()
            c = int(store.add(key, 1)) - 1
            tim["queue_s"] += _time.perf_counter() - tq
            if c >= len(chunks):
                break
            taken.append(c)
            tc = _time.perf_counter()
            res = list(compute([pairs[i] for i in chunks[c]]))
            tim["compute_s"] += _time.perf_counter() - tc
            mine.extend(chunks[c])
            local.extend(res)
    else:
        static = plan_shards([sum(work[i] for i in ch) for ch in chunks], world)[rank]
        for c in static:
            taken.append(c)
            tc = _time.perf_counter()
            res = list(compute([pairs[i] for i in chunks[c]]))
            tim["compute_s"] += _time.perf_counter() - tc
            mine.extend(chunks[c])
            local.extend(res)
    sharded_last_chunks[:] = taken  # (tests / reporting: which chunks this rank took)
    tim["chunks"], tim["pairs"] = len(taken), len(mine)
    t_gather = _time.perf_counter()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    # ---- (1) how much every rank has: [pairs, text bytes] per rank, one tiny all_gather ----
    texts = [str(x[1]).encode() for x in local] if with_cigar else []
    blob = b"".join(texts)
    cnt = torch.tensor([len(mine), len(blob)], dtype=torch.int64, device=dev)
    cnts = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt, group=group)
    cnts = cnts.cpu().view(world, 2)
    cap = max(1, int(cnts[:, 0].max()))
    width = max(1, int(cnts[:, 1].max()))
    # ---- (2) per rank one padded int32 row block [cap, 3] = (pair index, cost, CIGAR length) and one padded byte row ----
    head = torch.zeros((cap, 3), dtype=torch.int32)
    if mine:
        arr = np.zeros((len(mine), 3), np.int32)
        arr[:, 0] = mine
        arr[:, 1] = [int(x[0]) if with_cigar else int(x) for x in local]
        if with_cigar:
            arr[:, 2] = [len(t) for t in texts]
        head[: len(mine)] = torch.from_numpy(arr)
    row = torch.zeros(width, dtype=torch.uint8)
    if blob:
        row[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    head, row = head.to(dev), row.to(dev)
    root = 0
    i_collect = all_ranks or rank == root
    if all_ranks:
        heads = torch.empty((world * cap, 3), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(heads, head, group=group)
        rows = None
        if with_cigar:
            rows = torch.empty(world * width, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(rows, row, group=group)
    else:
        groot = dist.get_global_rank(group, root) if group is not None else root
        hl = [torch.empty_like(head) for _ in range(world)] if rank == root else None
        dist.gather(head, hl, dst=groot, group=group)
        heads = torch.cat(hl) if rank == root else None
        rows = None
        if with_cigar:
            rl = [torch.empty_like(row) for _ in range(world)] if rank == root else None
            dist.gather(row, rl, dst=groot, group=group)
            rows = torch.cat(rl) if rank == root else None
    if key is not None and rank == 0:  # (the gathers above were the barrier: nobody pulls from this queue any more)
        try:
            store.delete_key(key)
        except Exception:
            pass
    tim["gather_s"] = _time.perf_counter() - t_gather
    tim["total_s"] = _time.perf_counter() - t_call
    sharded_last_timing.clear()
    sharded_last_timing.update(tim)
    if not i_collect:
        return None
    t_unpack = _time.perf_counter()
    heads = heads.cpu().view(world, cap, 3).numpy()
    n = len(pairs)
    costs = [0] * n
    out = [(0, "")] * n
    seen = 0
    raw_rows = rows.cpu().view(world, width).numpy() if with_cigar else None
    for r in range(world):
        k_r = int(cnts[r, 0])
        off = 0
        raw = raw_rows[r].tobytes() if with_cigar else b""
        for k in range(k_r):
            i, c, ln = int(heads[r, k, 0]), int(heads[r, k, 1]), int(heads[r, k, 2])
            costs[i] = c
            if with_cigar:
                out[i] = (c, raw[off:off + ln].decode())
                off += ln
            seen += 1
    if seen != n:
        raise RuntimeError(f"sharded run returned {seen} results for {n} pairs")
    sharded_last_timing["unpack_s"] = _time.perf_counter() - t_unpack
    sharded_last_timing["total_s"] = _time.perf_counter() - t_call
    return out if with_cigar else costs


sharded_last_chunks: list[int] = []
# Where the last sharded call of THIS rank went (seconds): planning the queue, waiting for the queue's counter (one round trip to the process
# group's store per chunk), inside `compute` (the GPU's share), the result gather, unpacking on the collecting rank; chunks and pairs taken.
sharded_last_timing: dict = {}
