"""The N>1 path on CPU: world_size-2 gloo run of the pair sharding + gather.
The per-rank compute is injected (CPU oracle) because this container has no GPU; on a GPU box the same
function runs the HIP batch (covered by test_gpu_sharded_single_rank)."""
import os
import socket
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import oracle
    from astar_pairwise_aligner_amd.sharding import sharded_costs
    from tests.util_seq import gen_pair

    dist.init_process_group("gloo", rank=rank, world_size=world)
    pairs = [gen_pair(200 + 37 * i, 0.01 * (i % 15 + 1), seed=i) for i in range(23)] + [(b"", b"ACG"), (b"ACGT", b"")]
    calls = []

    def compute(sub):
        calls.append(len(sub))
        return [oracle.levenshtein(a, b) for a, b in sub]

    out = sharded_costs(pairs, compute=compute, min_chunk=3)  # a queue of nine chunks, pulled by whoever is free
    from astar_pairwise_aligner_amd import sharding as _sh

    tim = dict(_sh.sharded_last_timing)  # round 6: where this rank's call went (bench.py prints it per rank)
    ok_tim = ({"plan_s", "queue_s", "compute_s", "gather_s", "total_s", "chunks", "pairs"} <= set(tim) and tim["pairs"] == sum(calls)
              and tim["chunks"] == len(calls) and tim["total_s"] >= tim["compute_s"] >= 0.0)
    want = [oracle.levenshtein(a, b) for a, b in pairs]
    # gathered to rank 0 only: the other rank gets None
    out0 = sharded_costs(pairs, compute=lambda sub: [oracle.levenshtein(a, b) for a, b in sub], all_ranks=False, min_chunk=5)
    ok_root = (out0 == want) if rank == 0 else (out0 is None)
    # the traceback variant gathers (cost, CIGAR): per-rank compute = the engine over the CPU oracle kernels
    from astar_pairwise_aligner_amd.sharding import sharded_align

    prm = oracle.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=True,
                             incremental_doubling=False, dt_trace=False)
    al = sharded_align(pairs[:9], compute=lambda sub: [oracle.cpu_align(a, b, prm)[:2] for a, b in sub], min_chunk=2)
    ok_al = all(c == w and oracle.cigar_verify(g, a, b) == c for (c, g), w, (a, b) in zip(al, want, pairs))
    al0 = sharded_align(pairs[:9], compute=lambda sub: [oracle.cpu_align(a, b, prm)[:2] for a, b in sub], min_chunk=2, all_ranks=False)
    ok_al0 = (al0 == al) if rank == 0 else (al0 is None)
    q.put((rank, out == want and ok_al and ok_root and ok_al0 and ok_tim, calls))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_costs_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sum(sum(c) for _, _, c in res) == 25  # every pair computed exactly once, in chunks of at most 3
    assert all(x <= 3 for _, _, c in res for x in c)


def _skew_worker(rank, world, port, q):
    """Uneven work the estimate cannot see (the band of a pair depends on its divergence): one kind of pair costs 15x the other."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time

    import torch.distributed as dist

    from astar_pairwise_aligner_amd.sharding import sharded_costs

    dist.init_process_group("gloo", rank=rank, world_size=world)
    # 96 pairs of equal length; the first 48 are "15 % divergent" (15 ms each), the rest "1 %" (1 ms): a static plan by length gives
    # every rank the same number of pairs whatever they cost -- and here rank 1 is also slower by half (a busy GPU)
    pairs = [(bytes([65 + (i % 4)]) * 50, b"A" * 50) for i in range(96)]
    busy = [0.0]

    def compute(sub):
        t = time.perf_counter()
        cost = sum(0.015 if i < 48 else 0.001 for i in current[0])
        time.sleep(cost * (1.5 if rank == 1 else 1.0))
        busy[0] += time.perf_counter() - t
        return [0] * len(sub)

    # the compute hook does not get indices: wrap the sequence so that the chunk's indices are known
    class Tracked:
        def __len__(self):
            return len(pairs)

        def __getitem__(self, i):
            current[0].append(i)
            return pairs[i]

        def __iter__(self):
            return iter(pairs)

    current = [[]]

    def compute_tracked(sub):
        r = compute(sub)
        current[0] = []
        return r

    out = sharded_costs(Tracked(), compute=compute_tracked, min_chunk=4, work=[1] * 96)
    q.put((rank, out == [0] * 96, busy[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_work_queue_balances_what_the_estimate_cannot_see():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_skew_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    t = sorted(b for _, _, b in res)
    assert t[1] <= 1.25 * t[0] + 0.08, t  # busy times within 25 % (+ one chunk of slack); a static split by length: 1.5x apart


def _subgroup_worker(rank, world, port, q):
    """A call on a sub-group only some ranks take part in, then calls on the whole world (round 3's advisor finding: with a per-process
    call counter the ranks then used different queue keys, every chunk was computed once per rank and the gather saw 2 n results)."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from astar_pairwise_aligner_amd.sharding import sharded_costs

    dist.init_process_group("gloo", rank=rank, world_size=world)
    solo = dist.new_group([0])  # (every rank creates the group, only rank 0 is in it)
    pairs = [(b"ACGT" * (i + 1), b"ACGA" * (i + 1)) for i in range(40)]
    want = [i + 1 for i in range(40)]
    calls = []

    def compute(sub):
        calls.append(len(sub))
        return [len(a) // 4 for a, _ in sub]

    ok = True
    if rank == 0:
        ok = sharded_costs(pairs[:7], compute=compute, group=solo, min_chunk=2) == want[:7]
        ok = ok and sharded_costs(pairs[:5], compute=compute, group=solo, min_chunk=2) == want[:5]
    before = sum(calls)
    for _ in range(3):  # world calls after the ranks' histories differ
        ok = ok and sharded_costs(pairs, compute=compute, min_chunk=4) == want
    q.put((rank, ok, sum(calls) - before))
    dist.barrier()
    dist.destroy_process_group()


def test_queue_key_survives_calls_on_subgroups():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sum(n for _, _, n in res) == 3 * 40  # every pair of the three world calls computed exactly once


def _subgroup_without_rank0_worker(rank, world, port, q):
    """A sub-group that does NOT hold global rank 0 (round 4's advisor finding: the queue id was broadcast with `src=0`, a GLOBAL rank that
    is not a member of such a group): ranks 1 and 2 of a world of three share one queue, then the whole world runs a call."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from astar_pairwise_aligner_amd.sharding import sharded_costs

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sub = dist.new_group([1, 2])  # (every rank creates the group; rank 0 is not in it)
    pairs = [(b"ACGT" * (i + 1), b"ACGA" * (i + 1)) for i in range(40)]
    want = [i + 1 for i in range(40)]
    calls = []

    def compute(part):
        calls.append(len(part))
        return [len(a) // 4 for a, _ in part]

    ok = True
    n_sub = 0
    if rank in (1, 2):
        ok = sharded_costs(pairs[:24], compute=compute, group=sub, min_chunk=2) == want[:24]
        n_sub = sum(calls)
    ok = ok and sharded_costs(pairs, compute=compute, min_chunk=4) == want
    q.put((rank, ok, n_sub, sum(calls) - n_sub))
    dist.barrier()
    dist.destroy_process_group()


def test_queue_on_a_subgroup_without_global_rank0():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_without_rank0_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert sum(n for _, _, n, _ in res) == 24  # the sub-group's pairs: computed once, by ranks 1 and 2 together
    assert sum(n for _, _, _, n in res) == 40  # the world call


def test_single_process_takes_the_same_chunks_in_order():
    """Without a process group the call still goes through the queue's chunks (a batch's device buffers grow with its pairs: round 6),
    results come back in the order of the pairs, and the call reports where its time went."""
    sys.path.insert(0, str(ROOT))
    import oracle
    from astar_pairwise_aligner_amd import sharding as sh
    from tests.util_seq import gen_pair

    pairs = [gen_pair(20 + 37 * (i % 11), 0.1, seed=100 + i) for i in range(50)]
    sizes = []

    def compute(sub):
        sizes.append(len(sub))
        return [oracle.levenshtein(a, b) for a, b in sub]

    out = sh.sharded_costs(pairs, compute=compute, min_chunk=8)
    assert out == [oracle.levenshtein(a, b) for a, b in pairs]
    assert len(sizes) > 1 and sum(sizes) == 50 and max(sizes) <= 8 + 1
    t = sh.sharded_last_timing
    assert t["chunks"] == len(sizes) and t["pairs"] == 50 and t["total_s"] >= t["compute_s"] >= 0.0
    prm = oracle.params_simple()
    al = sh.sharded_align(pairs[:12], compute=lambda sub: [oracle.cpu_align(a, b, prm)[:2] for a, b in sub], min_chunk=5)
    assert [c for c, _ in al] == out[:12] and all(oracle.cigar_verify(g, a, b) == c for (c, g), (a, b) in zip(al, pairs))


def test_few_pairs_are_dealt_by_work_not_sliced():
    from astar_pairwise_aligner_amd.sharding import plan_chunks

    work = [1000, 900, 800, 10, 9, 8, 7, 6]  # fewer pairs than world * min_chunk: one chunk per rank
    chunks = plan_chunks(work, world=4, min_chunk=256)
    assert sorted(i for c in chunks for i in c) == list(range(8)) and len(chunks) == 4
    loads = sorted(sum(work[i] for i in c) for c in chunks)
    assert loads[-1] == 1000 and loads[0] >= 40 - 1000  # the three heavy pairs sit in three different chunks
    assert sum(1 for c in chunks if any(work[i] >= 800 for i in c)) == 3


def test_plan_shards_balanced_and_deterministic():
    sys.path.insert(0, str(ROOT))
    from astar_pairwise_aligner_amd.sharding import plan_shards

    work = [100, 1, 1, 50, 50, 1, 99, 2]
    s = plan_shards(work, 2)
    assert sorted(i for part in s for i in part) == list(range(8))
    loads = [sum(work[i] for i in part) for part in s]
    assert max(loads) - min(loads) <= 50
    assert plan_shards(work, 2) == s
    assert plan_shards([], 4) == [[], [], [], []]
    assert plan_shards([5], 8)[0] == [0]


@pytest.mark.gpu
def test_gpu_sharded_single_rank():
    sys.path.insert(0, str(ROOT))
    import oracle
    from astar_pairwise_aligner_amd.sharding import sharded_costs
    from tests.util_seq import gen_pair

    pairs = [gen_pair(500 + 100 * i, 0.05, seed=i) for i in range(6)]
    assert sharded_costs(pairs) == [oracle.levenshtein(a, b) for a, b in pairs]


def _gpu_worker(rank, world, port, q):
    """Two ranks sharing GPU 0 (a one-GPU box): the default per-rank compute is the HIP batch; gloo carries the gather."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import oracle
    from astar_pairwise_aligner_amd.sharding import sharded_align, sharded_costs
    from tests.util_seq import gen_pair

    dist.init_process_group("gloo", rank=rank, world_size=world)
    pairs = [gen_pair(300 + 211 * i, 0.02 * (i % 6 + 1), seed=i) for i in range(11)]
    costs = sharded_costs(pairs)
    aligned = sharded_align(pairs)
    want = [oracle.levenshtein(a, b) for a, b in pairs]
    ok = costs == want and [c for c, _ in aligned] == want and all(oracle.cigar_verify(g, a, b) == c for (c, g), (a, b) in zip(aligned, pairs))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_gpu_sharded_two_ranks_one_device():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
