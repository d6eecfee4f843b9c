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

    out = sharded_costs(pairs, compute=compute)
    want = [oracle.levenshtein(a, b) for a, b in pairs]
    # the traceback variant gathers (cost, CIGAR): per-rank compute = the engine over the CPU oracle kernels
    from astar_pairwise_aligner_amd.sharding import sharded_align

    prm = oracle.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=True,
                             incremental_doubling=False, dt_trace=False)
    al = sharded_align(pairs[:9], compute=lambda sub: [oracle.cpu_align(a, b, prm)[:2] for a, b in sub])
    ok_al = all(c == w and oracle.cigar_verify(g, a, b) == c for (c, g), w, (a, b) in zip(al, want, pairs))
    q.put((rank, out == want and ok_al, calls))
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
    sizes = sorted(c[0] for _, _, c in res)
    assert sum(sizes) == 25 and sizes[0] >= 8  # both ranks got a real share


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
