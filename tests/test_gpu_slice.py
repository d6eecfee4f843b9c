"""The bit-sliced full-DP kernel (csrc/slice_kernel.hpp: groups of 32 pairs, one register per DP row) against the oracle.

Big cost-only batches run this way (pa_batch_slice_info says so); the distances must be those of the reference's bit-parallel DP
(pa-bitpacking/src/myers.rs:27-55 through simd::compute, restated in oracle/pa_oracle.c) bit for bit -- ragged lengths inside a group
(capture events), groups that are not full, empty sequences, every instantiated number of rows per lane, chains of strips longer than the
chip has wave slots, and the same batch through the strip kernels (PA_SLICE=0)."""
import random

import numpy as np
import pytest

from tests.util_seq import gen_pair, mutate, rand_seq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import astar_pairwise_aligner_amd as pa

    pa.require_gpu()
    return pa


def ragged_pairs(count, seed, max_len=6000):
    rng = random.Random(seed)
    pairs = []
    for i in range(count):
        n = rng.choice([rng.randint(1, 40), rng.randint(40, 700), rng.randint(700, max_len)])
        a, b = gen_pair(n, rng.choice([0.0, 0.02, 0.1, 0.3]), rng.randint(1, 10**9))
        mode = rng.random()
        if mode < 0.15:
            b = rand_seq(rng.randint(1, n + 300), rng.randint(1, 10**9))  # unrelated, any length
        elif mode < 0.25:
            a, b = b, a
        pairs.append((a, b or b"A"))
    return pairs


@pytest.mark.parametrize("rows", [1, 28, 32, 36, 40, 42, 44, 46, 48, 50, 52])
def test_ragged_batch_equals_the_oracle(pa, oracle, monkeypatch, rows):
    """PA_SLICE=1: the library's own choice of rows per lane; 28 .. 52: each instantiation forced."""
    monkeypatch.setenv("PA_SLICE", str(rows))
    pairs = ragged_pairs(300 if rows == 1 else 150, seed=rows)
    pairs[7] = (b"", b"ACGT")  # empty sequences are in no group
    pairs[19] = (b"ACGTT", b"")
    pairs[23] = (b"", b"")
    bt = pa.Batch(pairs)
    sh = bt.shape()
    assert sh.get("sliced_rows_per_lane", 0) == (rows if rows > 1 else sh.get("sliced_rows_per_lane")) and sh["kernel"].startswith("pa::slice::slice_kernel")
    assert sh["groups"] == (len(pairs) - 3 + 31) // 32
    costs, ms = bt.run()
    want = [oracle.levenshtein(a, b) for a, b in pairs]
    assert costs.tolist() == want
    costs2, _ = bt.run()  # a second pass over the resident batch: boundary rows and captured columns are reset
    assert np.array_equal(costs, costs2)
    bt.close()


def test_same_costs_as_the_strip_kernels(pa, monkeypatch):
    """The batch through both families of kernels: pair_kernel / strip_kernel (PA_SLICE=0) and the bit-sliced one."""
    pairs = ragged_pairs(500, seed=77, max_len=20_000)
    monkeypatch.setenv("PA_SLICE", "0")
    b0 = pa.Batch(pairs)
    assert "sliced_rows_per_lane" not in b0.shape()
    c0, _ = b0.run()
    b0.close()
    monkeypatch.setenv("PA_SLICE", "1")
    b1 = pa.Batch(pairs)
    assert b1.shape()["sliced_rows_per_lane"] in (28, 32, 36, 40, 42, 44, 46, 48, 50, 52)
    c1, _ = b1.run()
    b1.close()
    assert np.array_equal(c0, c1)


def test_long_pairs_many_strips_per_group(pa, oracle, monkeypatch):
    """Groups whose strips chain through the boundary rows (polled values), more (group, strip) jobs than the chip has wave slots:
    96 pairs of 30-40 kbp at 28 rows per lane = 3 groups x 23 strips, and 2 groups of 100 kbp pairs at the default rows."""
    rng = random.Random(5)
    pairs = [gen_pair(rng.randint(30_000, 40_000), rng.choice([0.01, 0.05, 0.15]), seed=1000 + i) for i in range(96)]
    monkeypatch.setenv("PA_SLICE", "28")
    bt = pa.Batch(pairs)
    assert bt.shape()["jobs"] >= 50
    costs, _ = bt.run()
    bt.close()
    assert costs.tolist() == [oracle.nw_cost(a, b, True) for a, b in pairs]
    monkeypatch.setenv("PA_SLICE", "1")
    long_pairs = [gen_pair(100_000, 0.05, seed=s) for s in range(1, 41)]
    bt = pa.Batch(long_pairs)
    costs, _ = bt.run()
    bt.close()
    sample = [0, 1, 2, 17, 33, 39]
    assert [int(costs[i]) for i in sample] == [oracle.nw_cost(*long_pairs[i], True) for i in sample]
    # symmetry and identity through the sliced kernel (d(a, b) = d(b, a): rows and columns swap roles, other groups, other strips)
    bt = pa.Batch([(b, a) for a, b in long_pairs] + [(long_pairs[0][0], long_pairs[0][0])] * 24)
    back, _ = bt.run()
    bt.close()
    assert back[:40].tolist() == costs.tolist() and not back[40:].any()


def test_the_default_picks_the_sliced_kernel_for_a_big_batch_only(pa, monkeypatch):
    monkeypatch.delenv("PA_SLICE", raising=False)
    few = pa.Batch([gen_pair(3000, 0.05, seed=i) for i in range(1, 9)])
    assert "sliced_rows_per_lane" not in few.shape()
    few.close()
    a, b = gen_pair(10_000, 0.05, seed=3)
    many = pa.Batch([(a, mutate(b, 0.01, seed=i)) for i in range(1, 10_001)])
    sh = many.shape()
    assert sh.get("sliced_rows_per_lane", 0) > 0, sh
    costs, _ = many.run()
    many.close()
    monkeypatch.setenv("PA_SLICE", "0")
    ref = pa.Batch([(a, mutate(b, 0.01, seed=i)) for i in range(1, 10_001)])
    want, _ = ref.run()
    ref.close()
    assert np.array_equal(costs, want)


def test_invalid_base_is_rejected(pa, monkeypatch):
    monkeypatch.setenv("PA_SLICE", "1")
    pairs = [gen_pair(500, 0.05, seed=i) for i in range(1, 70)]
    pairs[40] = (pairs[40][0], b"ACGNACGT")
    bt = pa.Batch(pairs)
    with pytest.raises(Exception, match="outside ACGT"):
        bt.run()
    bt.close()
