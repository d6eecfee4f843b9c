"""GPU parity of the device-side A*PA2 sweep (csrc/sweep_wave.hpp + sweep_kernel.hpp): pa_align with Domain::Astar over
NoCost / GapCost / SH and the sparse, non-incremental block engine runs every align_for_bounded_dist pass as ONE persistent
launch with the band logic (domain.rs:117-350) in the kernel.  Cost, CIGAR string and every band statistic must equal the
host-driven engine over the CPU oracle kernels, and the host-driven engine over the HIP kernels (PA_ENGINE_NO_SWEEP)."""
import os
import random
import time

import pytest

from tests.test_sweep_emu import KEYS, variants
from tests.util_seq import gen_pair, rand_seq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import astar_pairwise_aligner_amd as pa

    pa.require_gpu()
    return pa


def both(pa, oracle, a, b, oc, trace=True):
    from tests.test_gpu_engine import gpu_params

    want = oracle.cpu_align(a, b, oc, trace=trace)
    cost, cigar, stats = gpu_params(pa, oc).make_aligner(trace).align_with_stats(a, b)
    if trace:
        assert cost == want[0]
        assert cigar == want[1]
        assert {k: stats[k] for k in KEYS} == {k: want[2][k] for k in KEYS}
    else:
        # The reference's cost-only path keeps ONE block whose fixed range only grows (blocks.rs:245-270) and can end on an upper
        # bound (tests/tools/fuzz_sweep.py found SH returning 11353 for a distance of 11325); the sweep always runs the traced
        # band: its cost-only answer is the distance itself.
        assert cigar is None
        assert cost == oracle.nw_cost(a, b, True) <= want[0]
    return cost, cigar, stats


def test_block_boundary_sizes(pa, oracle):
    prm = oracle.params_simple()
    for n in (1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 2047, 2048, 2049, 4095, 4096, 4097, 8191):
        for e in (0.0, 0.05, 0.4, 1.0):
            a, b = gen_pair(n, e, seed=n * 7 + int(e * 100))
            cost, _, _ = both(pa, oracle, a, b, prm)
            assert cost == oracle.levenshtein(a, b)


@pytest.mark.parametrize("name", ["simple", "dijkstra", "sh12", "sh5", "gap_nosparseh", "gap_nodt", "gap_startgap", "gap_startzero_f15", "linear"])
def test_variants_multi_strip(pa, oracle, name):
    """Several passes, bands of several 2048-row strips, both edges crossing strip boundaries."""
    oc = variants(oracle)[name]
    for n, e, seed in [(300, 0.05, 1), (3000, 0.1, 3), (10000, 0.15, 4), (30000, 0.2, 6), (20000, 0.3, 5)]:
        a, b = gen_pair(n, e, seed)
        cost, cigar, _ = both(pa, oracle, a, b, oc)
        assert oracle.cigar_verify(cigar, a, b) == cost


def test_random_pairs_all_variants(pa, oracle):
    rng = random.Random(7)
    vs = variants(oracle)
    for _ in range(150):
        name = rng.choice(list(vs))
        n = rng.choice([rng.randint(1, 600), rng.randint(600, 4000), rng.randint(4000, 40000)])
        e = rng.choice([0.0, 0.01, 0.05, 0.1, 0.2, 0.4, 0.8])
        s = rng.randint(1, 10**6)
        a, b = gen_pair(n, e, s)
        mode = rng.random()
        if mode < 0.25 and n > 50:  # a long indel
            cut = rng.randint(0, len(b) - 1)
            ln = rng.randint(1, max(1, min(3000, len(b) // 2)))
            b = b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, s + 1) + b[cut:]
            b = b or b"A"
        elif mode < 0.3:
            b = rand_seq(rng.randint(1, n + 50), s + 2)  # unrelated
        both(pa, oracle, a, b, vs[name], trace=rng.random() < 0.8)


def test_sweep_equals_host_driven_engine(pa, oracle):
    """The same library with the sweep switched off (one launch per block, band logic on the host) gives the same answers."""
    from tests.test_gpu_engine import gpu_params

    pairs = [gen_pair(n, e, s) for n, e, s in [(5000, 0.1, 11), (30000, 0.05, 12), (12000, 0.3, 13)]]
    al = gpu_params(pa, oracle.params_simple()).make_aligner(True)
    got = [al.align_with_stats(a, b) for a, b in pairs]
    os.environ["PA_ENGINE_NO_SWEEP"] = "1"
    try:
        ref = [al.align_with_stats(a, b) for a, b in pairs]
    finally:
        del os.environ["PA_ENGINE_NO_SWEEP"]
    for (c1, g1, s1), (c2, g2, s2) in zip(got, ref):
        assert (c1, g1) == (c2, g2)
        assert {k: s1[k] for k in KEYS} == {k: s2[k] for k in KEYS}


def test_c3_simple_100kbp(pa, oracle):
    """BASELINE C3 (`simple`): 100 kbp, 5 %, traceback on: cost, CIGAR and statistics equal the CPU-kernel engine."""
    a, b = gen_pair(100_000, 0.05, 1)
    cost, cigar, stats = both(pa, oracle, a, b, oracle.params_simple())
    assert oracle.cigar_verify(cigar, a, b) == cost == 4810
    assert stats["sanity_violations"] == 0


def test_1mbp_with_traceback(pa, oracle):
    from tests.test_gpu_engine import gpu_params

    a, b = gen_pair(1_000_000, 0.05, 1)
    cost, cigar, stats = gpu_params(pa, oracle.params_simple()).make_aligner(True).align_with_stats(a, b)
    assert oracle.cigar_verify(cigar, a, b) == cost
    costs, _ = pa.Batch([(a, b)]).run()  # the full matrix on the batch kernel
    assert int(costs[0]) == cost
    assert stats["sanity_violations"] == 0 and stats["f_max_tries"] == 9


def test_c5_10mbp_doubling_band(pa, oracle):
    """BASELINE C5: one 10 Mbp x 10 Mbp pair, 5 %, A*PA2 `simple` (GapCost band doubling), traceback off, through pa_align.
    Cost and block statistics equal tests/golden/c5.json -- the host engine over the CPU oracle kernels in traced mode (seven minutes
    of one core, tests/golden/make_c5.py): the sweep computes the traced band also when no CIGAR is asked for (include/pa_astarpa2.h).
    The cost also equals the GapGap-banded batch (an independent kernel and band rule).  (Wall-clock bounds live in bench.py.)"""
    import json
    from pathlib import Path

    from tests.test_gpu_engine import gpu_params

    gold = json.loads((Path(__file__).resolve().parent / "golden" / "c5.json").read_text())["trace_on"]
    a, b = gen_pair(10_000_000, 0.05, 1)
    al = gpu_params(pa, oracle.params_simple()).make_aligner(False)
    cost, cigar, stats = al.align_with_stats(a, b)
    assert cigar is None
    assert cost == gold["cost"] == 480_033
    for k in ("f_max_tries", "num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "sanity_violations"):
        assert stats[k] == gold[k], k
    costs, _ = pa.Batch([(a, b)], band=0.05).run()
    assert int(costs[0]) == cost


def test_c5_10mbp_with_traceback(pa, oracle):
    """C5 with the traceback on: cost, all twelve statistics and the CIGAR (length and SHA-256) of tests/golden/c5.json."""
    import hashlib
    import json
    from pathlib import Path

    from tests.test_gpu_engine import gpu_params

    gold = json.loads((Path(__file__).resolve().parent / "golden" / "c5.json").read_text())["trace_on"]
    a, b = gen_pair(10_000_000, 0.05, 1)
    cost, cigar, stats = gpu_params(pa, oracle.params_simple()).make_aligner(True).align_with_stats(a, b)
    assert cost == gold["cost"]
    assert {k: stats[k] for k in KEYS} == {k: gold[k] for k in KEYS}
    assert len(cigar) == gold["cigar_len"] and hashlib.sha256(cigar.encode()).hexdigest() == gold["cigar_sha256"]
    pa.capi.load().pa_release_pools()  # (gigabytes of pooled device buffers)


def test_linear_search_more_passes_than_tag_bits(pa, oracle):
    """LinearSearch with delta = 1 on a divergent pair needs more passes than the 12-bit pass id of the sweep's tagged words holds
    (plus the speculative launches): the sweep must hand the pair to the host-driven engine instead of reading another pass's
    records through aliased tags.  Cost, CIGAR and statistics as ever."""
    oc = oracle.make_params(domain="astar", heuristic="gap", doubling="linear", start="h0", delta=1.0, block_width=256, sparse=True,
                            incremental_doubling=False, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True)
    a, b = gen_pair(14000, 0.45, 11)
    cost, cigar, stats = both(pa, oracle, a, b, oc)
    assert stats["f_max_tries"] > 4100
    a, b = gen_pair(3000, 0.1, 12)  # and the pool is usable afterwards
    both(pa, oracle, a, b, oracle.params_simple())


@pytest.mark.parametrize("k", ["1", "2", "3"])
def test_giving_up_speculative_passes_changes_nothing(pa, oracle, monkeypatch, k):
    """tests/test_sweep_emu.py::test_giving_up_speculative_passes_changes_nothing on the device: after every k-th pass the
    passes launched ahead are cancelled and launched again from the real state (streams, cancel words, slots, merged records)."""
    monkeypatch.setenv("PA_SWEEP_TEST_GIVE_UP", k)
    rng = random.Random(int(k))
    vs = variants(oracle)
    for _ in range(12):
        name = rng.choice(list(vs))
        n = rng.choice([rng.randint(500, 4000), rng.randint(4000, 40000)])
        a, b = gen_pair(n, rng.choice([0.05, 0.15, 0.3, 0.5]), rng.randint(1, 10**6))
        both(pa, oracle, a, b, vs[name], trace=rng.random() < 0.8)
    a, b = gen_pair(100_000, 0.05, 1)
    cost, cigar, stats = both(pa, oracle, a, b, oracle.params_simple())
    assert cost == 4810
