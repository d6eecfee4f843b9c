"""The per-pair A*PA2 program of the batched mode (csrc/apa2_logic.hpp: ONE wavefront runs a pair's whole band search --
cost_or_align, the doubling loop, every align_for_bounded_dist pass) run WITHOUT a GPU over the oracle's CPU kernels
(oracle/apa2_emu.cpp).  It must reproduce the host-driven engine exactly: cost, CIGAR string (Blocks::trace over the blocks the
program leaves behind) and all twelve statistics; and the reference's jumping probes of fixed_j_range (domain.rs:306-328) must end
on the first / last row with f <= f_max, which is what the device backend's wave-parallel scans compute."""
import random

import pytest

from tests.test_sweep_emu import KEYS, variants
from tests.util_seq import PA_TEST_PAIRS, gen_pair, rand_seq


def both(o, a, b, prm):
    want = o.cpu_align(a, b, prm, trace=True)
    rc, cost, cigar, stats, info = o.apa2_emu_align(a, b, prm)
    assert rc == 0, (rc, info)
    assert info[2] == 0, f"{info[2]} of {info[1]} scans did not end on the first / last row with f <= f_max"
    assert cost == want[0]
    assert cigar == want[1]
    assert {k: stats[k] for k in KEYS} == {k: want[2][k] for k in KEYS}
    return cost


def test_block_boundary_sizes(oracle):
    prm = oracle.params_simple()
    for n in (1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 2047, 2048, 2049, 4095, 4096, 4097, 8191):
        for e in (0.0, 0.05, 0.4, 1.0):
            a, b = gen_pair(n, e, seed=n * 7 + int(e * 100))
            assert both(oracle, a, b, prm) == oracle.levenshtein(a, b)


def test_pa_test_pairs_all_variants(oracle):
    for name, prm in variants(oracle).items():
        for a, b in PA_TEST_PAIRS:
            if a and b:
                both(oracle, a, b, prm)


def test_unsupported_or_degenerate_inputs_are_reported(oracle):
    a, b = gen_pair(500, 0.1, 1)
    assert oracle.apa2_emu_align(a, b, oracle.params_full())[0] == 1
    assert oracle.apa2_emu_align(a, b, oracle.params_nw())[0] == 1
    assert oracle.apa2_emu_align(b"", b, oracle.params_simple())[0] == 1


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_pairs_all_variants(oracle, seed):
    rng = random.Random(seed)
    vs = variants(oracle)
    for _ in range(70):
        name = rng.choice(list(vs))
        n = rng.choice([rng.randint(1, 600), rng.randint(600, 4000), rng.randint(4000, 14000)])
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
        both(oracle, a, b, vs[name])


def test_linear_search_many_passes(oracle):
    """LinearSearch with a small delta: hundreds of passes, every block reused or recomputed pass after pass."""
    o = oracle
    prm = o.make_params(domain="astar", heuristic="gap", doubling="linear", start="h0", delta=3.0, block_width=256, sparse=True,
                        incremental_doubling=False, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True)
    for n, e, s in [(700, 0.2, 1), (1500, 0.1, 2), (3000, 0.05, 3)]:
        a, b = gen_pair(n, e, s)
        both(o, a, b, prm)
