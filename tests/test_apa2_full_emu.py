"""The flat per-pair program of the WHOLE A*PA2 family (csrc/apa2_full_logic.hpp: any heuristic behind h(i, j), incremental doubling
with the stored row of horizontal differences, pruning between blocks) over the CPU kernels -- groundwork for a batched `astarpa2_full`,
not yet run by the library.  Block columns live in per-block slots addressed by absolute word, as on the device.  Cost, CIGAR string
and twelve statistics must equal the host engine's (and, through tests/test_restated_engine.py, the second restatement's)."""
import random

import pytest

from oracle import astarpa2_restated as restated
from tests.test_restated_engine import variants
from tests.util_seq import PA_TEST_PAIRS, gen_pair, rand_seq

KEYS = ["num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "f_max_tries", "sanity_violations", "dt_trace_tries",
        "dt_trace_success", "dt_trace_fallback", "fill_tries", "fill_success", "fill_fallback"]
OUTSIDE = {"nw", "full_sparse", "full_sparse_dt", "gap_gap", "gap_start", "block64"}  # not Domain::Astar over sparse 256-column blocks


def compare(o, a, b, prm, tally=None):
    want = o.cpu_align(a, b, prm)
    rc, cost, cigar, stats, info = o.apa2_full_emu_align(a, b, prm)
    assert rc == 0, (rc, info)
    assert info[5] == 0, ("the flat GCSH probe disagrees with gcsh.hpp", info)
    assert info[7] == 0, ("the flat prune_block disagrees with gcsh.hpp", info)
    assert (cost, cigar) == want[:2], (len(a), len(b))
    assert {k: stats[k] for k in KEYS} == {k: want[2][k] for k in KEYS}, (len(a), len(b))
    if tally is not None:
        tally["prunes"] = tally.get("prunes", 0) + info[2]
        tally["three"] = tally.get("three", 0) + info[3]
        tally["tries"] = tally.get("tries", 0) + stats["f_max_tries"]
        tally["flat_builds"] = tally.get("flat_builds", 0) + info[6]
    return stats, info


def test_reference_harness_pairs_every_variant(oracle):
    for name, (prm, _) in variants(oracle).items():
        for a, b in PA_TEST_PAIRS:
            rc = oracle.apa2_full_emu_align(a, b, prm)[0]
            assert (rc == 1) == (name in OUTSIDE), name
            if rc != 1:
                compare(oracle, a, b, prm)


def test_random_pairs_every_field(oracle):
    vs = {k: v for k, v in variants(oracle).items() if k not in OUTSIDE}
    rng = random.Random(99)
    tally = {}
    for it in range(500):
        name = rng.choice(list(vs))
        n = rng.choice([rng.randint(1, 300), rng.randint(300, 2500), rng.randint(2500, 9000)])
        e = rng.choice([0.0, 0.01, 0.05, 0.1, 0.2, 0.4, 0.8])
        a, b = gen_pair(n, e, rng.randint(1, 10**9))
        mode = rng.random()
        if mode < 0.25 and n > 50:
            cut = rng.randint(0, len(b) - 1)
            ln = rng.randint(1, max(1, min(1500, len(b) // 2)))
            b = b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, it + 7) + b[cut:]
            b = b or b"A"
        elif mode < 0.3:
            b = rand_seq(rng.randint(1, n + 50), it + 9)
        compare(oracle, a, b, vs[name][0], tally)
    assert tally["prunes"] > 500 and tally["three"] > 100 and tally["tries"] > 800 and tally["flat_builds"] > 150, tally


@pytest.mark.parametrize("name", ["full", "gap_incr", "sh12_incr", "gcsh_k6_p3_prune_incr", "simple"])
def test_long_pairs_several_passes(oracle, name):
    prm, kw = variants(oracle)[name]
    for n, e, seed in [(20_000, 0.15, 4), (30_000, 0.08, 5), (12_000, 0.3, 6)]:
        a, b = gen_pair(n, e, seed)
        cut = len(b) // 3
        b = b[:cut] + rand_seq(700, seed + 1) + b[cut:2 * cut] + b[2 * cut + 400:]
        stats, info = compare(oracle, a, b, prm)
        assert stats["f_max_tries"] >= 2
    # ... and the second restatement on one of them directly
    got = restated.align(a, b, **kw)
    rc, cost, cigar, st, _ = oracle.apa2_full_emu_align(a, b, prm)
    assert (cost, cigar) == got[:2] and all(st[k] == got[2][k] for k in KEYS if k != "sanity_violations")


def test_c3_pair_full_preset(oracle):
    a, b = gen_pair(100_000, 0.05, seed=3_000_000)
    stats, info = compare(oracle, a, b, oracle.params_full())
    assert stats["f_max_tries"] == 1 and info[2] > 300  # one pass, matches pruned after (almost) every block


def test_one_pass_per_launch_orchestration(oracle, monkeypatch):
    """The search resumed from its saved state, one pass per step by a freshly built program object, the contours re-derived outside
    the program between two steps (what a kernel launch per pass with host threads in between would do): same results."""
    monkeypatch.setenv("PA_FULL_EMU_STEPWISE", "1")
    vs = {k: v for k, v in variants(oracle).items() if k not in OUTSIDE}
    rng = random.Random(5)
    tries = 0
    for it in range(150):
        name = rng.choice(["full", "gcsh_k6_p3_prune_incr", "gcsh_k8_p0_prune", "gap_incr", "simple", "linear300", "gcsh_k10_p5_nosparseh"])
        n = rng.choice([rng.randint(1, 300), rng.randint(300, 2500), rng.randint(2500, 9000)])
        a, b = gen_pair(n, rng.choice([0.0, 0.02, 0.1, 0.2, 0.4]), rng.randint(1, 10**9))
        if rng.random() < 0.3 and n > 50:
            cut = rng.randint(0, len(b) - 1)
            b = (b[:cut] + b[cut + rng.randint(1, max(1, len(b) // 3)):]) or b"A"
        stats, _ = compare(oracle, a, b, vs[name][0])
        tries += stats["f_max_tries"]
    assert tries > 180  # (many of the pairs took more than one step)
