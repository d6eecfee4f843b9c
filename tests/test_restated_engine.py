"""Two restatements of the reference's traced A*PA2 host logic against each other, on the CPU.

`oracle.cpu_align` is the product's host engine (csrc/engine.hpp) over the CPU oracle kernels; `oracle/astarpa2_restated.py` is a second
restatement written from the Rust text alone, in pure Python on big integers, sharing no code with the first.  For every pair the cost,
the CIGAR STRING and eleven statistics must be identical -- band doubling, ranges, block reuse, the stale-block quirks, DT-trace with its
x-drop, the re-fill fallback and the parent rules included.  (The twelfth statistic, sanity_violations, is this build's own.)"""
import random

import pytest

from oracle import astarpa2_restated as restated
from tests.util_seq import PA_TEST_ES, PA_TEST_NS, PA_TEST_PAIRS, gen_pair, rand_seq

KEYS = ["num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "f_max_tries", "dt_trace_tries", "dt_trace_success",
        "dt_trace_fallback", "fill_tries", "fill_success", "fill_fallback"]

BASE = dict(domain="astar", doubling="band", start="h0", factor=2.0, block_width=256, sparse=True, incremental_doubling=False,
            dt_trace=True, max_g=40, fr_drop=10, sparse_h=True)


def variants(o):
    """name -> (parameters of the engine, keyword arguments of the restatement)"""
    mk = lambda **kw: o.make_params(**{**BASE, **kw})
    return {
        "simple": (o.params_simple(), dict(heuristic="gap")),
        "dijkstra": (mk(heuristic="none"), dict(heuristic="none")),
        "sh12": (mk(heuristic="sh", k=12), dict(heuristic="sh", k=12)),
        "sh5": (mk(heuristic="sh", k=5), dict(heuristic="sh", k=5)),
        "gap_nosparseh": (mk(heuristic="gap", sparse_h=False), dict(heuristic="gap", sparse_h=False)),
        "gap_nodt": (mk(heuristic="gap", dt_trace=False), dict(heuristic="gap", dt_trace=False)),
        "gap_g20_drop5": (mk(heuristic="gap", max_g=20, fr_drop=5), dict(heuristic="gap", max_g=20, fr_drop=5)),
        "gap_nodrop": (mk(heuristic="gap", fr_drop=0), dict(heuristic="gap", fr_drop=0)),
        "gap_startgap": (mk(heuristic="gap", start="gap"), dict(heuristic="gap", start="gap")),
        "gap_startzero_f15": (mk(heuristic="gap", start="zero", factor=1.5), dict(heuristic="gap", start="zero", factor=1.5)),
        "linear300": (mk(heuristic="gap", doubling="linear", delta=300.0), dict(heuristic="gap", doubling="linear", delta=300.0)),
        "block64": (mk(heuristic="gap", block_width=64), dict(heuristic="gap", block_width=64)),
        "gap_incr": (mk(heuristic="gap", incremental_doubling=True), dict(heuristic="gap", incremental_doubling=True)),
        "sh12_incr": (mk(heuristic="sh", k=12, incremental_doubling=True), dict(heuristic="sh", k=12, incremental_doubling=True)),
        "dijkstra_incr_nodt": (mk(heuristic="none", incremental_doubling=True, dt_trace=False),
                               dict(heuristic="none", incremental_doubling=True, dt_trace=False)),
        "gap_incr_f15": (mk(heuristic="gap", incremental_doubling=True, start="zero", factor=1.5),
                         dict(heuristic="gap", incremental_doubling=True, start="zero", factor=1.5)),
        "full": (o.params_full(), dict(heuristic="gcsh", k=12, p=14, prune=True, incremental_doubling=True)),
        "gcsh_noprune": (mk(heuristic="gcsh", k=12, p=14, incremental_doubling=True), dict(heuristic="gcsh", k=12, p=14, incremental_doubling=True)),
        "gcsh_k8_p0_prune": (mk(heuristic="gcsh", k=8, p=0, prune=True), dict(heuristic="gcsh", k=8, p=0, prune=True)),
        "gcsh_k6_p3_prune_incr": (mk(heuristic="gcsh", k=6, p=3, prune=True, incremental_doubling=True),
                                  dict(heuristic="gcsh", k=6, p=3, prune=True, incremental_doubling=True)),
        "gcsh_k10_p5_nosparseh": (mk(heuristic="gcsh", k=10, p=5, prune=True, sparse_h=False, dt_trace=False),
                                  dict(heuristic="gcsh", k=10, p=5, prune=True, sparse_h=False, dt_trace=False)),
        "gap_gap": (mk(domain="gap_gap", heuristic="none", start="gap"), dict(domain="gap_gap", start="gap")),
        "gap_start": (mk(domain="gap_start", heuristic="none", start="zero"), dict(domain="gap_start", start="zero")),
        "nw": (o.params_nw(), dict(domain="full", doubling="none", sparse=False, dt_trace=False)),
        "full_sparse": (o.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=True, incremental_doubling=False,
                                      dt_trace=False), dict(domain="full", doubling="none", sparse=True, dt_trace=False)),
        "full_sparse_dt": (o.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=True,
                                         incremental_doubling=False, dt_trace=True, max_g=40, fr_drop=10),
                           dict(domain="full", doubling="none", sparse=True, dt_trace=True)),
    }


def compare(o, a, b, prm, kw, tally=None):
    want = o.cpu_align(a, b, prm)
    got = restated.align(a, b, **kw)
    assert got[0] == want[0], (len(a), len(b), kw)
    assert got[1] == want[1], (len(a), len(b), kw, got[1][:60], want[1][:60])
    assert {k: got[2][k] for k in KEYS} == {k: want[2][k] for k in KEYS}, (len(a), len(b), kw)
    if tally is not None:
        for k in KEYS:
            tally[k] = tally.get(k, 0) + got[2][k]
        tally["regrown"] = tally.get("regrown", 0) + (got[2]["f_max_tries"] > 1 and got[2]["unique_lanes"] < got[2]["computed_lanes"])
    return got


def test_the_reference_harness_pairs_and_grid(oracle):
    """pa-test/src/lib.rs:7-40: the literal pairs and the (n, e) grid, every variant."""
    vs = variants(oracle)
    for name, (prm, kw) in vs.items():
        for a, b in PA_TEST_PAIRS:
            compare(oracle, a, b, prm, kw)
    rng = random.Random(5)
    for n in PA_TEST_NS:
        for e in PA_TEST_ES:
            if n == 0:
                continue
            a, b = gen_pair(n, e, seed=n * 131 + int(e * 1000))
            name = rng.choice(list(vs))
            compare(oracle, a, b, *vs[name])


def test_random_pairs_every_field(oracle):
    vs = variants(oracle)
    rng = random.Random(20260927)
    tally = {}
    for it in range(700):
        name = rng.choice(list(vs))
        n = rng.choice([rng.randint(1, 300), rng.randint(300, 1500), rng.randint(1500, 6000)])
        e = rng.choice([0.0, 0.01, 0.05, 0.1, 0.2, 0.4, 0.8])
        a, b = gen_pair(n, e, rng.randint(1, 10**9))
        mode = rng.random()
        if mode < 0.25 and n > 50:  # one long indel: band edges move, DT-trace gives up, re-fills grow
            cut = rng.randint(0, len(b) - 1)
            ln = rng.randint(1, max(1, min(800, len(b) // 2)))
            b = b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, it + 7) + b[cut:]
            b = b or b"A"
        elif mode < 0.3:
            b = rand_seq(rng.randint(1, n + 50), it + 9)
        compare(oracle, a, b, *vs[name], tally=tally)
    # the comparison has to have been through the interesting paths, not only the straight ones
    assert tally["dt_trace_success"] > 500 and tally["dt_trace_fallback"] > 100
    assert tally["fill_success"] > 100 and tally["fill_fallback"] > 20
    assert tally["f_max_tries"] > 900 and tally["regrown"] > 50  # (regrown: pairs whose later passes recomputed blocks over wider ranges)


@pytest.mark.parametrize("name", ["simple", "sh12", "dijkstra", "gap_nodt", "gap_incr", "sh12_incr", "gap_incr_f15", "full", "gcsh_k8_p0_prune"])
def test_long_pairs_several_passes(oracle, name):
    prm, kw = variants(oracle)[name]
    for n, e, seed in [(20_000, 0.15, 4), (30_000, 0.08, 5), (12_000, 0.3, 6)]:
        a, b = gen_pair(n, e, seed)
        cut = len(b) // 3
        b = b[:cut] + rand_seq(700, seed + 1) + b[cut:2 * cut] + b[2 * cut + 400:]
        got = compare(oracle, a, b, prm, kw)
        assert got[2]["f_max_tries"] >= 3


def test_c3_pair_of_the_bench(oracle):
    """The 100 kbp pair at 5 % of bench.py's c3 legs, `simple`: cost, CIGAR string and statistics."""
    a, b = gen_pair(100_000, 0.05, seed=3_000_000)
    got = compare(oracle, a, b, *variants(oracle)["simple"])
    assert got[2]["f_max_tries"] == 6 and got[2]["dt_trace_tries"] == 391


def test_gcsh_matches_local_pruning_and_h_values(oracle):
    """GCSH by its definition (restated.Gcsh, numpy) against the engine's (csrc/gcsh.hpp over oracle/engine_cpu.cpp): the matches that
    survive the transform filter and local pruning, and h at random positions before any pruning."""
    rnd = random.Random(3)
    for n, e, k, p, seed in [(3000, 0.08, 6, 5, 7), (5000, 0.05, 12, 14, 8), (2000, 0.2, 5, 3, 9), (4000, 0.1, 8, 14, 10), (6000, 0.15, 10, 2, 11),
                             (1500, 0.02, 4, 0, 12)]:
        a, b = gen_pair(n, e, seed)
        g = restated.Gcsh(a, b, k, p, True)
        q = [(rnd.randrange(len(a) + 1), rnd.randrange(len(b) + 1)) for _ in range(300)] + [(0, 0), (len(a), len(b))]
        want_h, want_kept = oracle.gcsh_probe(a, b, k, p, q)
        assert sorted(map(tuple, want_kept)) == sorted(zip(g.mi.tolist(), g.mj.tolist()))
        assert [g.h(i, j) for i, j in q] == want_h
        if p:
            assert len(g.mi) < len(restated.Gcsh(a, b, k, 0, True).mi)  # (local pruning removed something)


def long_kmer_collision_pair(k: int = 20, seeds: int = 60, seed: int = 5, group: int = 4):
    """Seeds of a that share their LAST 16 characters (in groups of `group`) and differ in the first k - 16.  The reference keys its match
    table on `q as u32` (pa-heuristic/src/matches/exact.rs:47 `type Key = u32`, :53 `h.entry(q as Key)`, :56 `h.get(&(q as Key))`) and
    the q-gram has its first character in the HIGH bits (qgrams.rs:36-43), so for k > 16 only the last 16 characters are compared: all
    seeds of a group match every k-mer of b that ends in the group's tail.  b holds half of the seeds verbatim (shuffled) and then, per
    group, fresh heads in front of the tail -- k-mers that equal NO seed and still match `group` seeds each in the reference."""
    rnd = random.Random(seed)
    ngroups = (seeds + group - 1) // group
    tails = [bytes(rnd.choice(b"ACGT") for _ in range(16)) for _ in range(ngroups)]
    heads = [bytes(rnd.choice(b"ACGT") for _ in range(k - 16)) for _ in range(seeds)]
    a = b"".join(heads[s] + tails[s // group] for s in range(seeds))
    order = list(range(seeds))
    rnd.shuffle(order)
    b = b"".join(heads[s] + tails[s // group] for s in order[: seeds // 2])
    b += b"".join(bytes(rnd.choice(b"ACGT") for _ in range(k - 16)) + tails[g % ngroups] for g in range(seeds // 2))
    return a, b


def test_gcsh_kmers_longer_than_16_match_on_their_last_16_characters(oracle):
    """k > 16: the reference's u32 key (exact.rs:47,53,56) -- both restatements must keep the collisions as matches."""
    for k, p in ((20, 0), (24, 3), (31, 0), (17, 14)):
        a, b = long_kmer_collision_pair(k)
        g = restated.Gcsh(a, b, k, p, True)
        q = [(0, 0), (len(a), len(b)), (len(a) // 2, len(b) // 3)]
        want_h, want_kept = oracle.gcsh_probe(a, b, k, p, q)
        kept = sorted(map(tuple, want_kept))
        assert kept == sorted(zip(g.mi.tolist(), g.mj.tolist())), (k, p)
        assert [g.h(i, j) for i, j in q] == want_h
        if p == 0:  # by the definition, from the characters: equal LAST 16 characters + the transform filter T(start) <= T(target)
            tt = g.t_target
            brute = set()
            for i in range(0, len(a) - k + 1, k):
                for j in range(len(b) - k + 1):
                    if a[i + k - 16:i + k] == b[j + k - 16:j + k]:
                        t = g.T(i, j)
                        if t[0] <= tt[0] and t[1] <= tt[1]:
                            brute.add((i, j))
            assert set(kept) == brute, (k, len(kept), len(brute))
            collisions = [(i, j) for i, j in kept if a[i:i + k] != b[j:j + k]]
            assert len(collisions) > len(kept) // 2, (k, len(collisions), len(kept))  # most matches here are not equal k-mers at all
    # SH: a seed counts as matched as soon as ANY k-mer of b ends in its last 16 characters
    a, b = long_kmer_collision_pair(20)
    b2 = b[20 * 30:]  # only the fresh-head half: (nearly) no k-mer of b2 equals a seed of a, every tail of a occurs
    assert sum(a[i:i + 20] not in b2 for i in range(0, len(a), 20)) >= 50  # (whole k-mers compared: >= 50 of 60 seeds unmatched)
    assert restated.sh_table(a, b2, 20)[0] == 0  # u32 key: all sixty seeds matched
    compare(oracle, a, b2, oracle.make_params(**{**BASE, "heuristic": "sh", "k": 20}), dict(heuristic="sh", k=20))
    compare(oracle, a, b2, oracle.make_params(**{**BASE, "heuristic": "gcsh", "k": 20, "p": 3, "prune": True}), dict(heuristic="gcsh", k=20, p=3, prune=True))
    compare(oracle, a, b, oracle.make_params(**{**BASE, "heuristic": "gcsh", "k": 24, "p": 0, "prune": True, "incremental_doubling": True}),
            dict(heuristic="gcsh", k=24, p=0, prune=True, incremental_doubling=True))


def test_c3_pair_of_the_bench_full_preset(oracle):
    """The 100 kbp pair of the bench through `full` (GCSH k = 12, p = 14, pruning at the start, incremental doubling)."""
    a, b = gen_pair(100_000, 0.05, seed=3_000_000)
    prm, kw = variants(oracle)["full"]
    r = restated.Restated(a, b, **kw)
    got = r.align()
    want = oracle.cpu_align(a, b, prm)
    assert (got[0], got[1]) == want[:2] and {k: got[2][k] for k in KEYS} == {k: want[2][k] for k in KEYS}
    assert int((~r.gcsh.active).sum()) > 1000  # matches were pruned on the way


def test_sh_with_local_pruning(oracle):
    """HeuristicParams.p reaches SH as well (pa-heuristic/src/cli.rs:168-180 sets MatchConfig.local_pruning for every heuristic; sh.rs:48):
    a seed counts as matched only if one of its matches survives MatchBuilder::push's local pruning (matches.rs:205-247).  Both
    restatements, cost + CIGAR + statistics; the pruning has to change the table on a good share of the pairs."""
    rng = random.Random(1)
    changed = 0
    for it in range(80):
        n, e = rng.choice([200, 1500, 6000]), rng.choice([0.02, 0.1, 0.25])
        k, p = rng.choice([5, 8, 12]), rng.choice([1, 3, 14])
        a, b = gen_pair(n, e, rng.randint(1, 10**9))
        incr = rng.random() < 0.3
        compare(oracle, a, b, oracle.make_params(**{**BASE, "heuristic": "sh", "k": k, "p": p, "incremental_doubling": incr}),
                dict(heuristic="sh", k=k, p=p, incremental_doubling=incr))
        changed += restated.sh_table(a, b, k, p) != restated.sh_table(a, b, k, 0)
    assert changed >= 20, changed
