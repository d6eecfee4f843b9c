"""Host block engine (engine.hpp) over the CPU oracle kernels, checked with the reference's own
acceptance rules (pa-test/src/lib.rs:65-99): cost == Levenshtein, CIGAR valid & cost-consistent,
second align gives the same cost; plus the cfg!(test) incremental-doubling self check (blocks.rs:471-543).
The 9 configurations are those of astarpa2/src/tests.rs:6-119 (SH/GCSH ones mapped to GapCost until the
seed heuristics are restated)."""
import pytest

from tests.util_seq import PA_TEST_ES, PA_TEST_NS, PA_TEST_PAIRS, gen_pair, mutate, rand_seq


def configs(o):
    nw = dict(doubling="none", domain="full", heuristic="none", block_width=1, sparse_h=True, prune=True)  # tests.rs:6-17
    band = dict(doubling="band", start="gap")  # DoublingType::band_doubling(), band.rs:48-55
    return {
        "full": o.make_params(**nw),
        "band_doubling_gapgap": o.make_params(**{**nw, **band, "domain": "gap_gap", "block_width": 64}),
        "dt_trace_gapgap": o.make_params(**{**nw, **band, "domain": "gap_gap", "block_width": 256, "dt_trace": True}),
        "band_doubling_dijkstra": o.make_params(**{**nw, **band, "domain": "astar", "heuristic": "none", "block_width": 64}),
        "band_doubling_edlib": o.make_params(**{**nw, **band, "domain": "astar", "heuristic": "gap", "block_width": 64}),
        # tests.rs:68-79 runs block_width 1 with incremental doubling; see test_known_quirk_* below for why it is off here
        "band_doubling_w1": o.make_params(**{**nw, **band, "domain": "astar", "heuristic": "gap", "block_width": 1,
                                             "incremental_doubling": False}),
        # tests.rs:68-79 `band_doubling`: SH(k=5, exact, no pruning); block_width 4 here (1 in the reference: see the
        # known-quirk tests) and once as a 256-wide band with k=12
        "band_doubling_sh": o.make_params(**{**nw, **band, "domain": "astar", "heuristic": "sh", "k": 5, "block_width": 4,
                                             "incremental_doubling": False}),
        "sh_k12_w256": o.make_params(domain="astar", heuristic="sh", k=12, doubling="band", start="h0", block_width=256,
                                     sparse=True, incremental_doubling=True, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True),
        # tests.rs:81-105 `nw_prune` / `dt_trace`: GCSH(exact k, Pruning::start) with block_width 256 (k=15 there; the
        # grid is short, so also run small k to get real chains), and the `full` preset (params.rs:98-128)
        "gcsh_k15_prune": o.make_params(**{**nw, **band, "domain": "astar", "heuristic": "gcsh", "k": 15, "block_width": 256}),
        "gcsh_k6_dt": o.make_params(**{**nw, **band, "domain": "astar", "heuristic": "gcsh", "k": 6, "block_width": 64,
                                       "dt_trace": True}),
        "gcsh_k5_p3_noprune": o.make_params(**{**nw, **band, "domain": "astar", "heuristic": "gcsh", "k": 5, "p": 3,
                                               "block_width": 32, "prune": False}),
        "preset_full": o.params_full(),
        "incremental_doubling": o.make_params(**{**nw, **band, "domain": "astar", "heuristic": "gap", "block_width": 64,
                                                 "dt_trace": True, "incremental_doubling": True}),
        "gap_start": o.make_params(**{**nw, **band, "domain": "gap_start", "block_width": 64}),
        "linear_search": o.make_params(**{**nw, "doubling": "linear", "start": "zero", "delta": 3.0, "domain": "astar",
                                          "heuristic": "gap", "block_width": 32}),
        "preset_nw": o.params_nw(),
        "preset_simple": o.params_simple(),
        "simple_scalar_noilp": o.make_params(domain="astar", heuristic="gap", doubling="band", start="h0", sparse=True,
                                             incremental_doubling=True, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True,
                                             simd=True, no_ilp=True),
    }


CONFIG_NAMES = ["full", "band_doubling_gapgap", "dt_trace_gapgap", "band_doubling_dijkstra", "band_doubling_edlib",
                "band_doubling_w1", "incremental_doubling", "gap_start", "linear_search", "preset_nw", "preset_simple",
                "simple_scalar_noilp", "band_doubling_sh", "sh_k12_w256", "gcsh_k15_prune", "gcsh_k6_dt", "gcsh_k5_p3_noprune", "preset_full"]


def check(o, a, b, params, label, self_check=False):
    want = o.levenshtein(a, b)
    cost, cigar, stats = o.cpu_align(a, b, params, trace=True, self_check=self_check)
    assert cost == want, label
    assert cigar is not None
    assert o.cigar_verify(cigar, a, b) == want, (label, cigar)
    cost2, _, _ = o.cpu_align(a, b, params, trace=False)
    assert cost2 == want, label
    return stats


@pytest.mark.parametrize("name", CONFIG_NAMES)
def test_hardcoded_pairs(oracle, name):
    p = configs(oracle)[name]
    for a, b in PA_TEST_PAIRS:
        check(oracle, a, b, p, name, self_check=True)


@pytest.mark.parametrize("name", CONFIG_NAMES)
def test_random_grid(oracle, name):
    """The full pa-test grid with fixed seeds (the reference samples a random quarter of it per run)."""
    p = configs(oracle)[name]
    ns = PA_TEST_NS if name not in ("full", "band_doubling_w1") else [n for n in PA_TEST_NS if n <= 300]
    for n in ns:
        for e in PA_TEST_ES:
            a, b = gen_pair(n, e, seed=31415 + n * 7 + int(e * 1000))
            check(oracle, a, b, p, f"{name} n={n} e={e}", self_check=(n % 50 == 0))


def test_error_models_beyond_uniform(oracle):
    """pa-test uses 4 error models (Uniform, NoisyInsert, NoisyDelete, SymmetricRepeat); emulate the
    structural ones: a long insertion, a long deletion, and a repeated block."""
    p_all = configs(oracle)
    for seed in range(6):
        base = rand_seq(700, seed=seed)
        ins = base[:300] + rand_seq(150, seed=100 + seed) + base[300:]
        dele = base[:200] + base[420:]
        rep = base[:350] + base[250:350] * 2 + base[350:]
        for x, y in ((base, mutate(ins, 0.03, seed)), (base, mutate(dele, 0.03, seed)), (base, mutate(rep, 0.05, seed)),
                     (ins, base), (rep, dele)):
            for name in ("preset_simple", "incremental_doubling", "dt_trace_gapgap", "preset_nw"):
                check(oracle, x, y, p_all[name], name, self_check=True)


def test_c_example_pair(oracle):
    # astarpa-c/example.c:8-29: cost 2 through astarpa2_simple
    cost, cigar, _ = oracle.cpu_align(b"ACTCGCT", b"AACTCGTT", oracle.params_simple())
    assert cost == 2
    assert oracle.cigar_verify(cigar, b"ACTCGCT", b"AACTCGTT") == 2


def test_bigger_pairs_simple(oracle):
    for n, e, seed in [(5000, 0.05, 1), (10000, 0.1, 2), (20000, 0.02, 3), (3000, 0.3, 4)]:
        a, b = gen_pair(n, e, seed)
        want = oracle.nw_cost(a, b, True)
        for params in (oracle.params_simple(), configs(oracle)["incremental_doubling"]):
            cost, cigar, stats = oracle.cpu_align(a, b, params)
            assert cost == want
            assert oracle.cigar_verify(cigar, a, b) == want
        assert stats["f_max_tries"] >= 1


def test_empty_inputs(oracle):
    for name in ("preset_nw", "preset_simple", "incremental_doubling", "band_doubling_gapgap"):
        p = configs(oracle)[name]
        for a, b in ((b"", b""), (b"ACGT", b""), (b"", b"ACG"), (b"A", b"A"), (b"A", b"C")):
            check(oracle, a, b, p, name)


def test_known_quirk_reused_block_keeps_old_original_j_range(oracle):
    """band.rs:123-126 would abort here in the reference: a reused block keeps its old original_j_range
    (blocks.rs:190-197), fixed_j_range is clipped by it (domain.rs:298), the band at f_max == d misses the
    path, and a larger f_max then finds cost <= last_s.  We count it and still return the exact answer."""
    a, b = gen_pair(110, 0.02, seed=31415 + 110 * 7 + 20)
    p = oracle.make_params(doubling="band", start="gap", domain="astar", heuristic="gap", block_width=1, sparse_h=True,
                           incremental_doubling=False)
    cost, cigar, stats = oracle.cpu_align(a, b, p)
    assert cost == oracle.levenshtein(a, b) == 2
    assert oracle.cigar_verify(cigar, a, b) == 2
    assert stats["sanity_violations"] >= 1


def test_known_quirk_incremental_block_width_1(oracle):
    """block_width 1 + incremental doubling: the rows preserved by init_v_with_overlap_preserve_fixed
    (blocks.rs:803-808 'still guaranteed to be correct') are not always exact, so the traceback can fail with the
    reference's own 'PARENT NOT FOUND' panic.  Unverifiable without a Rust toolchain; the 64/256-wide configurations
    the presets use pass the incremental == from-scratch self check everywhere.  Either outcome is accepted here, a
    wrong answer is not."""
    a, b = gen_pair(260, 1.0, seed=31415 + 260 * 7 + 1000)
    p = oracle.make_params(doubling="band", start="gap", domain="astar", heuristic="gap", block_width=1, sparse_h=True)
    try:
        cost, cigar, _ = oracle.cpu_align(a, b, p)
    except oracle.EnginePanic:
        return
    assert cost == oracle.levenshtein(a, b)
    assert oracle.cigar_verify(cigar, a, b) == cost


def test_sh_heuristic_matches_definition(oracle):
    """sh.rs:88-92 / sh_contours.rs:63-75: h(i) = #seeds starting at >= i minus #those with an exact match in b,
    seeds = disjoint k-mers of a at 0,k,2k,.. (qgrams.rs:99-109)."""
    for n, e, k, seed in [(100, 0.1, 5, 1), (257, 0.3, 4, 2), (1000, 0.05, 8, 3), (64, 1.0, 3, 4), (7, 0.0, 8, 5), (50, 0.0, 5, 6)]:
        a, b = gen_pair(n, e, seed)
        kmers_b = {b[j:j + k] for j in range(len(b) - k + 1)}
        want = [sum(1 for s in range(0, len(a) - k + 1, k) if s >= i and a[s:s + k] not in kmers_b) for i in range(len(a) + 1)]
        assert oracle.sh_h(a, b, k) == want
    # identical sequences: every seed matches => h == 0 everywhere; empty b: nothing matches
    a = rand_seq(200, seed=9)
    assert set(oracle.sh_h(a, a, 10)) == {0}
    assert oracle.sh_h(a, b"", 10)[0] == 20


def _gcsh_reference_h(a, b, k, queries):
    """Definition of GCSH (no local pruning): chain score over exact matches in the gap-transformed partial order."""
    n, m = len(a), len(b)
    seeds = list(range(0, n - k + 1, k))
    pot = [sum(1 for s in seeds if s >= i) for i in range(n + 1)]
    T = lambda i, j: (i - j - pot[i], j - i - pot[i])
    le = lambda p, q: p[0] <= q[0] and p[1] <= q[1]
    tt = T(n, m)
    kmers = {}
    for s in seeds:
        kmers.setdefault(a[s:s + k], []).append(s)
    matches = sorted({(i, j) for j in range(m - k + 1) for i in kmers.get(b[j:j + k], []) if le(T(i, j), tt)})
    layer = {}
    points = []  # (T(start), layer)

    def score(q):
        return max([l for p, l in points if le(q, p)], default=0)

    for (i, j) in reversed(matches):
        e = T(i + k, j + k)
        if not le(e, tt):
            continue
        points.append((T(i, j), score(e) + 1))
    out = []
    for (i, j) in queries:
        v = score(T(i, j))
        out.append(max(abs((n - i) - (m - j)), pot[i]) if v == 0 else pot[i] - v)
    return out, matches


def test_gcsh_matches_definition(oracle):
    import random

    rnd = random.Random(5)
    for n, e, k, seed in [(120, 0.05, 4, 1), (300, 0.1, 5, 2), (400, 0.02, 6, 3), (200, 0.3, 3, 4), (64, 0.0, 8, 5)]:
        a, b = gen_pair(n, e, seed)
        queries = [(rnd.randrange(len(a) + 1), rnd.randrange(len(b) + 1)) for _ in range(300)] + [(0, 0), (len(a), len(b))]
        want_h, want_matches = _gcsh_reference_h(a, b, k, queries)
        got_h, got_matches = oracle.gcsh_probe(a, b, k, 0, queries)
        assert got_matches == want_matches
        assert got_h == want_h
        # admissible: h(0,0) never exceeds the true distance
        assert got_h[-2] <= oracle.levenshtein(a, b)


def test_gcsh_local_pruning_only_removes_matches(oracle):
    a, b = gen_pair(3000, 0.08, 7)
    _, all_m = oracle.gcsh_probe(a, b, 6, 0, [(0, 0)])
    h_p, kept = oracle.gcsh_probe(a, b, 6, 5, [(0, 0)])
    assert set(kept) <= set(all_m) and len(kept) < len(all_m)
    assert h_p[0] <= oracle.levenshtein(a, b)


def test_full_preset_medium_pairs(oracle):
    """A*PA2-full (GCSH k=12 p=14, pruning, incremental doubling) where seeds really chain."""
    for n, e, seed in [(5000, 0.02, 1), (20000, 0.05, 2), (50000, 0.01, 3), (8000, 0.12, 4), (30000, 0.0, 5)]:
        a, b = gen_pair(n, e, seed)
        want = oracle.nw_cost(a, b, True)
        cost, cigar, stats = oracle.cpu_align(a, b, oracle.params_full(), self_check=(n <= 20000))
        assert cost == want
        assert oracle.cigar_verify(cigar, a, b) == want
        simple = oracle.cpu_align(a, b, oracle.params_simple(), trace=False)[2]
        # the seed heuristic must not need more band than the gap heuristic
        assert stats["computed_lanes"] <= simple["computed_lanes"] * 1.05 + 64
