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
                "simple_scalar_noilp", "band_doubling_sh", "sh_k12_w256"]


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
