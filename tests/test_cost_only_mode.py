"""The reference's cost-only mode for the `simple` family (make_aligner(false) without incremental doubling, blocks.rs:252-277: ONE block
updated in place, its fixed_j_range the union over all columns) restated twice, independently: engine.hpp (C++, over the oracle kernels)
and tests/tools/cost_only_restatement.py (pure Python from the Rust text, Myers on big integers).  On the pair the GPU soak of round 2
found, both end on the SAME upper bound with the SAME number of passes and computed lanes -- so the mode itself returns upper bounds
(words of `v` below a block's range are not reset to +1 there, unlike init_v_with_overlap in the traced mode), it is not a slip of
engine.hpp.  pa_align(trace = 0) therefore does not reproduce it (include/pa_astarpa2.h): it returns the distance."""
import json
from pathlib import Path


def test_reference_cost_only_mode_returns_an_upper_bound_on_this_pair(oracle):
    from tests.tools.cost_only_restatement import CostOnly

    j = json.loads((Path(__file__).resolve().parent / "golden" / "cost_only_pair.json").read_text())
    a, b = j["a"].encode(), j["b"].encode()
    prm = oracle.make_params(domain="astar", heuristic="sh", k=12, doubling="band", start="h0", factor=2.0, block_width=256, sparse=True,
                             incremental_doubling=False, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True)
    dist = oracle.nw_cost(a, b, True)
    traced = oracle.cpu_align(a, b, prm, trace=True)
    cost_only = oracle.cpu_align(a, b, prm, trace=False)
    st = CostOnly(a, b, "sh", 12, True)
    got = st.cost()
    assert dist == traced[0] == 11325
    assert got == cost_only[0] == 11353
    assert (st.f_max_tries, st.computed_lanes) == (cost_only[2]["f_max_tries"], cost_only[2]["computed_lanes"])


def test_cost_only_restatement_agrees_with_the_engine_elsewhere(oracle):
    """... and on ordinary pairs the two restatements of that mode agree with each other and with the distance."""
    from tests.tools.cost_only_restatement import CostOnly
    from tests.util_seq import gen_pair

    for n, e, s, heur in [(300, 0.1, 1, "gap"), (3000, 0.05, 2, "gap"), (5000, 0.2, 3, "sh"), (2000, 0.3, 4, "none")]:
        a, b = gen_pair(n, e, s)
        prm = oracle.make_params(domain="astar", heuristic=heur, k=12, doubling="band", start="h0", factor=2.0, block_width=256, sparse=True,
                                 incremental_doubling=False, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True)
        want = oracle.cpu_align(a, b, prm, trace=False)
        st = CostOnly(a, b, heur, 12, True)
        assert st.cost() == want[0] == oracle.levenshtein(a, b)
        assert (st.f_max_tries, st.computed_lanes) == (want[2]["f_max_tries"], want[2]["computed_lanes"])


def test_cost_only_arms_of_the_second_restatement(oracle):
    """oracle/astarpa2_restated.py with trace = False (the cost-only arms of Blocks: one block updated in place without incremental
    doubling, the sparse blocks with it) against the engine's cost-only mode: the upper bound of the soak's pair, and cost, passes and
    block counters of random pairs over nine parameter sets (GCSH with pruning and incremental doubling among them)."""
    import random

    from oracle import astarpa2_restated as restated
    from tests.test_restated_engine import variants
    from tests.util_seq import gen_pair, rand_seq

    j = json.loads((Path(__file__).resolve().parent / "golden" / "cost_only_pair.json").read_text())
    a, b = j["a"].encode(), j["b"].encode()
    got = restated.align(a, b, heuristic="sh", k=12, trace=False)
    assert got[0] == 11353 and got[1] is None  # (the distance is 11325)
    keys = ["num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "f_max_tries"]
    vs = variants(oracle)
    rng = random.Random(3)
    for it in range(250):
        name = rng.choice(["simple", "sh12", "dijkstra", "gap_incr", "full", "gcsh_k8_p0_prune", "sh12_incr", "linear300", "gcsh_noprune"])
        n = rng.choice([rng.randint(1, 300), rng.randint(300, 2500), rng.randint(2500, 8000)])
        a, b = gen_pair(n, rng.choice([0.0, 0.02, 0.1, 0.2, 0.4]), rng.randint(1, 10**9))
        if rng.random() < 0.3 and n > 50:
            cut = rng.randint(0, len(b) - 1)
            ln = rng.randint(1, max(1, min(2000, len(b) // 2)))
            b = (b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, it) + b[cut:]) or b"A"
        prm, kw = vs[name]
        want = oracle.cpu_align(a, b, prm, trace=False)
        got = restated.align(a, b, trace=False, **kw)
        assert got[0] == want[0] and got[1] is None, (name, len(a), len(b))
        assert {k: got[2][k] for k in keys} == {k: want[2][k] for k in keys}, (name, len(a), len(b))
