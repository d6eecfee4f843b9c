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
