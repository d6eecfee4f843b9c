"""tests/golden/restated_<variant>.json on the CPU: the files are what the committed generator writes (a sample regenerated from
oracle/astarpa2_restated.py), and the product's host engine over the CPU oracle kernels (oracle.cpu_align: csrc/engine.hpp) returns the
same cost, CIGAR and statistics on a sample of every parameter set.  The GPU side of the same fixtures: tests/test_gpu_restated_fixtures.py."""
import pytest

from oracle import astarpa2_restated as restated
from tests import restated_fixture as rf
from tests.test_restated_engine import variants


def test_every_parameter_set_has_its_fixture(oracle):
    names = rf.variant_names()
    assert sorted(names) == sorted(variants(oracle)) and len(names) == 26
    total = 0
    for name in names:
        doc = rf.load(name)
        assert doc["n_pairs"] == rf.N_PAIRS >= 2000 and len(doc["rows"]) == rf.N_PAIRS and doc["row"][2:] == rf.KEYS
        assert doc["restated_kwargs"] == variants(oracle)[name][1]
        total += sum(r is not None for r in doc["rows"])
    assert total > 25 * rf.N_PAIRS


def test_pairs_cover_what_they_claim():
    ps = [rf.pair_for(i) for i in range(0, rf.N_PAIRS, 8)]
    ns = [len(a) for a, _ in ps]
    assert min(ns) <= 20 and max(ns) > 8000 and sum(n < 300 for n in ns) > 50 and sum(n > 2500 for n in ns) > 50
    assert sum(abs(len(a) - len(b)) > 100 for a, b in ps) > 20  # long indels


@pytest.mark.parametrize("part", range(4))
def test_sample_regenerates_and_equals_the_cpu_kernel_engine(oracle, part):
    vs = variants(oracle)
    for vi, name in enumerate(sorted(vs)):
        if vi % 4 != part:
            continue
        prm, kw = vs[name]
        rows = rf.load(name)["rows"]
        for i in range(vi % 32, rf.N_PAIRS, 32):
            if rows[i] is None:
                continue
            a, b = rf.pair_for(i)
            if i % 128 == vi % 32:  # (the pure-Python restatement is the slow side: a quarter of the sample)
                assert rf.row_of(*restated.align(a, b, **kw)) == rows[i], (name, i)  # the file is what the generator writes
            assert rf.row_of(*oracle.cpu_align(a, b, prm)) == rows[i], (name, i)   # ... and what csrc/engine.hpp over the CPU kernels returns


def test_long_pairs_fixture_on_the_cpu(oracle):
    """tests/golden/restated_long_<set>.json: a sample against the CPU-kernel engine (and, for two pairs per set, the generator itself)."""
    vs = variants(oracle)
    for name in rf.LONG_VARIANTS:
        prm, kw = vs[name]
        doc = rf.load_long(name)
        assert doc["n_pairs"] == rf.N_LONG and doc["restated_kwargs"] == kw
        for i in range(rf.LONG_VARIANTS.index(name), rf.N_LONG, 12):
            a, b = rf.long_pair_for(i)
            assert rf.row_of(*oracle.cpu_align(a, b, prm)) == doc["rows"][i], (name, i)
            if i < 24:
                assert rf.row_of(*restated.align(a, b, **kw)) == doc["rows"][i], (name, i)


@pytest.mark.parametrize("name", sorted(rf.COLL_VARIANTS))
def test_colliding_u32_keys_fixture_on_the_cpu(oracle, name):
    """tests/golden/restated_kcoll_<set>.json (k > 16, seeds that share their last 16 characters -- one key in the reference's u32-keyed
    match table, pa-heuristic/src/matches/exact.rs:47,53,56): every pair against the CPU-kernel engine, a sample against the generator."""
    kw = rf.COLL_VARIANTS[name]
    doc = rf.load_coll(name)
    assert doc["n_pairs"] == rf.N_COLL and doc["restated_kwargs"] == kw and len(doc["rows"]) == rf.N_COLL
    d = dict(heuristic="gap", k=12, sparse_h=True, block_width=256, dt_trace=True, max_g=40, fr_drop=10, domain="astar", sparse=True,
             doubling="band", start="h0", factor=2.0, delta=1.0, incremental_doubling=False, p=0, prune=False)
    d.update(kw)
    prm = oracle.make_params(**d)
    colliding = 0
    for i in range(rf.N_COLL):
        a, b = rf.collision_pair_for(i, kw["k"])
        assert rf.row_of(*oracle.cpu_align(a, b, prm)) == doc["rows"][i], (name, i)
        if i % 16 == 0:
            assert rf.row_of(*restated.align(a, b, **kw)) == doc["rows"][i], (name, i)
            colliding += rf.count_collision_matches(a, b, kw["k"]) > 0
    assert colliding >= rf.N_COLL // 16 * 3 // 4, colliding  # the pairs do what they are for: matches that are NOT equal k-mers
