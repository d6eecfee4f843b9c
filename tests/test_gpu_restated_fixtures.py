"""Independent engine-level parity on the GPU: the committed fixtures of the SECOND restatement (tests/golden/restated_<variant>.json, written by
tests/golden/make_restated.py from oracle/astarpa2_restated.py alone) against pa_align and the batch kernels -- cost, CIGAR (SHA-256 of the
string) and the eleven statistics of 2 048 seeded pairs under each of the 26 parameter sets.  Nothing of csrc/engine.hpp over the CPU
kernels stands between the expected values and the GPU's: the parameters come from the fixture file, the pairs from their index.

What goes where: every pair of a parameter set the batch kernels take goes through ONE batch (apa2_kernel / apa2_full_kernel + the
traceback kernel); a stride of the pairs of every parameter set goes through pa_align one call at a time (the sweep kernel, the host-driven
HIP engine); parameter sets the batch kernels do not take (dense or 64-wide blocks, the other domains) go through pa_align at a finer stride.
Reference rules restated: pa-test/src/lib.rs:65-99 (cost, valid CIGAR), astarpa2/src/blocks.rs:471-543 (the statistics)."""
import pytest

from tests import restated_fixture as rf

pytestmark = pytest.mark.gpu

NAMES = rf.variant_names()


@pytest.fixture(scope="module")
def pa():
    import astar_pairwise_aligner_amd as pa

    pa.require_gpu()
    return pa


@pytest.fixture(scope="module")
def pairs():
    return [rf.pair_for(i) for i in range(rf.N_PAIRS)]


@pytest.mark.parametrize("name", NAMES)
def test_fixture_through_the_batch_kernels_and_pa_align(pa, pairs, name):
    doc = rf.load(name)
    rows = doc["rows"]
    prm = rf.params_from_kwargs(pa, doc["restated_kwargs"])
    idx = [i for i in range(rf.N_PAIRS) if rows[i] is not None]
    batchable = bool(pa.capi.batch_params_supported(prm))
    if batchable:
        bt = pa.Batch([pairs[i] for i in idx], params=prm)
        costs, cigars, _, _ = bt.align()
        stats = bt.pair_stats()
        bt.close()
        bad = [i for k, i in enumerate(idx) if rf.row_of(costs[k], cigars[k], stats[k]) != rows[i]]
        assert not bad, (name, len(bad), bad[:5], [(rf.row_of(costs[idx.index(i)], cigars[idx.index(i)], stats[idx.index(i)]), rows[i]) for i in bad[:2]])
    al = prm.make_aligner(True)
    stride = 16 if batchable else 4
    off = NAMES.index(name) % stride
    for i in idx[off::stride]:
        a, b = pairs[i]
        got = rf.row_of(*al.align_with_stats(a, b))
        assert got == rows[i], (name, i, len(a), len(b), got, rows[i])


def test_the_batch_kernels_take_most_parameter_sets(pa):
    n = sum(bool(pa.capi.batch_params_supported(rf.params_from_kwargs(pa, rf.load(name)["restated_kwargs"]))) for name in NAMES)
    assert len(NAMES) == 26 and n >= 18


@pytest.mark.parametrize("name", rf.LONG_VARIANTS)
def test_long_pairs_fixture(pa, name, monkeypatch):
    """The LONG pairs of the second restatement (8 000 - 60 000 bases, up to 20 % divergence, long indels): bands of several strips -- the
    K = 2 / 3 / 4 strips, `eq` words in LDS --, pairs whose band leaves their window of the column store (aligned again with full
    columns), re-fills taller than a strip in the traceback.  One batch with two pairs per strip forced on, one with it off, and every
    eighth pair through pa_align."""
    doc = rf.load_long(name)
    rows = doc["rows"]
    prm = rf.params_from_kwargs(pa, doc["restated_kwargs"])
    pairs = [rf.long_pair_for(i) for i in range(rf.N_LONG)]
    assert pa.capi.batch_params_supported(prm)
    for mode in ("2", "0"):
        monkeypatch.setenv("PA_APA2_RDV", mode)
        bt = pa.Batch(pairs, params=prm)
        costs, cigars, _, _ = bt.align()
        stats = bt.pair_stats()
        bt.close()
        bad = [i for i in range(rf.N_LONG) if rf.row_of(costs[i], cigars[i], stats[i]) != rows[i]]
        assert not bad, (name, mode, bad[:5], [(rf.row_of(costs[i], cigars[i], stats[i]), rows[i]) for i in bad[:2]])
    al = prm.make_aligner(True)
    for i in range(rf.LONG_VARIANTS.index(name) % 8, rf.N_LONG, 8):
        assert rf.row_of(*al.align_with_stats(*pairs[i])) == rows[i], (name, i)


@pytest.mark.parametrize("name", sorted(rf.COLL_VARIANTS))
def test_colliding_u32_keys_fixture(pa, name):
    """k > 16 with seeds that share their last 16 characters (tests/golden/restated_kcoll_<set>.json): the reference's match table is keyed
    on `q as u32` (pa-heuristic/src/matches/exact.rs:47,53,56), so such seeds match each other's k-mers.  One batch -- matches built on the
    device by gcsh_build_kernel for the GCSH sets, or on the host when a pair outgrows the kernel's candidate buffers -- and every sixth pair
    through pa_align."""
    doc = rf.load_coll(name)
    rows, kw = doc["rows"], doc["restated_kwargs"]
    prm = rf.params_from_kwargs(pa, kw)
    pairs = [rf.collision_pair_for(i, kw["k"]) for i in range(rf.N_COLL)]
    assert pa.capi.batch_params_supported(prm)
    bt = pa.Batch(pairs, params=prm)
    costs, cigars, _, _ = bt.align()
    stats = bt.pair_stats()
    bt.close()
    bad = [i for i in range(rf.N_COLL) if rf.row_of(costs[i], cigars[i], stats[i]) != rows[i]]
    assert not bad, (name, len(bad), bad[:5], [(rf.row_of(costs[i], cigars[i], stats[i]), rows[i]) for i in bad[:2]])
    al = prm.make_aligner(True)
    for i in range(sorted(rf.COLL_VARIANTS).index(name), rf.N_COLL, 6):
        assert rf.row_of(*al.align_with_stats(*pairs[i])) == rows[i], (name, i)
