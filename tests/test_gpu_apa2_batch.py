"""Batched A*PA2 on the GPU (pa_batch_create_params / pa_batch_align / pa_batch_pair_stats): ONE wavefront runs a pair's whole band
search (csrc/apa2_logic.hpp over the gfx950 backend of csrc/apa2_kernel.hpp), the traceback kernel walks the banded blocks of
the successful pass.  For every pair the cost, the CIGAR string and all twelve statistics must equal what the host engine over
the CPU oracle kernels returns for the same parameters (`oracle.cpu_align`) -- i.e. what a loop over pa_align returns."""
import random

import numpy as np
import pytest

from tests.test_sweep_emu import KEYS, variants
from tests.util_seq import PA_TEST_PAIRS, gen_pair, rand_seq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import astar_pairwise_aligner_amd as pa

    pa.require_gpu()
    return pa


@pytest.fixture(autouse=True)
def _two_pairs_per_strip_whatever_the_batch_size(monkeypatch):
    """The rendezvous of half-wave blocks (csrc/rdv_logic.hpp, strip2_kernel.hpp) is on by itself only for batches that fill the chip; these
    tests use small ones, so it is forced on: every comparison below also covers the fused strips."""
    monkeypatch.setenv("PA_APA2_RDV", "2")


def check(pa, oracle, pairs, oc, fallbacks=None, verify_only_sample=None):
    from tests.test_gpu_engine import gpu_params

    batch = pa.Batch(pairs, params=gpu_params(pa, oc))
    costs, cigars, fwd_ms, trace_ms = batch.align()
    stats = batch.pair_stats()
    idx = range(len(pairs)) if verify_only_sample is None else verify_only_sample
    for i in idx:
        a, b = pairs[i]
        want_cost, want_cigar, want_stats = oracle.cpu_align(a, b, oc)
        assert costs[i] == want_cost, (i, len(a), len(b))
        assert cigars[i] == want_cigar, (i, len(a), len(b), cigars[i][:80], want_cigar[:80])
        assert {k: stats[i][k] for k in KEYS} == {k: want_stats[k] for k in KEYS}, (i, len(a), len(b))
    if fallbacks is not None:
        assert batch.trace_fallbacks() == fallbacks
    costs2, cigars2, _, _ = batch.align()  # idempotent on resident inputs
    assert np.array_equal(costs, costs2) and cigars == cigars2
    batch.close()
    return costs, cigars, fwd_ms, trace_ms


def test_block_boundary_sizes(pa, oracle):
    pairs = []
    for n in (1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 769, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191):
        for e in (0.0, 0.05, 0.4, 1.0):
            pairs.append(gen_pair(n, e, seed=n * 7 + int(e * 100)))
    costs, _, _, _ = check(pa, oracle, pairs, oracle.params_simple(), fallbacks=0)
    for (a, b), c in zip(pairs, costs):
        assert c == oracle.levenshtein(a, b)


def test_pa_test_pairs_and_degenerate_inputs(pa, oracle):
    """Empty sequences are handed to the host engine (counted as fallbacks); the reference's literal pairs run on the device."""
    pairs = [p for p in PA_TEST_PAIRS] + [(b"", b""), (b"ACGT", b""), (b"", b"ACGTA"), (b"A", b"A"), (b"A", b"C")]
    check(pa, oracle, pairs, oracle.params_simple())


@pytest.mark.parametrize("name", ["simple", "dijkstra", "sh12", "sh5", "gap_nosparseh", "gap_nodt", "gap_startgap", "gap_startzero_f15", "linear"])
def test_variants_multi_strip(pa, oracle, name):
    """Several passes, bands of several 2048-row strips."""
    oc = variants(oracle)[name]
    pairs = [gen_pair(n, e, seed) for n, e, seed in [(300, 0.05, 1), (3000, 0.1, 3), (10000, 0.15, 4), (30000, 0.2, 6), (20000, 0.3, 5)]]
    costs, cigars, _, _ = check(pa, oracle, pairs, oc)
    for (a, b), c, cg in zip(pairs, costs, cigars):
        assert oracle.cigar_verify(cg, a, b) == c


def test_indels_and_unequal_lengths(pa, oracle):
    a = rand_seq(3000, seed=5)
    pairs = [
        (a, a[:1000] + a[1400:]),                       # 400-column deletion
        (a[:1000] + a[1400:], a),                       # 400-row insertion
        (a, a[:700] + rand_seq(300, seed=6) + a[700:]),  # foreign insert
        (rand_seq(700, seed=1), rand_seq(2300, seed=2)),  # unrelated, tall
        (rand_seq(2300, seed=3), rand_seq(70, seed=4)),   # unrelated, wide
        (a, a),
        (a[500:], a),                                   # the alignment starts with 500 insertions
        (a, a[500:]),                                   # ... with 500 deletions (walks the top row)
    ]
    check(pa, oracle, pairs, oracle.params_simple())


@pytest.mark.parametrize("seed", [1, 2])
def test_random_pairs_all_variants(pa, oracle, seed):
    rng = random.Random(seed)
    vs = variants(oracle)
    for name in vs:
        pairs = []
        for _ in range(24):
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
            pairs.append((a, b))
        check(pa, oracle, pairs, vs[name])


def test_c4_like_mixed_divergence(pa, oracle):
    """BASELINE C4's shape at a size the CPU engine checks in seconds: 10 kbp pairs, 1-15 % mixed divergence."""
    pairs = [gen_pair(10000, e, seed=100 + k) for k, e in enumerate([0.01, 0.05, 0.10, 0.15] * 16)]
    check(pa, oracle, pairs, oracle.params_simple(), fallbacks=0)


def test_c3_like_100kbp(pa, oracle):
    """BASELINE C3's pair (100 kbp, 5 %) and three neighbours through the batched path."""
    pairs = [gen_pair(100000, e, seed=7 + k) for k, e in enumerate([0.05, 0.01, 0.10, 0.05])]
    check(pa, oracle, pairs, oracle.params_simple(), fallbacks=0)


def test_one_1mbp_pair_next_to_short_ones(pa, oracle):
    """A 1 Mbp pair (3907 blocks, bands of hundreds of words: K = 4 strips chained through the granule rows, multi-chunk prefix sums and
    scans) in the same batch as short pairs; also very unequal lengths."""
    pairs = [gen_pair(1_000_000, 0.03, seed=21), gen_pair(2000, 0.1, seed=22), (rand_seq(300, seed=23), rand_seq(40_000, seed=24)),
             (rand_seq(40_000, seed=25), rand_seq(300, seed=26)), gen_pair(70_000, 0.25, seed=27)]
    costs, cigars, _, _ = check(pa, oracle, pairs, oracle.params_simple())
    for (a, b), c, cg in zip(pairs, costs, cigars):
        assert oracle.cigar_verify(cg, a, b) == c


def test_cost_only_run(pa, oracle):
    """pa_batch_run on an A*PA2 batch: the band search without the traceback kernels; costs = distances, block statistics those of
    the traced band, trace statistics zero; empty sequences through the host engine."""
    pairs = [gen_pair(n, e, seed=n) for n, e in ((300, 0.05), (5000, 0.1), (20000, 0.2), (777, 0.0))] + [(b"", b"ACG"), (b"ACGT", b"")]
    batch = pa.Batch(pairs, params=pa.AstarPa2Params.simple())
    costs, ms = batch.run()
    stats = batch.pair_stats()
    for (a, b), c, st in zip(pairs, costs, stats):
        assert c == oracle.levenshtein(a, b)
        if a and b:
            want = oracle.cpu_align(a, b, oracle.params_simple())[2]
            assert all(st[k] == want[k] for k in ("num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "f_max_tries", "sanity_violations"))
            assert all(st[k] == 0 for k in ("dt_trace_tries", "dt_trace_success", "dt_trace_fallback", "fill_tries", "fill_success", "fill_fallback"))
    costs2, cigars, _, _ = batch.align()  # and the traced call on the same batch afterwards
    assert costs2.tolist() == costs.tolist() and all(oracle.cigar_verify(g, a, b) == c for (a, b), c, g in zip(pairs, costs2, cigars))
    batch.close()


def test_unsupported_parameters_are_refused(pa, oracle):
    from tests.test_gpu_engine import gpu_params

    for oc in (oracle.params_nw(), oracle.make_params(domain="gap_gap", heuristic="none", start="gap")):
        with pytest.raises(pa.PaError):
            pa.Batch([gen_pair(500, 0.1, 1)], params=gpu_params(pa, oc))


def test_few_long_pairs_take_the_single_pair_engine_and_agree_with_the_kernel_route(pa, oracle):
    """One or two pairs of >= 32 768 bases go through the single-pair engine (many wavefronts per pass) instead of one lone
    wavefront per pair; a third pair sends the same sequences through the batch kernels.  Both must give the CPU-kernel engine's
    cost, CIGAR string and statistics."""
    oc = oracle.params_simple()
    long_pairs = [gen_pair(40_000, 0.04, seed=91), gen_pair(33_000, 0.11, seed=92), (rand_seq(36_000, 5), rand_seq(34_000, 6))]
    c1, g1, ms1, _ = check(pa, oracle, long_pairs[:1], oc, fallbacks=0)
    c2, g2, _, _ = check(pa, oracle, long_pairs[1:], oc, fallbacks=0)
    c4, g4, _, _ = check(pa, oracle, long_pairs + [gen_pair(3000, 0.1, seed=93)], oc, fallbacks=0)
    assert list(c2) == list(c4[1:3]) and g2 == g4[1:3] and c1[0] == c4[0] and g1[0] == g4[0]
    from tests.test_gpu_engine import gpu_params

    short = pa.Batch(long_pairs[:1] + [gen_pair(20_000, 0.05, seed=94)], params=gpu_params(pa, oc))  # one short pair: kernel route
    short.align()
    short.close()


def test_allocation_cache_reuses_large_buffers_and_leaks_nothing_into_results(pa, oracle):
    """Large device buffers of a destroyed batch are handed to the next one (pa_astarpa2.h): the second creation is served from the
    cache, results are those of the oracle whatever the recycled memory held before, pa_release_pools empties it."""
    from astar_pairwise_aligner_amd import capi
    from tests.test_gpu_engine import gpu_params

    oc = oracle.params_simple()
    capi.release_pools()
    assert capi.alloc_cache_stats()["cached_bytes"] == 0
    pairs1 = [gen_pair(20_000, 0.05, seed=200 + i) for i in range(48)]
    pairs2 = [gen_pair(20_000, 0.12, seed=300 + i) for i in range(48)]  # the same sizes, other contents, wider bands
    s0 = capi.alloc_cache_stats()
    check(pa, oracle, pairs1, oc, fallbacks=0, verify_only_sample=range(0, 48, 7))
    s1 = capi.alloc_cache_stats()
    assert s1["cached_bytes"] >= 16 << 20 and s1["misses"] > s0["misses"]
    check(pa, oracle, pairs2, oc, fallbacks=0, verify_only_sample=range(0, 48, 5))
    s2 = capi.alloc_cache_stats()
    assert s2["hits"] > s1["hits"]
    bt = pa.Batch(pairs2[:8] + pairs1[:8], trace=True)  # a full-DP traced batch out of the same recycled memory
    costs, cigars, _, _ = bt.align()
    bt.close()
    prm = oracle.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=True, incremental_doubling=False, dt_trace=False)
    for (a, b), c, g in zip(pairs2[:8] + pairs1[:8], costs, cigars):
        assert (int(c), g) == oracle.cpu_align(a, b, prm)[:2]
    capi.release_pools()
    assert capi.alloc_cache_stats()["cached_bytes"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"PA_ALIGN_CHUNKS": "3"}, {"PA_TRACE_ORDER": "0"}, {"PA_ALIGN_CHUNKS": "4", "PA_TRACE_ORDER": "0"}])
def test_chunks_and_trace_order_do_not_change_results(env):
    """pa_batch_align runs ONE chunk by default and starts the traceback's most expensive pairs first (trace_order_kernel).  The
    other shapes -- several chunks on streams of their own (PA_ALIGN_CHUNKS), the traceback in index order (PA_TRACE_ORDER=0) -- must give
    the same cost, CIGAR string and statistics: both families, 300 pairs of mixed length and divergence, against the CPU-kernel engine."""
    import os
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import oracle
        import astar_pairwise_aligner_amd as pa
        from tests.test_gpu_engine import gpu_params
        from tests.test_sweep_emu import KEYS
        from tests.util_seq import gen_pair
        pairs = [gen_pair(400 + 37 * (i %% 50), (0.01, 0.05, 0.10, 0.15, 0.25)[i %% 5], 9000 + i) for i in range(300)]
        for oc in (oracle.params_simple(), oracle.params_full()):
            b = pa.Batch(pairs, params=gpu_params(pa, oc))
            for rep in range(2):
                costs, cigars, _, _ = b.align()
                st = b.pair_stats()
                for i, (x, y) in enumerate(pairs):
                    w = oracle.cpu_align(x, y, oc)
                    assert (int(costs[i]), cigars[i]) == w[:2] and all(st[i][k] == w[2][k] for k in KEYS), (i, rep)
            assert b.trace_fallbacks() == 0
            b.close()
        full = pa.Batch(pairs, trace=True)  # the full-DP traced batch shares the traceback kernel and the chunks
        costs, cigars, _, _ = full.align()
        nw = oracle.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=True, incremental_doubling=False, dt_trace=False)
        for i, (x, y) in enumerate(pairs[:60]):
            w = oracle.cpu_align(x, y, nw)
            assert (int(costs[i]), cigars[i]) == w[:2], i
        print("ok")
    """) % str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("preset", ["simple", "full"])
def test_two_pairs_per_strip_changes_nothing(pa, oracle, monkeypatch, preset):
    """Round 5: two blocks of at most 16 words from two pairs of one workgroup run as ONE strip (pair A in lanes 0..31, pair B in lanes
    32..63).  With the rendezvous off, on, impatient and very patient: cost, CIGAR and every statistic of every pair are the same, and the
    counters say the fused strips were really taken."""
    prm = pa.AstarPa2Params.full() if preset == "full" else pa.AstarPa2Params.simple()
    rng = random.Random(11)
    pairs = [gen_pair(rng.choice([700, 2500, 3000, 6000]), rng.choice([0.01, 0.05, 0.1, 0.15, 0.3]), seed=1000 + i) for i in range(600)]
    pairs += [gen_pair(200, 0.1, seed=5), (b"ACGT" * 300, b"ACGT" * 290), gen_pair(20_000, 0.2, seed=6)]  # (the last: bands taller than half a wave)
    got = {}
    for mode, patience in (("0", None), ("2", None), ("2", "0"), ("2", "400")):
        monkeypatch.setenv("PA_APA2_RDV", mode)
        if patience is None:
            monkeypatch.delenv("PA_APA2_RDV_PATIENCE_US", raising=False)
        else:
            monkeypatch.setenv("PA_APA2_RDV_PATIENCE_US", patience)
        bt = pa.Batch(pairs, params=prm)
        costs, cigars, _, _ = bt.align()
        stats = [{k: s[k] for k in KEYS} for s in bt.pair_stats()]
        rd = bt.rdv_stats()
        bt.close()
        got[(mode, patience)] = (costs.tolist(), cigars, stats)
        if mode == "0":
            assert rd["fused"] == rd["served"] == 0
        else:
            assert rd["fused"] == rd["served"] and (rd["fused"] > 100 or patience == "0"), rd
            if patience == "400":
                assert rd["fused"] > rd["alone"], rd  # (a patient block mostly finds its partner)
    base = got[("0", None)]
    assert all(v == base for v in got.values())
    for (a, b), c in list(zip(pairs, base[0]))[::40]:
        assert c == oracle.levenshtein(a, b)


def test_sh_with_local_pruning_batch_and_single(pa, oracle):
    """SH with HeuristicParams.p != 0 (local pruning applies to SH's matches too: pa-heuristic/src/cli.rs:168-180, sh.rs:48): the host-built
    table reaches the batch kernel and the sweep of pa_align; both equal the engine over the CPU kernels."""
    from tests.test_gpu_engine import gpu_params
    from tests.test_restated_engine import BASE

    pairs = [gen_pair(n, e, seed) for n, e, seed in [(300, 0.05, 1), (3000, 0.1, 3), (9000, 0.2, 4), (6000, 0.25, 6), (12000, 0.15, 5)]]
    for k, p in ((8, 3), (12, 14)):
        oc = oracle.make_params(**{**BASE, "heuristic": "sh", "k": k, "p": p})
        check(pa, oracle, pairs, oc)
        al = gpu_params(pa, oc).make_aligner(True)
        for a, b in pairs[:3]:
            want = oracle.cpu_align(a, b, oc)
            got = al.align_with_stats(a, b)
            assert (got[0], got[1]) == want[:2] and {k_: got[2][k_] for k_ in KEYS} == {k_: want[2][k_] for k_ in KEYS}
