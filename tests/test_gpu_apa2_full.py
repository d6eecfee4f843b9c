"""Batched A*PA2 for the WHOLE family on the GPU -- AstarPa2Params::full() (GCSH with local pruning, pruning of matches between blocks,
incremental doubling) and its relatives through pa_batch_create_params / pa_batch_align: one wavefront runs a pair's whole band search
(csrc/apa2_full_logic.hpp over the gfx950 backend of csrc/apa2_full_kernel.hpp, the heuristic's contours derived and probed on the
device), the traceback kernel walks the banded blocks of the successful pass.  For every pair the cost, the CIGAR string and all twelve
statistics must equal what the host engine over the CPU oracle kernels returns for the same parameters (`oracle.cpu_align`) -- i.e.
what a loop over pa_align / astarpa2_full returns -- and the second, independent restatement (oracle/astarpa2_restated.py)."""
import random

import numpy as np
import pytest

from tests.test_restated_engine import variants
from tests.test_sweep_emu import KEYS
from tests.util_seq import PA_TEST_PAIRS, gen_pair, rand_seq

pytestmark = pytest.mark.gpu

FULL_FAMILY = ["full", "gcsh_noprune", "gcsh_k8_p0_prune", "gcsh_k6_p3_prune_incr", "gcsh_k10_p5_nosparseh", "gap_incr", "sh12_incr", "dijkstra_incr_nodt",
               "gap_incr_f15"]


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


def check(pa, oracle, pairs, oc, verify_only_sample=None, max_fallbacks=None):
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
    if max_fallbacks is not None:
        assert batch.trace_fallbacks() <= max_fallbacks, batch.trace_fallbacks()
    costs2, cigars2, _, _ = batch.align()  # the pruning state starts from scratch: aligning again gives the same
    assert np.array_equal(costs, costs2) and cigars == cigars2
    batch.close()
    return costs, cigars, fwd_ms, trace_ms


def test_full_preset_is_taken(pa, oracle):
    from astar_pairwise_aligner_amd import capi
    from tests.test_gpu_engine import gpu_params

    for name in FULL_FAMILY + ["simple", "sh12"]:
        assert capi.batch_params_supported(gpu_params(pa, variants(oracle)[name][0])), name
    for name in ["nw", "gap_gap", "block64", "full_sparse"]:
        assert not capi.batch_params_supported(gpu_params(pa, variants(oracle)[name][0])), name


def test_device_heuristic_equals_host(pa, oracle):
    """h(i, j) of GCSH as one wavefront computes it (contours derived on the device, 64 layers per probe round) against csrc/gcsh.hpp on
    the host, at positions along the alignment, around it and all over the matrix."""
    from astar_pairwise_aligner_amd import capi

    rng = random.Random(11)
    for n, e, k, p, seed in [(3000, 0.05, 12, 14, 1), (20000, 0.08, 12, 14, 2), (100_000, 0.05, 12, 14, 3), (8000, 0.15, 6, 3, 4), (5000, 0.02, 8, 0, 5),
                             (2000, 0.6, 5, 2, 6), (40, 0.1, 12, 14, 7)]:
        a, b = gen_pair(n, e, seed)
        qs = [(0, 0), (len(a), len(b)), (0, len(b)), (len(a), 0)]
        for _ in range(3000):
            i = rng.randint(0, len(a))
            j = min(len(b), max(0, i + rng.randint(-40, 40))) if rng.random() < 0.7 else rng.randint(0, len(b))
            qs.append((i, j))
        want, _ = oracle.gcsh_probe(a, b, k, p, qs)
        got, layers = capi.gcsh_probe(a, b, k, p, qs)
        bad = [(q, w, g) for q, w, g in zip(qs, want, got) if w != g]
        assert not bad, (n, e, k, p, len(bad), bad[:5], layers)


def test_device_built_matches_equal_host(pa, oracle):
    """The matches GCSH keeps -- seeds, exact k-mer matches in the reference's push order, the transform filter, local pruning with its
    `next_match_per_diag` -- as csrc/gcsh_build_kernel.hpp finds them on the GPU against csrc/gcsh.hpp on the host, match for match:
    random pairs at every divergence, low-complexity and repetitive sequences (chains of seeds with the same k-mer, many candidates per
    row, matches that survive local pruning only through a match kept before them), k from 4 to 20, p from 0 to 14."""
    from astar_pairwise_aligner_amd import capi

    rng = random.Random(77)
    cases = []
    for n, e, k, p, seed in [(3000, 0.05, 12, 14, 1), (20000, 0.08, 12, 14, 2), (100_000, 0.05, 12, 14, 3), (8000, 0.15, 6, 3, 4), (5000, 0.02, 8, 0, 5),
                             (2000, 0.6, 5, 2, 6), (40, 0.1, 12, 14, 7), (10_000, 0.15, 12, 14, 8), (10_000, 0.25, 12, 14, 9), (30_000, 0.12, 10, 7, 10),
                             (6000, 0.1, 6, 14, 11), (9000, 0.1, 20, 5, 12), (12, 0.0, 12, 14, 13), (11, 0.0, 12, 14, 14), (25, 0.0, 12, 1, 15)]:
        cases.append((gen_pair(n, e, seed), k, p))
    from tests.test_restated_engine import long_kmer_collision_pair

    for k, p in ((20, 0), (24, 3), (31, 0), (17, 14)):  # seeds that share their last 16 characters match each other: the reference's u32 key (exact.rs:47-56)
        cases.append((long_kmer_collision_pair(k), k, p))
    must = len(cases)
    for it in range(60):  # repeats: tandem copies of a unit with a few edits -- chains of seeds with one k-mer, several candidates per row
        short = it < 30  # short enough for the kernel's candidate buffers (the others may be refused: such a pair goes to the host engine)
        unit = rand_seq(rng.randint(20, 120) if short else rng.randint(3, 40), it + 100)
        a = (unit * 400)[: rng.randint(100, 900) if short else rng.randint(1000, 4000)]
        b = bytearray(a)
        for _ in range(rng.randint(0, len(b) // 20)):
            q = rng.randrange(len(b))
            b[q] = rng.choice(b"ACGT")
        cut = rng.randint(0, len(b) // 2)
        cases.append(((a, bytes(b[:cut] + b[cut + rng.randint(0, 30):]) or b"A"), rng.choice([4, 5, 8, 12]), rng.choice([0, 1, 3, 14])))
        must += 1 if short else 0
    refused, multi, gentle_done, coll_done = 0, 0, 0, 0
    for t, ((a, b), k, p) in enumerate(cases):
        want = sorted(oracle.gcsh_probe(a, b, k, p, [(0, 0)])[1])
        try:
            got = capi.gcsh_matches(a, b, k, p)
        except capi.PaError as e:
            # only repeats may be refused: candidate buffers outgrown (rc -101) or more than 64 kept matches within reach of one search
            # (rc -102); such a pair goes to the host engine
            assert t >= 15 and ("rc=-101" in str(e) or "rc=-102" in str(e)), (t, len(a), len(b), k, p, str(e))
            refused += 1
            continue
        assert got == want, (len(a), len(b), k, p, len(got), len(want), [x for x in got if x not in set(want)][:5], [x for x in want if x not in set(got)][:5])
        multi += len(want) > len(a) // k  # more matches than seeds: rows with several candidates
        gentle_done += 19 <= t < must
        coll_done += 15 <= t < 19
    assert multi >= 3 and gentle_done >= 12 and refused <= 45 and coll_done >= 3, (multi, gentle_done, refused, coll_done)


def test_block_boundary_sizes_full(pa, oracle):
    pairs = []
    for n in (1, 2, 11, 12, 13, 31, 64, 65, 255, 256, 257, 511, 512, 513, 1025, 2047, 2049, 4097):
        for e in (0.0, 0.05, 0.4, 1.0):
            pairs.append(gen_pair(n, e, seed=n * 7 + int(e * 100)))
    costs, _, _, _ = check(pa, oracle, pairs, oracle.params_full(), max_fallbacks=0)
    for (a, b), c in zip(pairs, costs):
        assert c == oracle.levenshtein(a, b)


def test_pa_test_pairs_and_degenerate_inputs_full(pa, oracle):
    pairs = [p for p in PA_TEST_PAIRS] + [(b"", b""), (b"ACGT", b""), (b"", b"ACGTA"), (b"A", b"A"), (b"A", b"C")]
    check(pa, oracle, pairs, oracle.params_full(), max_fallbacks=3)  # (the three pairs with an empty sequence go to the host engine)


@pytest.mark.parametrize("name", FULL_FAMILY)
def test_family_several_passes(pa, oracle, name):
    """Several passes (three-range splits of incremental doubling, contours re-derived after pruning), bands of several strips."""
    oc = variants(oracle)[name][0]
    pairs = [gen_pair(n, e, seed) for n, e, seed in [(300, 0.05, 1), (3000, 0.1, 3), (10000, 0.15, 4), (30000, 0.2, 6), (20000, 0.3, 5)]]
    for n, e, seed in [(20_000, 0.15, 4), (12_000, 0.3, 6)]:  # long indels: more passes
        a, b = gen_pair(n, e, seed)
        cut = len(b) // 3
        pairs.append((a, b[:cut] + rand_seq(700, seed + 1) + b[cut:2 * cut] + b[2 * cut + 400:]))
    costs, cigars, _, _ = check(pa, oracle, pairs, oc, max_fallbacks=0)
    for (a, b), c, cg in zip(pairs, costs, cigars):
        assert oracle.cigar_verify(cg, a, b) == c


def test_random_pairs_random_family(pa, oracle):
    rng = random.Random(2024)
    vs = variants(oracle)
    for name in FULL_FAMILY:
        pairs = []
        for it in range(60):
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
            pairs.append((a, b))
        check(pa, oracle, pairs, vs[name][0], max_fallbacks=0)


def test_restatement_agrees_on_the_device_results(pa, oracle):
    """The second restatement (pure Python, shares nothing with engine.hpp or the kernels) on a sample of what the GPU returned."""
    from oracle import astarpa2_restated as restated
    from tests.test_gpu_engine import gpu_params

    oc, kw = variants(oracle)["full"]
    pairs = [gen_pair(n, e, seed) for n, e, seed in [(2000, 0.05, 21), (5000, 0.12, 22), (9000, 0.2, 23), (700, 0.3, 24)]]
    batch = pa.Batch(pairs, params=gpu_params(pa, oc))
    costs, cigars, _, _ = batch.align()
    stats = batch.pair_stats()
    for i, (a, b) in enumerate(pairs):
        got = restated.align(a, b, **kw)
        assert (int(costs[i]), cigars[i]) == got[:2]
        assert all(stats[i][k] == got[2][k] for k in KEYS if k != "sanity_violations")
    batch.close()


def test_c3_pair_and_small_batch_full(pa, oracle):
    """BASELINE C3 with the `full` parameter set: 100 kbp pairs at 5 % -- 16 of them in one batch (the batch kernels) and one alone."""
    pairs = [gen_pair(100_000, 0.05, seed=3_000_000 + s) for s in range(16)]
    check(pa, oracle, pairs, oracle.params_full(), verify_only_sample=[0, 5, 15], max_fallbacks=0)
    check(pa, oracle, pairs[:1], oracle.params_full())


@pytest.mark.parametrize("name", ["simple", "full", "sh12", "gap_incr"])
def test_band_proportional_columns_and_the_second_round(pa, oracle, name, monkeypatch):
    """The block-column store of a pair is a window around the main diagonal (pa_bitpacking_hip.h).  With a window far too small for the
    pairs (PA_APA2_WINDOW=4 words) most of them leave it and are aligned again with full columns: same cost, CIGAR string and statistics;
    with the default windows none of these pairs does, and a long indel (the band wanders off the diagonal) does."""
    from tests.test_gpu_engine import gpu_params

    oc = variants(oracle)[name][0]
    pairs = [gen_pair(n, e, seed) for n, e, seed in [(300, 0.05, 1), (3000, 0.1, 3), (10000, 0.15, 4), (30000, 0.08, 6), (20000, 0.03, 5), (9000, 0.3, 8)]]
    a = rand_seq(40_000, seed=17)
    indel = (a, a[:10_000] + a[16_000:])  # 6000 columns deleted: the alignment runs ~94 words off the diagonal of the rectangle
    want = [oracle.cpu_align(x, y, oc) for x, y in pairs + [indel]]

    def run(ps):
        batch = pa.Batch(ps, params=gpu_params(pa, oc))
        costs, cigars, _, _ = batch.align()
        stats = batch.pair_stats()
        out = (batch.window_retries(), batch.trace_fallbacks())
        batch.close()
        return costs, cigars, stats, out

    def same(costs, cigars, stats, idx):
        for t, i in enumerate(idx):
            assert (int(costs[t]), cigars[t]) == want[i][:2], (name, i)
            assert {k: stats[t][k] for k in KEYS} == {k: want[i][2][k] for k in KEYS}, (name, i)

    costs, cigars, stats, (retries, fallbacks) = run(pairs)
    same(costs, cigars, stats, range(len(pairs)))
    assert fallbacks == 0
    costs, cigars, stats, (retries_indel, fallbacks) = run(pairs + [indel])
    same(costs, cigars, stats, range(len(pairs) + 1))
    assert fallbacks == 0 and retries_indel >= retries
    monkeypatch.setenv("PA_APA2_WINDOW", "4")
    # (the library reads the variable once per process: this part runs in a child)
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import oracle
        import astar_pairwise_aligner_amd as pa
        from tests.test_gpu_engine import gpu_params
        from tests.test_restated_engine import variants
        from tests.util_seq import gen_pair
        oc = variants(oracle)[%r][0]
        pairs = [gen_pair(n, e, seed) for n, e, seed in [(300, 0.05, 1), (3000, 0.1, 3), (10000, 0.15, 4), (30000, 0.08, 6), (20000, 0.03, 5), (9000, 0.3, 8)]]
        b = pa.Batch(pairs, params=gpu_params(pa, oc))
        costs, cigars, _, _ = b.align()
        for (x, y), c, g in zip(pairs, costs, cigars):
            w = oracle.cpu_align(x, y, oc)
            assert (int(c), g) == w[:2]
        assert b.window_retries() >= 4 and b.trace_fallbacks() == 0, (b.window_retries(), b.trace_fallbacks())
        print("ok", b.window_retries())
    """) % (str(__import__("pathlib").Path(__file__).resolve().parent.parent), name)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]


def test_matches_from_host_threads_give_the_same(pa, oracle):
    """PA_GCSH_HOST_BUILD=1: the matches of GCSH come from host threads at creation (csrc/gcsh.hpp) instead of the GPU's build kernel -- also
    what a parameter set with a look-ahead beyond the kernel's LDS arrays (p > 14) gets.  Same cost, CIGAR string and statistics."""
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
        pairs = [gen_pair(n, e, seed) for n, e, seed in [(300, 0.05, 1), (3000, 0.1, 3), (10000, 0.15, 4), (30000, 0.08, 6), (9000, 0.3, 8)]]
        for oc in (oracle.params_full(), oracle.make_params(domain="astar", heuristic="gcsh", k=10, p=20, doubling="band", start="h0", factor=2.0,
                                                             block_width=256, sparse=True, incremental_doubling=True, dt_trace=True, max_g=40,
                                                             fr_drop=10, sparse_h=True, prune=True)):
            b = pa.Batch(pairs, params=gpu_params(pa, oc))
            costs, cigars, _, _ = b.align()
            st = b.pair_stats()
            assert b.full_info()["build_ms"] > 0  # (positive: host threads)
            for i, (x, y) in enumerate(pairs):
                w = oracle.cpu_align(x, y, oc)
                assert (int(costs[i]), cigars[i]) == w[:2] and all(st[i][k] == w[2][k] for k in KEYS), i
            assert b.trace_fallbacks() == 0
        print("ok")
    """) % str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    import os

    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, PA_GCSH_HOST_BUILD="1"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]
    # ... and p = 20 without the variable: the library itself falls back to the host's threads for that parameter set
    from tests.test_gpu_engine import gpu_params

    oc = oracle.make_params(domain="astar", heuristic="gcsh", k=10, p=20, doubling="band", start="h0", factor=2.0, block_width=256, sparse=True,
                            incremental_doubling=True, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True, prune=True)
    pairs = [gen_pair(4000, 0.1, 5), gen_pair(700, 0.02, 6)]
    check(pa, oracle, pairs, oc, max_fallbacks=0)


def test_pairs_the_match_builder_refuses_go_to_the_host_engine(pa, oracle):
    """A tandem repeat has more candidate matches than the GPU's match builder keeps room for (and more kept matches within reach of one
    search than its ring holds): the builder flags the pair, the band search hands it back, the host engine aligns it -- inside the same
    batch as ordinary pairs, with the same results as the CPU-kernel engine for all of them."""
    from astar_pairwise_aligner_amd import capi
    from tests.test_gpu_engine import gpu_params

    unit = rand_seq(7, seed=3)
    a = (unit * 600)[:3000]
    b = bytearray(a)
    for q in (100, 900, 1700, 2500):
        b[q] = ord("A") if b[q] != ord("A") else ord("C")
    rep = (a, bytes(b[:1200] + b[1230:]))
    with pytest.raises(capi.PaError):
        capi.gcsh_matches(rep[0], rep[1], 12, 14)  # (the builder alone refuses it)
    pairs = [gen_pair(2000, 0.05, 1), rep, gen_pair(5000, 0.1, 2), (rep[1], rep[0]), gen_pair(300, 0.2, 3)]
    oc = oracle.params_full()
    batch = pa.Batch(pairs, params=gpu_params(pa, oc))
    costs, cigars, _, _ = batch.align()
    stats = batch.pair_stats()
    assert 1 <= batch.trace_fallbacks() <= 2
    batch.close()
    for i, (x, y) in enumerate(pairs):
        w = oracle.cpu_align(x, y, oc)
        assert (int(costs[i]), cigars[i]) == w[:2], i
        assert {k: stats[i][k] for k in KEYS} == {k: w[2][k] for k in KEYS}, i


def harness_pairs():
    from tests.util_seq import PA_TEST_ES, PA_TEST_NS, mutate

    pairs = list(PA_TEST_PAIRS)
    for n in PA_TEST_NS:
        for e in PA_TEST_ES:
            pairs.append(gen_pair(n, e, seed=31415 + n * 7 + int(e * 1000)))
    for seed in range(6):
        base = rand_seq(700, seed=seed)
        ins = base[:300] + rand_seq(150, seed=100 + seed) + base[300:]
        dele = base[:200] + base[420:]
        rep = base[:350] + base[250:350] * 2 + base[350:]
        pairs += [(base, mutate(ins, 0.03, seed)), (base, mutate(dele, 0.03, seed)), (base, mutate(rep, 0.05, seed)), (ins, base), (rep, dele)]
    return pairs


SIMPLE_FAMILY = ["dijkstra", "sh12", "sh5", "gap_nosparseh", "gap_nodt", "gap_startgap", "gap_startzero_f15", "linear"]


@pytest.mark.parametrize("name", SIMPLE_FAMILY + [n for n in FULL_FAMILY if n != "full"])
def test_reference_harness_every_batched_configuration(pa, oracle, name):
    """The harness of the test above through every other parameter set the batch kernels take -- the configurations of
    astarpa2/src/tests.rs that are Domain::Astar over sparse blocks (band doubling over Dijkstra / GapCost / SH, GCSH with pruning,
    DT-trace on and off, incremental doubling) and their relatives: cost = Levenshtein, CIGAR and statistics = the CPU-kernel engine."""
    from tests.test_sweep_emu import variants as simple_variants

    oc = simple_variants(oracle)[name] if name in SIMPLE_FAMILY else variants(oracle)[name][0]
    pairs = harness_pairs()
    costs, _, _, _ = check(pa, oracle, pairs, oc)
    for (a, b), c in zip(pairs, costs):
        assert c == oracle.levenshtein(a, b), (len(a), len(b))


@pytest.mark.parametrize("preset", ["simple", "full"])
def test_reference_harness_as_one_batch(pa, oracle, preset):
    """pa-test's `test_aligner` (pa-test/src/lib.rs:7-40; astarpa2/src/tests.rs runs it per configuration) as ONE batch per preset: the 8
    literal pairs, the whole length x error-rate grid (fixed seeds; the reference samples a random quarter per run) and the structural
    error models (a long insertion, a long deletion, a repeated block).  The reference's acceptance rules -- the cost is the plain
    Levenshtein distance, the CIGAR is the engine's -- for every pair: cost, CIGAR string and statistics against the CPU-kernel engine."""
    from tests.util_seq import PA_TEST_ES, PA_TEST_NS

    pairs = harness_pairs()
    oc = oracle.params_full() if preset == "full" else oracle.params_simple()
    costs, cigars, _, _ = check(pa, oracle, pairs, oc)
    for (a, b), c in zip(pairs, costs):
        assert c == oracle.levenshtein(a, b), (len(a), len(b))
    assert len(pairs) == 8 + len(PA_TEST_NS) * len(PA_TEST_ES) + 30


def test_the_second_round_is_bounded_in_memory(pa, oracle, monkeypatch):
    """A batch of long-indel pairs: every band leaves its window of the block-column store, every pair is aligned again with full-height
    columns -- in sub-batches, one at a time, whose stores stay below the bound (forced down here to two pairs' worth): the largest one is
    reported, results equal the CPU-kernel engine, nothing goes to the host engine."""
    a = rand_seq(30_000, seed=23)
    pairs = []
    for t in range(10):
        cut = 3000 + 2000 * t
        pairs.append((a, a[:cut] + a[cut + 6000:]))  # 6000 columns deleted somewhere: ~94 words off the diagonal
    per_pair = (30_000 / 256 + 2) * ((24_000 + 63) // 64) * 16
    monkeypatch.setenv("PA_WINDOW_RETRY_BYTES", str(2.5 * per_pair))
    for prm, oc in ((pa.AstarPa2Params.simple(), oracle.params_simple()), (pa.AstarPa2Params.full(), oracle.params_full())):
        bt = pa.Batch(pairs, params=prm)
        costs, cigars, _, _ = bt.align()
        retries, peak, fallbacks = bt.window_retries(), bt.window_retry_bytes(), bt.trace_fallbacks()
        bt.close()
        assert retries == len(pairs) and fallbacks == 0, (retries, fallbacks)
        assert per_pair <= peak <= 2.5 * per_pair, (peak, per_pair)  # two pairs at a time: five sub-batches
        for (x, y), c, g in list(zip(pairs, costs, cigars))[::3]:
            assert (int(c), g) == oracle.cpu_align(x, y, oc)[:2]
