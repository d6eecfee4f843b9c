"""Batched alignment WITH traceback on the GPU (pa_batch_create_trace / pa_batch_align).

The device-side traceback restates Blocks::trace for sparse 256-column blocks without DT-trace
(astarpa2/src/blocks/trace.rs:21-228), so every cost AND every CIGAR string must equal what the engine over the CPU
oracle kernels returns for AstarPa2Params::nw() with front.sparse = true -- the parameter set `pa_params_batch_align`
names.  Sizes probe the block boundaries (n = 1, 255, 256, 257, 513: the last one makes the final sparse block a
single column, which the reference walks without a re-fill), empty sequences, long indels and high divergence."""
import numpy as np
import pytest

from tests.util_seq import PA_TEST_PAIRS, gen_pair, rand_seq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import astar_pairwise_aligner_amd as pa

    pa.require_gpu()
    return pa


def traced_params(oracle):
    return oracle.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=True,
                              incremental_doubling=False, dt_trace=False)


def check(pa, oracle, pairs, fallbacks=0):
    batch = pa.Batch(pairs, trace=True)
    costs, cigars, fwd_ms, trace_ms = batch.align()
    prm = traced_params(oracle)
    for (a, b), c, cg in zip(pairs, costs, cigars):
        want_cost, want_cigar, _ = oracle.cpu_align(a, b, prm)
        assert c == want_cost, (len(a), len(b))
        assert cg == want_cigar, (len(a), len(b), cg[:80], want_cigar[:80])
        assert oracle.cigar_verify(cg, a, b) == c
    assert batch.trace_fallbacks() == fallbacks
    costs2, cigars2, _, _ = batch.align()  # idempotent on resident inputs
    assert np.array_equal(costs, costs2) and cigars == cigars2
    batch.close()
    return fwd_ms, trace_ms


def test_small_and_boundaries(pa, oracle):
    pairs = list(PA_TEST_PAIRS)
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 258, 511, 512, 513, 769, 1000, 1025, 2049):
        for e in (0.0, 0.05, 0.2):
            pairs.append(gen_pair(n, e, seed=n * 13 + int(100 * e)))
    pairs += [(b"", b""), (b"ACGT", b""), (b"", b"ACGTA"), (b"A", b"A"), (b"A", b"C")]
    check(pa, oracle, pairs)


@pytest.mark.parametrize("k", [1, 2, 4, 8])
@pytest.mark.parametrize("mode", ["chain", "seq"])
def test_every_forward_shape_checkpoints_the_same_columns(pa, oracle, monkeypatch, k, mode):
    """The checkpointing forward pass exists for chained strips and for one-wavefront-per-pair, at every strip height."""
    monkeypatch.setenv("PA_STRIP_K", str(k))
    monkeypatch.setenv("PA_BATCH_MODE", mode)
    rows = 2048 * k
    pairs = [gen_pair(n, 0.07, seed=n + k) for n in (300, 1000, rows + 70, 2 * rows + 300)]
    pairs.append((rand_seq(600, seed=3), rand_seq(rows + 2500, seed=4)))
    check(pa, oracle, pairs)


def test_indels_and_unequal_lengths(pa, oracle):
    a = rand_seq(3000, seed=5)
    pairs = [
        (a, a[:1000] + a[1400:]),                       # 400-column deletion
        (a[:1000] + a[1400:], a),                       # 400-row insertion
        (a, a[:700] + rand_seq(300, seed=6) + a[700:]),  # foreign insert
        (rand_seq(700, seed=1), rand_seq(2300, seed=2)),  # unrelated, tall
        (rand_seq(2300, seed=3), rand_seq(70, seed=4)),   # unrelated, wide
        (a, a),
    ]
    check(pa, oracle, pairs)


def test_c4_like_mixed_divergence(pa, oracle):
    """C4 shape at a size the CPU engine checks in seconds: 10 kbp pairs, 1-15 % mixed divergence."""
    pairs = [gen_pair(10_000, e, seed=100 + i) for i, e in enumerate((0.01, 0.05, 0.10, 0.15) * 3)]
    check(pa, oracle, pairs)


def test_tall_refill_runs_as_several_strips(pa, oracle):
    """3000 and 5000 inserted bases that match nothing (poly-A into an A-free sequence) force vertical runs of that many rows
    inside one 256-column block: the re-fill is taller than one 2048-row strip and runs as chained strips inside the trace
    kernel.  A 9000-row run exceeds the per-pair scratch (8192 rows): that pair alone goes to the host engine.  Either way
    the answer is the reference's."""
    a = bytes(b"CGT"[x % 3] for x in rand_seq(1000, seed=21))
    big = bytes(b"CGT"[x % 3] for x in rand_seq(2000, seed=23))
    pairs = [(a, a[:900] + b"A" * 3000 + a[900:]), gen_pair(3000, 0.05, seed=13),
             (rand_seq(400, seed=11), rand_seq(9000, seed=12)),
             (big, big[:300] + b"A" * 5000 + big[300:])]
    check(pa, oracle, pairs, fallbacks=0)
    check(pa, oracle, [(big, big[:300] + b"A" * 9000 + big[300:1500] + b"A" * 5000 + big[1500:]), gen_pair(500, 0.1, seed=2)], fallbacks=1)


def test_100kbp_cost_and_valid_cigar(pa, oracle):
    """C2/C3 size: cost equals the full-DP oracle, the CIGAR verifies at that cost, and it equals the CPU-kernel engine's."""
    a, b = gen_pair(100_000, 0.05, seed=1)
    batch = pa.Batch([(a, b)], trace=True)
    costs, cigars, _, _ = batch.align()
    assert costs[0] == oracle.nw_cost(a, b, True)
    assert oracle.cigar_verify(cigars[0], a, b) == costs[0]
    want_cost, want_cigar, _ = oracle.cpu_align(a, b, traced_params(oracle))
    assert (costs[0], cigars[0]) == (want_cost, want_cigar)
    batch.close()


def test_edge_batches(pa, oracle):
    """Empty batch, a batch of empty pairs, an invalid base: same error behaviour as the cost-only batch."""
    b0 = pa.Batch([], trace=True)
    costs, cigars, _, _ = b0.align()
    assert len(costs) == 0 and cigars == []
    b0.close()
    b1 = pa.Batch([(b"", b""), (b"", b"ACG"), (b"TT", b"")], trace=True)
    costs, cigars, _, _ = b1.align()
    assert costs.tolist() == [0, 3, 2] and cigars == ["", "3I", "2D"]
    b1.close()
    with pytest.raises(ValueError):
        pa.Batch([(b"ACGTN", b"ACGT")], trace=True).align()
    with pytest.raises(pa.PaError):
        pa.Batch([(b"ACGT", b"ACGT")]).align()  # not a traced batch


def test_batch_align_equals_engine_with_pa_params_batch_align(pa):
    """`pa_params_batch_align` names the parameter set: the HIP block engine with it returns the same cost and CIGAR."""
    pairs = [gen_pair(n, 0.06, seed=n) for n in (100, 777, 3000)]
    got = pa.align_batch(pairs)
    p = pa.AstarPa2Params.nw()
    p.front.sparse = True
    al = p.make_aligner(True)
    for (a, b), (c, g) in zip(pairs, got):
        assert al.align(a, b) == (c, g)


def test_two_shards_in_one_process_on_one_device(pa, oracle):
    """pa_batch_align_multi with devices = {0, 0}: two host threads inside the library, each binding device 0 (pa_set_device is
    per thread) and running its own batch concurrently -- the per-(kernel, device) launch attributes, the per-thread device
    properties and the error text must not interfere.  Results equal the single-batch call, pair for pair."""
    pairs = [gen_pair(n, e, seed=n + int(100 * e)) for n in (300, 5000, 10_000, 20_000, 2049, 777) for e in (0.02, 0.1, 0.2)]
    pairs += [(b"", b"ACGT"), (b"ACGT", b"")]
    costs, cigars = pa.align_multi(pairs, [0, 0])
    single = pa.Batch(pairs, trace=True)
    want_costs, want_cigars, _, _ = single.align()
    single.close()
    assert costs.tolist() == want_costs.tolist() and cigars == want_cigars
    cost_only, none = pa.align_multi(pairs, [0, 0, 0], trace=False)
    assert none is None and cost_only.tolist() == want_costs.tolist()
    with pytest.raises(pa.PaError):
        pa.align_multi(pairs, [0, 99])


def test_work_queue_of_chunks_over_devices(pa, oracle, monkeypatch):
    """pa_batch_align_multi[_params] is a queue: chunks of pairs pulled by the device threads.  Forced down to 5 pairs per chunk so
    that every thread takes many; full-DP traced, cost-only, and the batched A*PA2 (with statistics) -- all equal to one batch."""
    from tests.test_sweep_emu import KEYS

    monkeypatch.setenv("PA_MULTI_CHUNK", "5")
    pairs = [gen_pair(n, e, seed=n + int(100 * e)) for n in (300, 5000, 10_000, 3000, 2049, 777, 1500, 64, 9000) for e in (0.02, 0.1, 0.2)]
    pairs += [(b"", b"ACGT"), (b"ACGT", b"")]
    single = pa.Batch(pairs, trace=True)
    want_costs, want_cigars, _, _ = single.align()
    single.close()
    costs, cigars = pa.align_multi(pairs, [0, 0, 0])
    assert costs.tolist() == want_costs.tolist() and cigars == want_cigars
    prm = pa.AstarPa2Params.simple()
    sb = pa.Batch(pairs, params=prm)
    a_costs, a_cigars, _, _ = sb.align()
    a_stats = sb.pair_stats()
    sb.close()
    costs, cigars, stats = pa.align_multi(pairs, [0, 0], params=prm, stats=True)
    assert costs.tolist() == a_costs.tolist() == want_costs.tolist() and cigars == a_cigars
    assert [{k: s[k] for k in KEYS} for s in stats] == [{k: s[k] for k in KEYS} for s in a_stats]


def test_few_long_pairs_over_devices_are_cut_at_the_byte_cap(pa, oracle, monkeypatch):
    """Few pairs over several devices are dealt out longest-processing-time-first, one bin per device -- and a bin is still cut where its
    block-column store would pass the cap (round 4's advisor finding: the dealt bins skipped the cap that the byte-bounded path applies).
    The cap is forced down to one pair's worth, so every bin becomes several chunks; results equal one batch."""
    pairs = [gen_pair(n, 0.08, seed=n) for n in (9000, 8000, 7000, 6000, 5000, 4000, 3000, 2000, 1000)]
    single = pa.Batch(pairs, trace=True)
    want_costs, want_cigars, _, _ = single.align()
    single.close()
    monkeypatch.setenv("PA_MULTI_CHUNK_BYTES", str((9000 / 256 + 2) * 141 * 16 + 1))
    costs, cigars = pa.align_multi(pairs, [0, 0])
    assert costs.tolist() == want_costs.tolist() and cigars == want_cigars
    prm = pa.AstarPa2Params.simple()
    costs, cigars, _ = pa.align_multi(pairs, [0, 0, 0], params=prm, stats=True)
    assert costs.tolist() == want_costs.tolist()
    assert all(oracle.cigar_verify(g, x, y) == c for g, (x, y), c in zip(cigars, pairs, costs.tolist()))


def test_concurrent_host_threads_through_the_c_abi(pa, oracle):
    """Python threads calling different entry points at once (ctypes releases the GIL): pa_align through the sweep, a cost-only
    batch and a traced batch, all on device 0."""
    import threading

    from tests.test_gpu_engine import gpu_params

    a, b = gen_pair(30_000, 0.05, seed=9)
    pairs = [gen_pair(4000, 0.1, seed=s) for s in range(40)]
    out = {}

    def t_engine():
        out["engine"] = [gpu_params(pa, oracle.params_simple()).make_aligner(True).align(a, b) for _ in range(3)]

    def t_batch():
        out["batch"] = [pa.Batch(pairs).run()[0].tolist() for _ in range(3)]

    def t_trace():
        bt = pa.Batch(pairs, trace=True)
        out["trace"] = [bt.align()[:2] for _ in range(2)]
        bt.close()

    ts = [threading.Thread(target=f) for f in (t_engine, t_batch, t_trace)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    want = oracle.cpu_align(a, b, oracle.params_simple())
    assert all((c, g) == (want[0], want[1]) for c, g in out["engine"])
    want_costs = [oracle.levenshtein(x, y) for x, y in pairs]
    assert all(c == want_costs for c in out["batch"])
    assert all(c.tolist() == want_costs for c, _ in out["trace"])
    assert all(oracle.cigar_verify(g, x, y) == w for _, gs in out["trace"] for g, (x, y), w in zip(gs, pairs, want_costs))


def dt_params(oracle):
    """Full-domain forward pass, the `simple` preset's traceback options (params.rs:70-96: dt_trace, max_g 40, fr_drop 10)."""
    return oracle.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=True,
                              incremental_doubling=False, dt_trace=True)


def check_dt(pa, oracle, pairs):
    from tests.test_gpu_engine import gpu_params

    prm = dt_params(oracle)
    batch = pa.Batch(pairs, trace=True, trace_params=gpu_params(pa, prm))
    costs, cigars, _, trace_ms = batch.align()
    for (a, b), c, cg in zip(pairs, costs, cigars):
        want_cost, want_cigar, _ = oracle.cpu_align(a, b, prm)
        assert c == want_cost, (len(a), len(b))
        assert cg == want_cigar, (len(a), len(b), cg[:80], want_cigar[:80])
    assert batch.trace_fallbacks() == 0
    batch.close()
    return trace_ms


def test_dt_trace_small_and_boundaries(pa, oracle):
    """Device-side DT-trace (blocks/trace.rs:231-416) in the batched traceback: cost and CIGAR string equal the engine over the
    CPU oracle kernels with the same `front` -- which takes the diagonal-transition path wherever it succeeds and re-fills the
    block where it gives up (max_g, the midpoint early-out, fr_drop)."""
    pairs = list(PA_TEST_PAIRS)
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 258, 511, 512, 513, 769, 1000, 1025, 2049):
        for e in (0.0, 0.05, 0.2):
            pairs.append(gen_pair(n, e, seed=n * 13 + int(100 * e)))
    pairs += [(b"", b""), (b"ACGT", b""), (b"", b"ACGTA"), (b"A", b"A"), (b"A", b"C")]
    check_dt(pa, oracle, pairs)


def test_dt_trace_divergence_mix_and_indels(pa, oracle):
    a = rand_seq(3000, seed=5)
    pairs = [gen_pair(10_000, e, seed=300 + i) for i, e in enumerate((0.01, 0.03, 0.05, 0.08, 0.10, 0.12, 0.15, 0.25, 0.4))]
    pairs += [(a, a[:1000] + a[1400:]), (a[:1000] + a[1400:], a), (a, a[:700] + rand_seq(300, seed=6) + a[700:]),
              (rand_seq(700, seed=1), rand_seq(2300, seed=2)), (rand_seq(2300, seed=3), rand_seq(70, seed=4)), (a, a)]
    pairs += [gen_pair(n, e, seed=n + 7) for n in (5000, 20_000, 40_000) for e in (0.02, 0.1)]
    check_dt(pa, oracle, pairs)


def test_trace_params_are_validated(pa, oracle):
    from tests.test_gpu_engine import gpu_params

    pairs = [gen_pair(500, 0.1, seed=1)]
    bad = gpu_params(pa, oracle.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=True, dt_trace=True))
    bad.front.max_g = 41  # the device table is sized for the presets' max_g = 40
    with pytest.raises(pa.PaError):
        pa.Batch(pairs, trace=True, trace_params=bad)
    dense = gpu_params(pa, oracle.make_params(domain="full", heuristic="none", doubling="none", block_width=256, sparse=False))
    with pytest.raises(pa.PaError):
        pa.Batch(pairs, trace=True, trace_params=dense)


@pytest.mark.gpu
def test_view_and_c_string_entry_points_agree(pa, oracle):
    """pa_batch_align_view (texts left in the plan's host buffer: what Batch.align() calls) and pa_batch_align (one malloc'ed C string per
    pair: Batch.align_c_strings()) return the same costs and CIGARs -- full-DP traced batch, both A*PA2 families, a batch small enough for
    the host-driven route, and one whose repeat-rich pair goes to the host engine (a string of its own inside the view)."""
    from tests.test_gpu_engine import gpu_params
    from tests.util_seq import gen_pair, rand_seq

    pairs = [gen_pair(300 + 53 * (i % 40), (0.02, 0.08, 0.2)[i % 3], 7000 + i) for i in range(150)]
    unit = rand_seq(7, seed=3)
    a = (unit * 600)[:3000]
    b = bytearray(a)
    for q in (100, 900, 1700, 2500):
        b[q] = ord("A") if b[q] != ord("A") else ord("C")
    with_repeat = pairs[:80] + [(a, bytes(b[:1200] + b[1230:]))] + pairs[80:]
    batches = [
        pa.Batch(pairs, trace=True),
        pa.Batch(pairs, params=gpu_params(pa, oracle.params_simple())),
        pa.Batch(with_repeat, params=gpu_params(pa, oracle.params_full())),
        pa.Batch(pairs[:3], params=gpu_params(pa, oracle.params_full())),
    ]
    for bt in batches:
        for rep in range(2):
            c1, g1, _, _ = bt.align()
            c2, g2, _, _ = bt.align_c_strings()
            c3, g3, _, _ = bt.align()
            assert list(c1) == list(c2) == list(c3)
            assert g1 == g2 == g3
            assert all(isinstance(x, str) and x for x in g1)
        bt.close()
    full = oracle.params_full()
    want = oracle.cpu_align(*with_repeat[80], full)
    bt = pa.Batch(with_repeat, params=gpu_params(pa, full))
    costs, cigars, _, _ = bt.align()
    assert (int(costs[80]), cigars[80]) == want[:2] and bt.trace_fallbacks() >= 1
    bt.close()


@pytest.mark.gpu
def test_c_batch_view_program(tmp_path):
    """tests/c_abi/batch_view_check.c: pa_batch_align_view from plain C beside pa_batch_align and pa_align, both presets."""
    import os
    import shutil
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    libdir = root / "astar-pairwise-aligner_amd"
    gcc = shutil.which("gcc") or shutil.which("cc")
    if gcc is None or not (libdir / "libastarpa_c_hip.so").exists():
        pytest.skip("no C compiler or library")
    exe = tmp_path / "batch_view_check"
    subprocess.run([gcc, str(root / "tests" / "c_abi" / "batch_view_check.c"), "-I", str(root / "include"), "-L", str(libdir),
                    "-lastarpa_c_hip", "-Wl,-rpath," + str(libdir), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, LD_LIBRARY_PATH=str(libdir) + ":" + os.environ.get("LD_LIBRARY_PATH", "")))
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("batch_view_check ok pairs=48 chars=")


@pytest.mark.gpu
def test_reference_harness_through_the_full_dp_batches(pa, oracle):
    """pa-test's `test_aligner` set (the 8 literal pairs, the whole length x error-rate grid with fixed seeds, the structural error models:
    tests/test_gpu_apa2_full.py harness_pairs) through the full-DP traced batch, re-fill-only and with the DT-trace options: cost = plain
    Levenshtein, CIGAR = the engine's with the same `front` (astarpa2/src/tests.rs `full` / `dt_trace` configurations, batched)."""
    from tests.test_gpu_apa2_full import harness_pairs

    pairs = harness_pairs()
    check(pa, oracle, pairs)
    check_dt(pa, oracle, pairs)
    batch = pa.Batch(pairs, trace=True)
    costs, _, _, _ = batch.align()
    batch.close()
    for (a, b), c in zip(pairs, costs):
        assert c == oracle.levenshtein(a, b), (len(a), len(b))
