"""GPU tests at BASELINE.json's full sizes (configs C1..C5).

Where the oracle finishes in seconds (one 100 kbp pair = 0.2 s of AVX2) results are compared bit-exactly;
beyond that, size-independent properties of the edit distance are used: symmetry d(a,b) = d(b,a), identity
d(a,a) = 0, d(a, a + suffix) = |suffix|, the k-substitution bound, idempotence of a second pass, and a checksum
of costs against an oracle sample."""
import numpy as np
import pytest

from tests.util_seq import gen_pair, mutate, rand_seq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import astar_pairwise_aligner_amd as pa

    pa.require_gpu()
    return pa


def test_c1_1kbp_pairs_vs_plain_levenshtein(pa, oracle):
    """C1: 1 kbp, 5 % uniform error, seeds 1..100 -- plumbing check against the plain O(nm) DP."""
    pairs = [gen_pair(1000, 0.05, seed=s) for s in range(1, 101)]
    costs, _ = pa.Batch(pairs).run()
    assert costs.tolist() == [oracle.levenshtein(a, b) for a, b in pairs]


def test_c2_100kbp_full_dp_bit_exact(pa, oracle):
    """C2: 100 kbp x 100 kbp, 5 %, full DP, seeds 1..3: cost identical to the CPU port of the reference schedule."""
    pairs = [gen_pair(100_000, 0.05, seed=s) for s in (1, 2, 3)]
    batch = pa.Batch(pairs + [(b, a) for a, b in pairs] + [(pairs[0][0], pairs[0][0])])
    costs, ms = batch.run()
    want = [oracle.nw_cost(a, b, True) for a, b in pairs]
    assert costs[:3].tolist() == want
    assert costs[3:6].tolist() == want          # symmetry
    assert costs[6] == 0                        # identity
    costs2, _ = batch.run()                     # idempotent on resident inputs
    assert np.array_equal(costs, costs2)


def test_c3_100kbp_astarpa2_simple_with_trace(pa, oracle):
    """C3: A*PA2-simple (GapCost band doubling, sparse blocks, DT trace) on a 100 kbp pair, traceback on."""
    a, b = gen_pair(100_000, 0.05, seed=1)
    want = oracle.nw_cost(a, b, True)
    cost, cigar, stats = pa.AstarPa2Params.simple().make_aligner(True).align_with_stats(a, b)
    assert cost == want
    assert oracle.cigar_verify(cigar, a, b) == want
    want_cost, want_cigar, want_stats = oracle.cpu_align(a, b, oracle.params_simple())
    assert (cost, cigar) == (want_cost, want_cigar)
    for k in ("num_blocks", "computed_lanes", "f_max_tries", "dt_trace_tries", "fill_tries"):
        assert stats[k] == want_stats[k], k
    # A*PA2-full: GCSH seed heuristic (k=12, p=14) with pruning and incremental doubling
    cost_f, cigar_f, stats_f = pa.AstarPa2Params.full().make_aligner(True).align_with_stats(a, b)
    want_f = oracle.cpu_align(a, b, oracle.params_full())
    assert (cost_f, cigar_f) == (want_f[0], want_f[1]) and cost_f == want
    assert oracle.cigar_verify(cigar_f, a, b) == want
    assert stats_f["computed_lanes"] == want_f[2]["computed_lanes"] < stats["computed_lanes"]


def test_c4_batch_10kbp_mixed_divergence(pa, oracle):
    """C4 (scaled to one GPU test run): 2000 pairs of 10 kbp, divergence drawn from {1,5,10,15} % by pair index."""
    divs = (0.01, 0.05, 0.10, 0.15)
    pairs = [gen_pair(10_000, divs[i % 4], seed=i) for i in range(2000)]
    costs, _ = pa.Batch(pairs).run()
    sample = range(0, 2000, 40)
    assert [int(costs[i]) for i in sample] == [oracle.nw_cost(*pairs[i], True) for i in sample]
    # k edits never cost more than k; more divergence never makes these pairs closer on average
    for i, (a, b) in enumerate(pairs[:200]):
        assert 0 < costs[i] <= int(divs[i % 4] * len(a))
    by_div = [np.mean(costs[j::4]) for j in range(4)]
    assert by_div == sorted(by_div)


def test_c4_full_10000_pairs_with_traceback(pa, oracle):
    """C4 as BASELINE.json states it: 10 000 pairs of 10 kbp, divergence 1 / 5 / 10 / 15 % by pair index, global alignment WITH
    traceback in one pa_batch_align call.  Every CIGAR replays to its pair at exactly the reported cost; a 1 % sample equals the
    CPU engine over the oracle kernels (cost and CIGAR string, the parameter set `pa_params_batch_align` names); the costs equal
    the cost-only batch (an independent kernel path); no pair fell back to the host."""
    from tests.test_gpu_batch_align import traced_params

    divs = (0.01, 0.05, 0.10, 0.15)
    pairs = [gen_pair(10_000, divs[i % 4], seed=2_000_000 + i) for i in range(10_000)]
    batch = pa.Batch(pairs, trace=True)
    costs, cigars, _, _ = batch.align()
    assert batch.trace_fallbacks() == 0
    batch.close()
    assert len(cigars) == 10_000
    for (a, b), c, cg in zip(pairs, costs, cigars):
        assert oracle.cigar_verify(cg, a, b) == c
    prm = traced_params(oracle)
    for i in range(37, 10_000, 100):  # 100 pairs, every divergence class
        want_cost, want_cigar, _ = oracle.cpu_align(*pairs[i], prm)
        assert (int(costs[i]), cigars[i]) == (want_cost, want_cigar), i
    plain, _ = pa.Batch(pairs).run()
    assert np.array_equal(np.asarray(costs), np.asarray(plain))


def test_c4_full_10000_pairs_astarpa2_simple(pa, oracle):
    """C4 through the batched A*PA2 (pa_batch_create_params, AstarPa2Params::simple()): ONE wavefront runs each pair's band search.
    For EVERY one of the 10 000 pairs the cost, the CIGAR string and all twelve statistics equal the host engine over the CPU
    oracle kernels -- what a loop over pa_align / astarpa2_simple returns; no pair fell back to the host."""
    from concurrent.futures import ThreadPoolExecutor

    from tests.test_sweep_emu import KEYS

    divs = (0.01, 0.05, 0.10, 0.15)
    pairs = [gen_pair(10_000, divs[i % 4], seed=2_000_000 + i) for i in range(10_000)]
    batch = pa.Batch(pairs, params=pa.AstarPa2Params.simple())
    costs, cigars, _, _ = batch.align()
    stats = batch.pair_stats()
    assert batch.trace_fallbacks() == 0
    batch.close()
    prm = oracle.params_simple()
    with ThreadPoolExecutor(max_workers=16) as ex:  # (ctypes releases the GIL: the CPU engine runs on several cores)
        want = list(ex.map(lambda p: oracle.cpu_align(p[0], p[1], prm), pairs))
    for i, (w, c, cg, st) in enumerate(zip(want, costs, cigars, stats)):
        assert (int(c), cg) == (w[0], w[1]), i
        assert {k: st[k] for k in KEYS} == {k: w[2][k] for k in KEYS}, i


def test_c3_batch_512_pairs_astarpa2_simple(pa, oracle):
    """512 pairs of 100 kbp at 5 % (BASELINE C3's pair, many of them) through the batched A*PA2: cost, CIGAR string and all twelve
    statistics of every pair equal the host engine over the CPU oracle kernels."""
    from concurrent.futures import ThreadPoolExecutor

    from tests.test_sweep_emu import KEYS

    pairs = [gen_pair(100_000, 0.05, seed=3_000_000 + i) for i in range(512)]
    batch = pa.Batch(pairs, params=pa.AstarPa2Params.simple())
    costs, cigars, _, _ = batch.align()
    stats = batch.pair_stats()
    assert batch.trace_fallbacks() == 0
    batch.close()
    prm = oracle.params_simple()
    with ThreadPoolExecutor(max_workers=16) as ex:
        want = list(ex.map(lambda p: oracle.cpu_align(p[0], p[1], prm), pairs))
    for i, (w, c, cg, st) in enumerate(zip(want, costs, cigars, stats)):
        assert (int(c), cg) == (w[0], w[1]), i
        assert {k: st[k] for k in KEYS} == {k: w[2][k] for k in KEYS}, i


def _full_against_both_oracles(pa, oracle, pairs, restated_sample):
    """Every pair through the batched `full` preset: cost, CIGAR string and twelve statistics equal the host engine over the CPU oracle
    kernels (`oracle.cpu_align(params_full())`); a sample also equals the second, independent restatement (oracle/astarpa2_restated.py)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import astarpa2_restated as restated
    from tests.test_restated_engine import variants
    from tests.test_sweep_emu import KEYS

    batch = pa.Batch(pairs, params=pa.AstarPa2Params.full())
    costs, cigars, _, _ = batch.align()
    stats = batch.pair_stats()
    assert batch.trace_fallbacks() == 0
    batch.close()
    prm, kw = variants(oracle)["full"]
    with ThreadPoolExecutor(max_workers=16) as ex:  # (ctypes releases the GIL: the CPU engine runs on several cores)
        want = list(ex.map(lambda p: oracle.cpu_align(p[0], p[1], prm), pairs))
    for i, (w, c, cg, st) in enumerate(zip(want, costs, cigars, stats)):
        assert (int(c), cg) == (w[0], w[1]), i
        assert {k: st[k] for k in KEYS} == {k: w[2][k] for k in KEYS}, i
    for i in restated_sample:
        got = restated.align(pairs[i][0], pairs[i][1], **kw)
        assert (int(costs[i]), cigars[i]) == got[:2], i
        assert all(stats[i][k] == got[2][k] for k in KEYS if k != "sanity_violations"), i
    return stats


def test_c4_full_10000_pairs_astarpa2_full(pa, oracle):
    """C4 through the batched A*PA2 with AstarPa2Params::full() (GCSH + pruning + incremental doubling on the device): EVERY one of the
    10 000 pairs against the CPU-kernel engine, 104 of them (every divergence class) against the second restatement as well."""
    divs = (0.01, 0.05, 0.10, 0.15)
    pairs = [gen_pair(10_000, divs[i % 4], seed=2_000_000 + i) for i in range(10_000)]
    stats = _full_against_both_oracles(pa, oracle, pairs, restated_sample=range(3, 10_000, 97))
    assert sum(s["f_max_tries"] > 1 for s in stats) > 1000  # (the 15 % pairs need a second pass: contours re-derived, three-range splits)


def test_c3_batch_512_pairs_astarpa2_full(pa, oracle):
    """512 pairs of 100 kbp at 5 % (BASELINE C3's pair, many of them) through the batched `full`; sixteen of them also against the restatement."""
    pairs = [gen_pair(100_000, 0.05, seed=3_000_000 + i) for i in range(512)]
    _full_against_both_oracles(pa, oracle, pairs, restated_sample=range(0, 512, 32))


def test_c4_properties_suffix_and_substitutions(pa):
    a = rand_seq(50_000, seed=77)
    suffix = rand_seq(1234, seed=78)
    sub = bytearray(a)
    for p in range(100, 50_000, 500):  # 100 isolated substitutions
        sub[p] = ord("A") if sub[p] != ord("A") else ord("C")
    costs, _ = pa.Batch([(a, a + suffix), (a + suffix, a), (a, bytes(sub)), (a, a)]).run()
    assert costs.tolist() == [1234, 1234, 100, 0]


def test_c5_long_pair_properties(pa, oracle):
    """C5-shaped stress (1 Mbp here; the 10 Mbp run is in profiles/README.md): full DP through ~490 chained strips.
    d(a,b) = d(b,a), bounded by the number of edits, and equal to the banded CPU engine's answer."""
    a = rand_seq(1_000_000, seed=5)
    b = mutate(a, 0.01, seed=5)
    costs, _ = pa.Batch([(a, b), (b, a), (a, a + b"ACGT" * 250)]).run()
    assert costs[0] == costs[1]
    assert 0 < costs[0] <= 10_000
    assert costs[2] == 1000
    # the banded CPU engine agrees on a 300 kbp prefix (seconds of CPU)
    a3, b3 = a[:300_000], mutate(a[:300_000], 0.01, seed=6)
    want, _, _ = oracle.cpu_align(a3, b3, oracle.params_simple(), trace=False)
    assert pa.Batch([(a3, b3)]).run()[0][0] == want


def test_bench_two_rank_code_path_dry_run(tmp_path):
    """bench.py's N>1 branch (barrier, max-over-ranks time, checksum reduction, rank-0-only report) on one GPU: both ranks
    use GPU 0 and gloo carries the collectives (PA_BENCH_DRY_MULTI=1).  Only the plumbing is checked, not the numbers."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, PA_BENCH_DRY_MULTI="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "8",
           "--seq-len", "20000", "--no-cpu-baseline", "--no-single-pair", "--no-c4", "--no-banded", "--c4-pairs", "300", "--c4-strong-pairs", "900"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["pairs_per_gpu"] == 8
    # the C4 strong-scaling leg: both ranks aligned their shard with traceback and every rank got all results
    assert j["c4_sharded"]["n_gpus"] == 2 and j["c4_sharded"]["scaling"] == "strong" and j["c4_sharded"]["pairs_per_sec"] > 0
    assert j["c4_sharded"]["cost_checksum"] > 0 and j["c4_sharded"]["cigar_bytes"] > 300
    # round 6: the strong leg repeats the C4 pairs up to --c4-strong-pairs (the same work for every N), reports where every rank's time went,
    # caps the host threads per rank at the node's share, and a WEAK leg (own pairs per rank) stands next to it
    cs = j["c4_sharded"]
    assert cs["pairs"] == 900 and cs["ms_repetitions"]["reps"] == 3 and cs["host_threads_per_rank"] >= 1
    assert len(cs["rank_busy_s"]) == 2 and len(cs["rank_timing_s"]) == 2
    for t in cs["rank_timing_s"]:
        assert {"plan_s", "queue_s", "compute_s", "gather_s", "chunks", "pairs", "total_s"} <= set(t)
    assert sum(t["pairs"] for t in cs["rank_timing_s"]) == 900
    assert cs["weak"]["scaling"] == "weak" and cs["weak"]["n_gpus"] == 2 and cs["weak"]["pairs_per_sec"] > 0 and cs["weak"]["ms_repetitions"]["reps"] == 3
    assert cs["astarpa2_simple"]["pairs"] == 900
