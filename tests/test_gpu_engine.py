"""GPU parity of the A*PA2 engine: the HIP-backed engine (libastarpa_c_hip.so, through the C ABI) must
reproduce the engine over the CPU oracle kernels exactly -- cost, CIGAR string and every band statistic --
and satisfy the reference's acceptance rules (cost == Levenshtein, CIGAR valid; pa-test/src/lib.rs:65-99)."""
import pytest

from tests.util_seq import PA_TEST_PAIRS, gen_pair

pytestmark = pytest.mark.gpu

STAT_KEYS = ["num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "dt_trace_tries",
             "dt_trace_success", "dt_trace_fallback", "fill_tries", "fill_success", "fill_fallback", "f_max_tries",
             "sanity_violations"]


@pytest.fixture(scope="module")
def pa():
    import astar_pairwise_aligner_amd as pa

    pa.require_gpu()
    return pa


def gpu_params(pa, oc):
    """oracle ctypes params -> package dataclass with the same fields."""
    inv = lambda d, v: [k for k, x in d.items() if x == v][0]
    from astar_pairwise_aligner_amd import aligner as al

    f = oc.front
    return pa.AstarPa2Params(
        domain=inv(al.DOMAIN, oc.domain), heuristic=inv(al.HEURISTIC, oc.heuristic), k=oc.heuristic_k, p=oc.heuristic_p, doubling=inv(al.DOUBLING, oc.doubling),
        doubling_start=inv(al.START, oc.doubling_start), factor=oc.factor, delta=oc.delta, block_width=oc.block_width,
        front=pa.BlockParams(bool(f.sparse), bool(f.simd), bool(f.no_ilp), bool(f.incremental_doubling), bool(f.dt_trace),
                             f.max_g, f.fr_drop), sparse_h=bool(oc.sparse_h), prune=bool(oc.prune))


def both(pa, oracle, a, b, oc, trace=True):
    want_cost, want_cigar, want_stats = oracle.cpu_align(a, b, oc, trace=trace)
    cost, cigar, stats = gpu_params(pa, oc).make_aligner(trace).align_with_stats(a, b)
    assert cost == want_cost
    assert cigar == want_cigar
    assert {k: stats[k] for k in STAT_KEYS} == {k: want_stats[k] for k in STAT_KEYS}
    return cost, cigar


def test_c_abi_example(pa, oracle):
    """astarpa-c/example.c:8-33 / example.cpp:8-20: cost 2 through all four entry points; CIGAR valid."""
    a, b = b"ACTCGCT", b"AACTCGTT"
    for sym, extra in (("astarpa2_simple", ()), ("astarpa2_full", ()), ("astarpa", ()), ("astarpa_gcsh", (1, 15, False))):
        cost, cigar = pa.c_abi_align(sym, a, b, *extra)
        assert cost == 2, sym
        assert oracle.cigar_verify(cigar, a, b) == 2, (sym, cigar)


def test_presets_on_pa_test_pairs(pa, oracle):
    for a, b in PA_TEST_PAIRS:
        want = oracle.levenshtein(a, b)
        for oc in (oracle.params_nw(), oracle.params_simple(), oracle.params_full()):
            cost, cigar = both(pa, oracle, a, b, oc)
            assert cost == want and oracle.cigar_verify(cigar, a, b) == want


@pytest.mark.parametrize("name", ["preset_simple", "preset_nw", "incremental_doubling", "dt_trace_gapgap", "band_doubling_dijkstra",
                                  "band_doubling_sh", "sh_k12_w256", "gcsh_k6_dt", "preset_full"])
def test_configs_small_grid(pa, oracle, name):
    from tests.test_engine_cpu import configs

    oc = configs(oracle)[name]
    for n in (0, 1, 17, 64, 100, 255, 256, 257, 300, 513):
        for e in (0.0, 0.05, 0.2, 1.0):
            a, b = gen_pair(n, e, seed=n * 13 + int(e * 100))
            cost, cigar = both(pa, oracle, a, b, oc)
            assert cost == oracle.levenshtein(a, b)
            assert oracle.cigar_verify(cigar, a, b) == cost


def test_simple_medium_pairs(pa, oracle):
    """Several band-doubling iterations, multi-strip bands, DT trace + block re-fill."""
    for n, e, seed in [(3000, 0.05, 1), (10000, 0.10, 2), (20000, 0.03, 3), (5000, 0.30, 4)]:
        a, b = gen_pair(n, e, seed)
        for oc in (oracle.params_simple(), oracle.params_full()):
            cost, cigar = both(pa, oracle, a, b, oc)
            assert cost == oracle.nw_cost(a, b, True)
            assert oracle.cigar_verify(cigar, a, b) == cost


def test_incremental_medium_pair(pa, oracle):
    from tests.test_engine_cpu import configs

    a, b = gen_pair(6000, 0.08, 11)
    both(pa, oracle, a, b, configs(oracle)["incremental_doubling"])
    both(pa, oracle, a, b, configs(oracle)["simple_scalar_noilp"])


def test_cost_only_mode(pa, oracle):
    a, b = gen_pair(4000, 0.1, 5)
    for oc in (oracle.params_nw(), oracle.params_simple()):
        want, _, _ = oracle.cpu_align(a, b, oc, trace=False)
        assert gpu_params(pa, oc).make_aligner(False).align(a, b) == (want, None)
    # nw cost-only runs as ONE launch of chained strips; cost and every band statistic still equal the block-by-block engine
    for a, b in (gen_pair(4000, 0.1, 5), gen_pair(257, 0.3, 6), gen_pair(70_000, 0.04, 7), (b"ACGT", b"A")):
        both(pa, oracle, a, b, oracle.params_nw(), trace=False)


def test_reference_cost_only_mode_on_the_gpu(pa, oracle):
    """pa_set_reference_cost_only(1): pa_align(trace = 0) runs the REFERENCE's cost-only arm (astarpa2/src/blocks.rs:252-277, one block updated
    in place) with its rectangles on the GPU -- cost, passes and block counters of both restatements of that arm, the upper bound 11353 of
    tests/golden/cost_only_pair.json (distance 11325) included.  Default: the distance over the traced band (include/pa_astarpa2.h)."""
    import json
    from pathlib import Path

    from astar_pairwise_aligner_amd import capi
    from oracle import astarpa2_restated as restated
    from tests.test_restated_engine import variants

    keys = ["num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "f_max_tries"]
    j = json.loads((Path(__file__).resolve().parent / "golden" / "cost_only_pair.json").read_text())
    a, b = j["a"].encode(), j["b"].encode()
    prm = oracle.make_params(domain="astar", heuristic="sh", k=12, doubling="band", start="h0", factor=2.0, block_width=256, sparse=True,
                             incremental_doubling=False, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True)
    al = gpu_params(pa, prm).make_aligner(False)
    assert al.align(a, b) == (11325, None)  # the default: the distance
    capi.load().pa_set_reference_cost_only(1)
    try:
        cost, cigar, stats = al.align_with_stats(a, b)
        want = oracle.cpu_align(a, b, prm, trace=False)
        second = restated.align(a, b, heuristic="sh", k=12, trace=False)
        assert cost == want[0] == second[0] == 11353 and cigar is None
        assert {k: stats[k] for k in keys} == {k: want[2][k] for k in keys} == {k: second[2][k] for k in keys}
        vs = variants(oracle)
        for name, (n, e, seed) in [("simple", (4000, 0.1, 5)), ("sh12", (9000, 0.2, 6)), ("dijkstra", (3000, 0.05, 7)), ("full", (8000, 0.1, 8)),
                                   ("gap_incr", (5000, 0.15, 9)), ("linear300", (2500, 0.1, 10))]:
            oc, kw = vs[name]
            a2, b2 = gen_pair(n, e, seed)
            cost, cigar, stats = gpu_params(pa, oc).make_aligner(False).align_with_stats(a2, b2)
            want = oracle.cpu_align(a2, b2, oc, trace=False)
            assert (cost, cigar) == (want[0], None) and {k: stats[k] for k in keys} == {k: want[2][k] for k in keys}, name
    finally:
        capi.load().pa_set_reference_cost_only(0)
    assert al.align(a, b) == (11325, None)


def test_invalid_base_raises(pa):
    with pytest.raises(ValueError):
        pa.astarpa2_simple(b"ACGTN", b"ACGT")


def test_c_program_links_and_runs(tmp_path):
    """A plain C program built with gcc against include/astarpa.h and libastarpa_c_hip.so: the drop-in for astarpa-c."""
    import os
    import re
    import shutil
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    libdir = root / "astar-pairwise-aligner_amd"
    gcc = shutil.which("gcc") or shutil.which("cc")
    if gcc is None or not (libdir / "libastarpa_c_hip.so").exists():
        pytest.skip("no C compiler or library")
    exe = tmp_path / "link_check"
    subprocess.run([gcc, str(root / "tests" / "c_abi" / "link_check.c"), "-I", str(root / "include"), "-L", str(libdir),
                    "-lastarpa_c_hip", "-Wl,-rpath," + str(libdir), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, LD_LIBRARY_PATH=str(libdir) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))).stdout
    lines = out.strip().splitlines()
    assert [l.split()[0] for l in lines] == ["astarpa2_simple", "astarpa2_full", "astarpa", "astarpa_gcsh"]
    for l in lines:
        name, cost, cigar, n = l.split()
        assert cost == "2" and int(n) == len(cigar) and re.fullmatch(r"(\d*[=XID])+", cigar)


def test_c_layout_program(tmp_path):
    """tests/c_abi/layout_check.c from plain C: struct layouts pinned with _Static_assert, pa_params_{nw,simple,full} + pa_align and
    the operator boundary (pa_bp_profile_build / pa_bp_compute) on the reference's example pair."""
    import os
    import shutil
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    libdir = root / "astar-pairwise-aligner_amd"
    gcc = shutil.which("gcc") or shutil.which("cc")
    if gcc is None or not (libdir / "libastarpa_c_hip.so").exists():
        pytest.skip("no C compiler or library")
    exe = tmp_path / "layout_check"
    subprocess.run([gcc, str(root / "tests" / "c_abi" / "layout_check.c"), "-I", str(root / "include"), "-L", str(libdir),
                    "-lastarpa_c_hip", "-Wl,-rpath," + str(libdir), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, LD_LIBRARY_PATH=str(libdir) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))).stdout
    lines = out.strip().splitlines()
    assert [l.split()[1] for l in lines[:3]] == ["nw", "simple", "full"]
    assert all("rc=0 cost=2" in l and "block_width=256" in l for l in lines[:3])
    assert lines[3] == "pa_bp_compute sum=%d cost=2" % int(lines[3].split("sum=")[1].split()[0])


def test_c_ctx_program(tmp_path, oracle):
    """tests/c_abi/ctx_check.c: the device-resident operator handles driven from plain C the way a host engine calls them -- create,
    an Output / Input / Update chain over three column blocks, fill, destroy -- against pa_bp_compute on the whole rectangle."""
    import os
    import shutil
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    libdir = root / "astar-pairwise-aligner_amd"
    gcc = shutil.which("gcc") or shutil.which("cc")
    if gcc is None or not (libdir / "libastarpa_c_hip.so").exists():
        pytest.skip("no C compiler or library")
    exe = tmp_path / "ctx_check"
    subprocess.run([gcc, str(root / "tests" / "c_abi" / "ctx_check.c"), "-I", str(root / "include"), "-L", str(libdir),
                    "-lastarpa_c_hip", "-Wl,-rpath," + str(libdir), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, LD_LIBRARY_PATH=str(libdir) + ":" + os.environ.get("LD_LIBRARY_PATH", "")))
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ctx_check ok cost=")
    # the same pair, generated the same way, through the oracle
    s, a = 12345, bytearray()
    for _ in range(600):
        s = (s * 1103515245 + 12345) & 0xFFFFFFFF
        a.append(b"ACGT"[(s >> 16) & 3])
    b = bytearray(a[j] if j < 600 else 65 for j in range(500))
    for j in range(7, 500, 23):
        b[j] = ord("C") if b[j] == ord("A") else ord("A")
    assert int(r.stdout.split("=")[1]) == oracle.levenshtein(bytes(a), bytes(b))


def test_c_abi_is_reentrant_across_threads(pa, oracle):
    """astarpa-c is stateless and re-entrant: several host threads align different pairs at the same time."""
    import threading

    from tests.util_seq import gen_pair

    pairs = [gen_pair(3000 + 257 * t, 0.03 + 0.02 * t, seed=90 + t) for t in range(6)]
    want = [oracle.levenshtein(a, b) for a, b in pairs]
    results = [None] * len(pairs)
    errors = []

    def work(t):
        try:
            out = []
            for _ in range(3):
                out.append(pa.c_abi_align("astarpa2_simple" if t % 2 else "astarpa2_full", *pairs[t]))
            results[t] = out
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(len(pairs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    for t, out in enumerate(results):
        assert out is not None
        for cost, cigar in out:
            assert cost == want[t] and oracle.cigar_verify(cigar, *pairs[t]) == cost
        assert len({c for _, c in out}) == 1  # deterministic


def test_concurrent_callers_are_combined_and_get_the_single_call_results(pa, oracle, monkeypatch):
    """Round 5: callers that are inside pa_align / an astarpa-c symbol at the same time are combined into one batch on the GPU
    (csrc/engine_hip.hip combine_align).  Sixteen threads, both presets through the drop-in symbols and through pa_align with statistics:
    EVERY result equals what the same call returns when it is made alone (the single-pair route), and the counters say calls were combined."""
    import ctypes as C
    import threading

    from tests.util_seq import gen_pair

    lib = pa.capi.load()
    lib.pa_combine_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.pa_combine_stats.restype = None

    def combined():
        c, b = C.c_uint64(0), C.c_uint64(0)
        lib.pa_combine_stats(C.byref(c), C.byref(b))
        return c.value, b.value

    monkeypatch.setenv("PA_COMBINE_MIN", "2")  # (by default a dozen callers have to be inside at once before anybody is combined)
    T, PER = 16, 12
    pairs = [gen_pair(500 + 331 * (i % 23), (0.01, 0.05, 0.1, 0.2)[i % 4], seed=700 + i) for i in range(T * PER)]
    pairs[5] = (b"ACGT", b"ACGGT")
    keys = ("num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "f_max_tries", "dt_trace_tries", "dt_trace_success",
            "dt_trace_fallback", "fill_tries", "fill_success", "fill_fallback")
    al = {"simple": pa.AstarPa2Params.simple().make_aligner(True), "full": pa.AstarPa2Params.full().make_aligner(True)}

    def one(i):
        a, b = pairs[i]
        if i % 3 == 0:
            return pa.c_abi_align("astarpa2_simple", a, b)
        if i % 3 == 1:
            return pa.c_abi_align("astarpa2_full", a, b)
        c, g, st = al["simple" if i % 2 else "full"].align_with_stats(a, b)
        return c, g, tuple(int(st[k]) for k in keys)

    alone = [one(i) for i in range(len(pairs))]  # one caller at a time: the single-pair route
    before = combined()
    assert before[0] == 0 or True
    got = [None] * len(pairs)
    errors = []
    gate = threading.Barrier(T)

    def work(t):
        try:
            gate.wait()
            for i in range(t, len(pairs), T):
                got[i] = one(i)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    assert got == alone
    after = combined()
    assert after[0] - before[0] >= len(pairs) // 4 and after[1] > before[1], (before, after)  # calls really went out in batches
    for i in range(0, len(pairs), 17):
        assert alone[i][0] == oracle.levenshtein(*pairs[i])


def test_c_pthreads_through_the_drop_in_symbols(tmp_path):
    """tests/c_abi/dropin_threads.c: 24 and 48 pthreads calling astarpa2_simple / astarpa2_full from plain C -- above the call combiner's
    threshold, so the calls go out as batches -- and every (cost, CIGAR) must equal what the same call returned when it was made alone
    (the program exits non-zero otherwise)."""
    import os
    import shutil
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    libdir = root / "astar-pairwise-aligner_amd"
    gcc = shutil.which("gcc") or shutil.which("cc")
    if gcc is None or not (libdir / "libastarpa_c_hip.so").exists():
        pytest.skip("no C compiler or library")
    exe = tmp_path / "dropin_threads"
    subprocess.run([gcc, "-O2", str(root / "tests" / "c_abi" / "dropin_threads.c"), "-I", str(root / "include"), "-L", str(libdir), "-lastarpa_c_hip", "-lpthread",
                    "-Wl,-rpath," + str(libdir), "-o", str(exe)], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=str(libdir) + ":" + os.environ.get("LD_LIBRARY_PATH", ""), GPU_MAX_HW_QUEUES="16")
    for sym in ("simple", "full"):
        out = subprocess.run([str(exe), "192", sym, "24", "48"], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0 and "DIFFER" not in out.stdout, out.stdout + out.stderr
        assert out.stdout.count("pairs/s") == 3


def test_gpu_results_equal_the_second_restatement(pa, oracle):
    """Every GPU route against oracle/astarpa2_restated.py directly (pure Python on big integers, no line shared with csrc/engine.hpp):
    pa_align with the three presets (the sweep kernel for `simple`, the host-driven HIP engine for `full` and `nw`) and the batched A*PA2."""
    from oracle import astarpa2_restated as restated
    from tests.test_restated_engine import KEYS

    kw = {"nw": dict(domain="full", doubling="none", sparse=False, dt_trace=False), "simple": dict(heuristic="gap"),
          "full": dict(heuristic="gcsh", k=12, p=14, prune=True, incremental_doubling=True)}
    mk = {"nw": pa.AstarPa2Params.nw, "simple": pa.AstarPa2Params.simple, "full": pa.AstarPa2Params.full}
    pairs = [gen_pair(n, e, seed=500 + t) for t, (n, e) in enumerate([(700, 0.1), (3000, 0.04), (5000, 0.15), (12_000, 0.08), (20_000, 0.02), (9000, 0.3)])]
    a, b = pairs[3]
    pairs.append((a, b[:4000] + b[4600:]))  # a long deletion
    want = {name: [restated.align(x, y, **kw[name]) for x, y in pairs if name != "nw" or len(x) <= 5000] for name in kw}
    for name in kw:
        sel = [p_ for p_ in pairs if name != "nw" or len(p_[0]) <= 5000]
        al = mk[name]().make_aligner(True)
        for (x, y), w in zip(sel, want[name]):
            cost, cigar, stats = al.align_with_stats(x, y)
            assert (cost, cigar) == w[:2], (name, len(x))
            assert {k: int(stats[k]) for k in KEYS} == {k: w[2][k] for k in KEYS}, (name, len(x))
    bt = pa.Batch(pairs, params=pa.AstarPa2Params.simple())
    costs, cigars, _, _ = bt.align()
    st = bt.pair_stats()
    bt.close()
    for i, w in enumerate(want["simple"]):
        assert (int(costs[i]), cigars[i]) == w[:2] and {k: int(st[i][k]) for k in KEYS} == {k: w[2][k] for k in KEYS}, i


@pytest.mark.parametrize("preset", ["simple", "full"])
def test_reference_harness_one_call_at_a_time(pa, oracle, preset):
    """pa-test's whole `test_aligner` set (tests/test_gpu_apa2_full.py harness_pairs: 8 literal pairs, the full length x error-rate grid,
    structural error models) through pa_align, one pair per call -- what the drop-in symbols astarpa2_simple / astarpa2_full run: cost =
    Levenshtein, CIGAR and statistics = the CPU-kernel engine."""
    from tests.test_gpu_apa2_full import harness_pairs

    oc = oracle.params_full() if preset == "full" else oracle.params_simple()
    aligner = gpu_params(pa, oc).make_aligner(True)
    for a, b in harness_pairs():
        want_cost, want_cigar, want_stats = oracle.cpu_align(a, b, oc)
        cost, cigar, stats = aligner.align_with_stats(a, b)
        assert (cost, cigar) == (want_cost, want_cigar), (len(a), len(b))
        assert {k: stats[k] for k in STAT_KEYS} == {k: want_stats[k] for k in STAT_KEYS}, (len(a), len(b))
        assert cost == oracle.levenshtein(a, b)
