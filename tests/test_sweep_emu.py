"""The device-side A*PA2 sweep (csrc/sweep_wave.hpp: one persistent launch per align_for_bounded_dist pass, band decisions in the
kernel) run WITHOUT a GPU: the same wave program over an array-emulated wavefront, one host thread per wavefront
(oracle/sweep_emu.cpp).  It must reproduce the host-driven engine over the oracle kernels exactly: cost, CIGAR string and every
band statistic (the band logic is the reference's, astarpa2/src/domain.rs:117-350; the schedule is not)."""
import random

import pytest

from tests.util_seq import gen_pair, rand_seq

KEYS = ["num_blocks", "num_incremental_blocks", "computed_lanes", "unique_lanes", "f_max_tries", "sanity_violations", "dt_trace_tries",
        "dt_trace_success", "dt_trace_fallback", "fill_tries", "fill_success", "fill_fallback"]


def variants(o):
    base = dict(domain="astar", doubling="band", start="h0", factor=2.0, block_width=256, sparse=True, incremental_doubling=False,
                dt_trace=True, max_g=40, fr_drop=10, sparse_h=True)
    return {
        "simple": o.params_simple(),
        "dijkstra": o.make_params(**{**base, "heuristic": "none"}),
        "sh12": o.make_params(**{**base, "heuristic": "sh", "k": 12}),
        "sh5": o.make_params(**{**base, "heuristic": "sh", "k": 5}),
        "gap_nosparseh": o.make_params(**{**base, "heuristic": "gap", "sparse_h": False}),
        "gap_nodt": o.make_params(**{**base, "heuristic": "gap", "dt_trace": False}),
        "gap_startgap": o.make_params(**{**base, "heuristic": "gap", "start": "gap"}),
        "gap_startzero_f15": o.make_params(**{**base, "heuristic": "gap", "start": "zero", "factor": 1.5}),
        "linear": o.make_params(**{**base, "heuristic": "gap", "doubling": "linear", "start": "h0", "delta": 300.0}),
    }


def both(o, a, b, prm, trace=True, nwaves=16):
    want = o.cpu_align(a, b, prm, trace=trace)
    rc, cost, cigar, stats, info = o.sweep_emu_align(a, b, prm, trace=trace, nwaves=nwaves)
    assert rc == 0, info
    if trace:
        assert cost == want[0]
        assert cigar == want[1]
        assert {k: stats[k] for k in KEYS} == {k: want[2][k] for k in KEYS}
    else:
        # The reference's cost-only path keeps ONE block whose fixed range only grows (blocks.rs:245-270) and can end on an upper
        # bound (seen with SH: 11353 for a distance of 11325); the sweep always runs the traced band and returns the distance.
        assert cigar is None or cigar == ""
        assert cost == o.nw_cost(a, b, True) <= want[0]
    return cost


def test_block_boundary_sizes(oracle):
    prm = oracle.params_simple()
    for n in (1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 512, 513, 2047, 2048, 2049, 4095, 4096, 4097, 8191):
        for e in (0.0, 0.05, 0.4, 1.0):
            a, b = gen_pair(n, e, seed=n * 7 + int(e * 100))
            assert both(oracle, a, b, prm) == oracle.levenshtein(a, b)


def test_multi_strip_bands(oracle):
    """Bands of several 2048-row strips, several passes, both edges moving across strip boundaries."""
    for name in ("simple", "dijkstra", "sh12"):
        prm = variants(oracle)[name]
        for n, e, seed in [(10000, 0.15, 4), (20000, 0.3, 5), (30000, 0.2, 6), (8192, 0.4, 9)]:
            a, b = gen_pair(n, e, seed)
            both(oracle, a, b, prm)


def test_unsupported_parameters_are_reported(oracle):
    a, b = gen_pair(500, 0.1, 1)
    assert oracle.sweep_emu_align(a, b, oracle.params_full())[0] == 1
    assert oracle.sweep_emu_align(a, b, oracle.params_nw())[0] == 1


@pytest.mark.parametrize("seed", [1, 2])
def test_random_pairs_all_variants(oracle, seed):
    rng = random.Random(seed)
    vs = variants(oracle)
    for _ in range(60):
        name = rng.choice(list(vs))
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
        both(oracle, a, b, vs[name], trace=rng.random() < 0.8)


def test_jr_end_fast_forward_equals_literal_loops(oracle):
    """sweep_logic.hpp jr_end_astar (galloping over the reference's +8 / +1 probing runs) against the literal loops of
    domain.rs:171-233 as restated here."""
    import ctypes as C

    import numpy as np

    L = oracle.sweep_emu_lib()

    def literal(kind, n, m, sh, is_, ie, fe, gu, f_max, sparse_h):
        def h(x, y):
            return abs((n - x) - (m - y)) if kind == 1 else (int(sh[x]) if kind == 2 else 0)

        u0, u1 = is_, fe
        f = lambda x, y: gu + abs((x - u0) - (y - u1)) + h(x, y)
        v0, v1 = u0, u1
        if not sparse_h:
            while v0 < ie:
                v0 += 1
                v1 += 2
                while v1 <= m and f(v0, v1) <= f_max:
                    v1 += 1
                v1 -= 1
            return v1
        v0 += 1
        v1 = min(v1 + 1 + 256, m)
        while True:
            if v1 < v0 - u0 + u1:
                v1 = v0 - u0 + u1
                break
            fv = f(v0, v1)
            if fv <= f_max:
                if v1 == m:
                    break
                v1 = min(v1 + 8, m)
            else:
                v0 += -((f_max - fv) // 2)
                if v0 > ie:
                    v0 = ie
                    break
        v0 = ie
        while True:
            if v1 < v0 - u0 + u1:
                v1 = v0 - u0 + u1
                break
            fv = f(v0, v1)
            if fv <= f_max:
                break
            v1 -= -((f_max - fv) // 2)
        return v1

    rng = random.Random(5)
    for _ in range(3000):
        n, m = rng.randint(1, 5000), rng.randint(1, 5000)
        kind = rng.choice([0, 1, 2])
        sh = np.sort(np.array([rng.randint(0, 40) for _ in range(n + 1)], np.int32))[::-1].copy()
        is_ = rng.choice([-1, rng.randint(0, max(0, n - 1))])
        ie = 0 if is_ < 0 else min(n, is_ + rng.choice([1, 17, 256]))
        fe = -1 if is_ < 0 else rng.randint(0, m)
        gu = 0 if is_ < 0 else rng.randint(0, 300)
        f_max = rng.randint(0, 3000)
        sp = rng.choice([0, 1])
        got = L.pa_sweep_jr_end(kind, n, m, sh.ctypes.data_as(C.c_void_p), is_, ie, fe, gu, f_max, sp)
        assert got == literal(kind, n, m, sh, is_, ie, fe, gu, f_max, sp), (kind, n, m, is_, ie, fe, gu, f_max, sp)


def test_pipelined_passes_equal_sequential_passes(oracle, monkeypatch):
    """The passes of one band search run pipelined (sweep_host.hpp search()): the pass for the next bound is launched while the
    current one runs and reads its block records as they appear.  Whatever the depth, cost / CIGAR / statistics are those of
    one pass after the other -- which in turn are the host-driven engine's."""
    import random

    rng = random.Random(5)
    vs = variants(oracle)
    cases = []
    for _ in range(40):
        n = rng.choice([rng.randint(300, 3000), rng.randint(3000, 15000)])
        cases.append((rng.choice(list(vs)), gen_pair(n, rng.choice([0.02, 0.1, 0.25, 0.5]), rng.randint(1, 10**6))))
    results = {}
    for depth in ("1", "2", "3", "6"):
        monkeypatch.setenv("PA_SWEEP_EMU_DEPTH", depth)
        results[depth] = [oracle.sweep_emu_align(a, b, vs[name], trace=True, nwaves=16)[:4] for name, (a, b) in cases]
    for (name, (a, b)), r1, r2, r3, r6 in zip(cases, results["1"], results["2"], results["3"], results["6"]):
        want = oracle.cpu_align(a, b, vs[name], trace=True)
        for rc, cost, cigar, stats in (r1, r2, r3, r6):
            assert rc == 0 and (cost, cigar) == (want[0], want[1])
            assert {k: stats[k] for k in KEYS} == {k: want[2][k] for k in KEYS}


def test_giving_up_speculative_passes_changes_nothing(oracle, monkeypatch):
    """The passes launched ahead rest on two assumptions about the pass before them (sweep_host.hpp search()); when one fails they
    are cancelled and the search goes on from the real state.  That almost never happens on its own, so PA_SWEEP_TEST_GIVE_UP makes
    it happen after every k-th pass: cancel words, slots taken again by later passes, merged records of given-up passes -- the
    results must not move."""
    import random

    rng = random.Random(21)
    vs = variants(oracle)
    for k in ("1", "2", "3", "5"):
        monkeypatch.setenv("PA_SWEEP_EMU_DEPTH", "6" if k == "5" else "3")
        monkeypatch.setenv("PA_SWEEP_TEST_GIVE_UP", k)
        for _ in range(20):
            name = rng.choice(list(vs))
            n = rng.choice([rng.randint(500, 4000), rng.randint(4000, 20000)])
            a, b = gen_pair(n, rng.choice([0.05, 0.15, 0.3, 0.5]), rng.randint(1, 10**6))
            both(oracle, a, b, vs[name], trace=rng.random() < 0.8)
