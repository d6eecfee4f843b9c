"""Pin the CPU oracle against the reference's own known answers."""
import numpy as np
import pytest

from tests.util_seq import PA_TEST_PAIRS, gen_pair, rand_seq


def test_search_doctest(oracle):
    # pa-bitpacking/src/search.rs:30-31
    assert oracle.search(b"AC", b"CTTACTTA", 0.0) == [0, 0, 1, 2, 1, 0, 1, 2, 1, 0, 0]


def test_search_python_readme(oracle):
    # pa_python/readme.md:13-16
    assert oracle.search(b"CT", b"ACTG", 1.0) == [2, 2, 1, 0, 1, 2, 2]


def test_search_wildcards(oracle):
    # profile.rs:39-50: N/* match anything, Y = C|T, R = A|G (pattern side only)
    assert oracle.search(b"NN", b"ACGT", 0.0)[4] == 0
    assert oracle.search(b"Y", b"C", 1.0)[1] == 0
    assert oracle.search(b"R", b"C", 1.0)[1] == 1


def test_c_example_pair(oracle):
    # astarpa-c/example.c:8-29 asserts cost 2 for this pair
    a, b = b"ACTCGCT", b"AACTCGTT"
    assert oracle.levenshtein(a, b) == 2
    assert oracle.nw_cost(a, b, False) == 2
    assert oracle.nw_cost(a, b, True) == 2
    # astarpa-c/example.cpp:16: "=I4=X=" is a valid cost-2 CIGAR for it
    assert oracle.cigar_verify("=I4=X=", a, b) == 2


def test_cigar_verify_rejects(oracle):
    a, b = b"ACTCGCT", b"AACTCGTT"
    assert oracle.cigar_verify("=I4=X", a, b) == -1      # does not reach the end
    assert oracle.cigar_verify("2=I3=X=", a, b) == -1    # '=' on a mismatch
    assert oracle.cigar_verify("=I4=2=", a, b) == -1
    assert oracle.cigar_verify("", b"", b"") == 0
    assert oracle.cigar_verify("3I", b"", b"ACG") == 3
    assert oracle.cigar_verify("3D", b"ACG", b"") == 3


@pytest.mark.parametrize("rows", [64, 128, 192, 256, 320, 384, 448, 512])
def test_bench_rule_all_schedules(oracle, rows):
    """pa-bitpacking/benches/nw/main.rs:139-160: for h=+1, v=+1 every schedule returns lev(a,b)-|b|."""
    a = rand_seq(256, seed=31415)
    b = rand_seq(rows, seed=31415 + rows)
    want = oracle.levenshtein(a, b) - len(b)
    pa, pb = oracle.bitprofile_build(a, b)
    outs = []
    for fn in ("row", "col", "simd_exact", "simd_pad", "avx_exact", "avx_pad", "fill"):
        h, v = oracle.ones_h(len(pa)), oracle.ones_v(len(pb))
        if fn == "row":
            r = oracle.scalar_row(pa, pb, h, v)
        elif fn == "col":
            r = oracle.scalar_col(pa, pb, h, v)
        elif fn == "simd_exact":
            r = oracle.simd_compute(pa, pb, h, v, True)
        elif fn == "simd_pad":
            r = oracle.simd_compute(pa, pb, h, v, False)
        elif fn == "avx_exact":
            r = oracle.strip_compute_avx2(pa, pb, h, v, True)
        elif fn == "avx_pad":
            r = oracle.strip_compute_avx2(pa, pb, h, v, False)
        else:
            r, values = oracle.scalar_fill(pa, pb, h, v)
            assert np.array_equal(values[-1], v)
        assert r == want, fn
        outs.append((fn, h.copy(), v.copy()))
    for fn, h, v in outs[1:]:
        assert np.array_equal(v, outs[0][2]), fn
        if fn not in ("simd_pad", "avx_pad") or oracle.simd_pad_rows(256, len(pb), False) == 0:
            assert np.array_equal(h, outs[0][1]), fn


def test_strip_schedule_matches_restatement_ragged(oracle):
    """AVX2 strip port == schedule-independent restatement on ragged shapes, both tail modes,
    including h out in the padded (non-exact) case."""
    rng = np.random.default_rng(7)
    for _ in range(300):
        n = int(rng.integers(1, 300))
        m = int(rng.integers(1, 64 * 21))
        a = rand_seq(n, seed=int(rng.integers(1 << 30)))
        b = rand_seq(m, seed=int(rng.integers(1 << 30)))
        pa, pb = oracle.bitprofile_build(a, b)
        for exact in (True, False):
            h0 = np.zeros(n, oracle.H_DTYPE)
            sel = rng.integers(0, 3, n)
            h0["p"] = sel == 0
            h0["m"] = sel == 1
            # random but valid vertical deltas
            v0 = np.zeros(len(pb), oracle.V_DTYPE)
            bits = rng.integers(0, 1 << 63, len(pb), dtype=np.uint64) * 2 + rng.integers(0, 2, len(pb), dtype=np.uint64)
            other = rng.integers(0, 1 << 63, len(pb), dtype=np.uint64) * 2
            v0["p"] = bits
            v0["m"] = other & ~bits
            h1, v1, h2, v2 = h0.copy(), v0.copy(), h0.copy(), v0.copy()
            r1 = oracle.simd_compute(pa, pb, h1, v1, exact)
            r2 = oracle.strip_compute_avx2(pa, pb, h2, v2, exact)
            assert r1 == r2
            assert np.array_equal(v1, v2)
            assert np.array_equal(h1, h2)
            if exact:
                h3, v3 = h0.copy(), v0.copy()
                r3 = oracle.scalar_col(pa, pb, h3, v3)
                assert (r3, h3.tobytes(), v3.tobytes()) == (r1, h1.tobytes(), v1.tobytes())


def test_nonexact_return_is_exact(oracle):
    """simd.rs:184-225: with padding the returned sum and v are exact, only h differs."""
    rng = np.random.default_rng(3)
    for w in range(1, 20):
        n = 256
        a = rand_seq(n, seed=w)
        b = rand_seq(64 * w, seed=100 + w)
        pa, pb = oracle.bitprofile_build(a, b)
        h1, v1, h2, v2 = oracle.ones_h(n), oracle.ones_v(w), oracle.ones_h(n), oracle.ones_v(w)
        assert oracle.simd_compute(pa, pb, h1, v1, True) == oracle.simd_compute(pa, pb, h2, v2, False)
        assert np.array_equal(v1, v2)


def test_nw_cost_pa_test_pairs(oracle):
    # pa-test/src/lib.rs:7-20 literal pairs; rule: cost == Levenshtein
    for a, b in PA_TEST_PAIRS:
        want = oracle.levenshtein(a, b)
        assert oracle.nw_cost(a, b, False) == want
        assert oracle.nw_cost(a, b, True) == want


@pytest.mark.parametrize("n", [0, 1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 1000])
def test_nw_cost_grid(oracle, n):
    for e in (0.0, 0.01, 0.05, 0.2, 0.5, 1.0):
        a, b = gen_pair(n, e, seed=n * 131 + int(e * 100))
        want = oracle.levenshtein(a, b)
        assert oracle.nw_cost(a, b, False) == want
        assert oracle.nw_cost(a, b, True) == want


def test_bitprofile_rejects_non_acgt(oracle):
    # profile.rs:113-126: RankTransform over "ACGT" panics on anything else (lowercase, N)
    with pytest.raises(ValueError):
        oracle.bitprofile_build(b"ACGN", b"ACGT")
    with pytest.raises(ValueError):
        oracle.bitprofile_build(b"ACGT", b"acgt")


def test_bitprofile_encoding(oracle):
    # profile.rs:97-110: a: (-(r&1), -((r>>1)&1)); b: negated bits packed; pad rows (0,0)
    pa, pb = oracle.bitprofile_build(b"ACGT", b"ACGTA")
    full = 0xFFFFFFFFFFFFFFFF
    assert pa["b0"].tolist() == [0, full, 0, full]
    assert pa["b1"].tolist() == [0, 0, full, full]
    assert pb["b0"].tolist() == [0b10101]
    assert pb["b1"].tolist() == [0b10011]


def test_search_trace_doc_example(oracle):
    """The documented matrix of search.rs:28-45 pins the hit "AC" at text[3..5): tracing output index 5 (cost 0) walks
    (3,0) -> (4,1) -> (5,2) with two matches; index 4 (cost 1) needs one insertion.  The reference holds no vector for
    SearchResult::trace itself (pa_python exposes `.out` only): everything else about it is parity unpinned."""
    assert oracle.search_trace(b"AC", b"CTTACTTA", 0.0, 5) == ("2=", [(3, 0), (4, 1), (5, 2)])
    out = oracle.search(b"AC", b"CTTACTTA", 0.0)
    for idx in range(9):  # bottom row: the CIGAR's edit count is the reported cost
        cigar, path = oracle.search_trace(b"AC", b"CTTACTTA", 0.0, idx)
        import re

        edits = sum(int(n or 1) for n, op in re.findall(r"(\d*)([=XID])", cigar) if op != "=")
        assert edits == out[idx]
        assert path[-1] == (idx, 2)
