"""GPU parity of the HIP operators (through the C ABI) against the CPU oracle -- bit exact."""
import numpy as np
import pytest

from tests.util_seq import PA_TEST_PAIRS, gen_pair, rand_seq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import astar_pairwise_aligner_amd as pa

    pa.require_gpu()
    return pa


def _as_u64(arr):
    return np.ascontiguousarray(arr).view(np.uint64).reshape(len(arr), 2)


def _rand_hv(rng, n, w):
    h = np.zeros((n, 2), np.uint64)
    sel = rng.integers(0, 3, n)
    h[:, 0] = sel == 0
    h[:, 1] = sel == 1
    p = rng.integers(0, 1 << 63, w, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, w, dtype=np.uint64)
    m = (rng.integers(0, 1 << 63, w, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, w, dtype=np.uint64)) & ~p
    v = np.stack([p, m], axis=1).astype(np.uint64)
    return h, v


def test_profile_build_matches_oracle(pa, oracle):
    for n, m in [(1, 1), (7, 8), (64, 64), (100, 129), (1000, 2049), (5000, 4097)]:
        a, b = rand_seq(n, seed=n), rand_seq(m, seed=m + 1)
        a2, b2 = pa.profile_build(a, b)
        oa, ob = oracle.bitprofile_build(a, b)
        assert np.array_equal(a2, _as_u64(oa))
        assert np.array_equal(b2, _as_u64(ob))
    with pytest.raises(ValueError):
        pa.profile_build(b"ACGN", b"ACGT")
    with pytest.raises(ValueError):
        pa.profile_build(b"ACGT", b"ACgT")


# shapes: tiny, ragged tails (1..7 words past 8), one strip exactly, strip boundaries, several strips
SHAPES = [(1, 1), (3, 1), (15, 2), (16, 3), (17, 5), (31, 7), (64, 8), (100, 9), (256, 12), (256, 31), (256, 32),
          (256, 33), (250, 63), (256, 64), (256, 65), (300, 100), (512, 129), (1000, 40), (77, 97),
          # around the half-wave threshold (strips of <= 16 words run in lanes 32..63 with a 32-step skew) and its chunk edges
          (1, 16), (31, 16), (32, 16), (33, 16), (64, 15), (255, 15), (256, 16), (257, 17), (288, 16), (1000, 16)]


@pytest.mark.parametrize("n,w", SHAPES)
@pytest.mark.parametrize("exact", [True, False])
def test_compute_matches_oracle(pa, oracle, n, w, exact):
    rng = np.random.default_rng(n * 1000 + w)
    for trial in range(2):
        m = 64 * w - int(rng.integers(0, 64)) if trial else 64 * w
        a, b = rand_seq(n, seed=n + trial), rand_seq(m, seed=w * 7 + trial)
        oa, ob = oracle.bitprofile_build(a, b)
        if trial == 0:
            h, v = np.zeros((n, 2), np.uint64), np.zeros((w, 2), np.uint64)
            h[:, 0] = 1
            v[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        else:
            h, v = _rand_hv(rng, n, w)
        h_or, v_or = h.copy().view(oracle.H_DTYPE).reshape(n), v.copy().view(oracle.V_DTYPE).reshape(w)
        want = oracle.simd_compute(oa, ob, h_or, v_or, True)
        h_gpu, v_gpu = h.copy(), v.copy()
        got = pa.compute(_as_u64(oa), _as_u64(ob), h_gpu, v_gpu, exact)
        assert got == want
        assert np.array_equal(v_gpu, _as_u64(v_or))
        if exact:
            assert np.array_equal(h_gpu, _as_u64(h_or))


@pytest.mark.parametrize("n,w", [(1, 1), (5, 3), (16, 4), (100, 9), (256, 10), (256, 33), (64, 70), (33, 16), (256, 16), (300, 15), (31, 17)])
def test_fill_matches_oracle(pa, oracle, n, w):
    rng = np.random.default_rng(n * 31 + w)
    a, b = rand_seq(n, seed=n), rand_seq(64 * w - 5, seed=w)
    oa, ob = oracle.bitprofile_build(a, b)
    h, v = _rand_hv(rng, n, w)
    h_or, v_or = h.copy().view(oracle.H_DTYPE).reshape(n), v.copy().view(oracle.V_DTYPE).reshape(w)
    want, values_or = oracle.scalar_fill(oa, ob, h_or, v_or)
    h_gpu, v_gpu = h.copy(), v.copy()
    got, values = pa.fill(_as_u64(oa), _as_u64(ob), h_gpu, v_gpu)
    assert got == want
    assert np.array_equal(v_gpu, _as_u64(v_or))
    assert np.array_equal(h_gpu, _as_u64(h_or))
    assert np.array_equal(values.reshape(n * w, 2), values_or.reshape(n * w).view(np.uint64).reshape(n * w, 2))


def test_bench_rule_on_gpu(pa, oracle):
    """benches/nw/main.rs:145-149: h=+1, v=+1 => returned bottom sum == lev(a,b) - |b| (256 x 64..512 rows)."""
    a = rand_seq(256, seed=31415)
    for rows in (64, 128, 192, 256, 320, 384, 448, 512):
        b = rand_seq(rows, seed=31415 + rows)
        a2, b2 = pa.profile_build(a, b)
        h, v = np.zeros((256, 2), np.uint64), np.zeros((rows // 64, 2), np.uint64)
        h[:, 0] = 1
        v[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        assert pa.compute(a2, b2, h, v, False) == oracle.levenshtein(a, b) - len(b)


def test_batch_costs_small(pa, oracle):
    pairs = list(PA_TEST_PAIRS)
    for n in (0, 1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 2047, 2048, 2049, 4100):
        for e in (0.0, 0.05, 0.3, 1.0):
            pairs.append(gen_pair(n, e, seed=n * 17 + int(100 * e)))
    pairs.append((b"", b""))
    pairs.append((b"ACGT", b""))
    pairs.append((b"", b"ACGTA"))
    batch = pa.Batch(pairs)
    costs, ms = batch.run()
    for (a, b), c in zip(pairs, costs):
        assert c == oracle.levenshtein(a, b), (len(a), len(b))
    # idempotent: a second pass over the same resident inputs gives the same answers
    costs2, _ = batch.run()
    assert np.array_equal(costs, costs2)
    batch.close()


def test_c_example_pair_batch(pa):
    # astarpa-c/example.c:8-29
    batch = pa.Batch([(b"ACTCGCT", b"AACTCGTT")])
    assert batch.run()[0].tolist() == [2]


def test_batch_multi_strip_chain(pa, oracle):
    """Many strips chained through granules (20 kbp => 10 strips), mixed with short pairs."""
    pairs = [gen_pair(20000, 0.05, seed=5), gen_pair(300, 0.1, seed=6), gen_pair(12345, 0.15, seed=7)]
    costs, _ = pa.Batch(pairs).run()
    for (a, b), c in zip(pairs, costs):
        assert c == oracle.nw_cost(a, b, True)


@pytest.mark.parametrize("k", [1, 2, 4, 8])
@pytest.mark.parametrize("mode", ["chain", "seq"])
def test_batch_shapes(pa, oracle, monkeypatch, k, mode):
    """Every strip height (32*k rows per lane) and both schedules (chained strips / one wavefront per pair) give the
    same costs: ragged lengths around the lane, word and strip boundaries, empty sequences, multi-strip pairs."""
    monkeypatch.setenv("PA_STRIP_K", str(k))
    monkeypatch.setenv("PA_BATCH_MODE", mode)
    pairs = list(PA_TEST_PAIRS)
    rows_per_strip = 2048 * k
    for n in (1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 257, rows_per_strip - 1, rows_per_strip, rows_per_strip + 1,
              rows_per_strip + 65, rows_per_strip + 2048 + 70, rows_per_strip + 3 * 2048 + 5, 3 * rows_per_strip + 100):
        pairs.append(gen_pair(n, 0.1, seed=n * 7 + k))
    pairs.append((rand_seq(700, seed=1), rand_seq(2 * rows_per_strip + 130, seed=2)))   # tall and narrow
    pairs.append((rand_seq(2 * rows_per_strip + 130, seed=3), rand_seq(70, seed=4)))    # short and wide
    pairs += [(b"", b""), (b"ACGT", b""), (b"", b"ACGTA")]
    batch = pa.Batch(pairs)
    costs, _ = batch.run()
    for (a, b), c in zip(pairs, costs):
        want = oracle.levenshtein(a, b) if len(a) * len(b) < 4_000_000 else oracle.nw_cost(a, b, True)
        assert c == want, (len(a), len(b))
    costs2, _ = batch.run()
    assert np.array_equal(costs, costs2)
    batch.close()


@pytest.mark.parametrize("k", [1, 2, 4, 8])
@pytest.mark.parametrize("mode", ["chain", "seq"])
def test_batch_shapes_low_complexity_first_column(pa, oracle, monkeypatch, k, mode):
    """The first character of `a` does not occur in the first rows of `b` (several lanes deep): every lane must see column 0's
    own eq word.  (Random sequences hide a wrong eq there: a[0] almost surely matches somewhere in lane 0's rows, after which
    the horizontal delta is -1 all the way down whatever eq says.)  With eq words prefetched from LDS (k = 4, 8) lanes >= 1
    once took the eq word of 'A' for column 0."""
    monkeypatch.setenv("PA_STRIP_K", str(k))
    monkeypatch.setenv("PA_BATCH_MODE", mode)
    pairs = []
    for first, fill in ((b"C", b"A"), (b"G", b"T"), (b"T", b"C"), (b"A", b"G")):
        for depth in (300, 700, 2048 * k + 300):
            y = rand_seq(900, seed=depth + k)
            pairs.append((first + y, fill * depth + y))
            pairs.append((first * 3 + y, fill * depth + first + y))
    batch = pa.Batch(pairs)
    costs, _ = batch.run()
    for (a, b), c in zip(pairs, costs):
        want = oracle.levenshtein(a, b) if len(a) * len(b) < 4_000_000 else oracle.nw_cost(a, b, True)
        assert c == want, (a[:4], b[:4], len(a), len(b))
    batch.close()
    tb = pa.Batch(pairs, trace=True)  # the checkpointing variants of the same kernels
    costs2, cigars, _, _ = tb.align()
    assert np.array_equal(costs, costs2)
    for (a, b), c, cg in zip(pairs, costs2, cigars):
        assert oracle.cigar_verify(cg, a, b) == c
    tb.close()


def test_batch_shape_k16_experiment(pa, oracle, monkeypatch):
    """PA_STRIP_K=16 (one wavefront per pair, 16 subwords = 512 rows per lane, eq words from LDS; an experiment recorded in
    profiles/README.md): same costs on ragged sizes around its lane / strip boundaries and on low-complexity first columns."""
    monkeypatch.setenv("PA_STRIP_K", "16")
    monkeypatch.setenv("PA_BATCH_MODE", "seq")
    rows = 2048 * 16
    pairs = list(PA_TEST_PAIRS)
    for n in (1, 33, 511, 512, 513, 2049, rows - 1, rows, rows + 1, rows + 2048 + 70, 2 * rows + 100):
        pairs.append(gen_pair(n, 0.1, seed=n * 7 + 16))
    pairs.append((rand_seq(700, seed=1), rand_seq(rows + 130, seed=2)))
    for first, fill in ((b"C", b"A"), (b"G", b"T")):
        y = rand_seq(900, seed=5)
        pairs.append((first + y, fill * 1500 + y))
    pairs += [(b"", b""), (b"ACGT", b""), (b"", b"ACGTA")]
    batch = pa.Batch(pairs)
    assert batch.shape()["k"] == 16
    costs, _ = batch.run()
    for (a, b), c in zip(pairs, costs):
        want = oracle.levenshtein(a, b) if len(a) * len(b) < 4_000_000 else oracle.nw_cost(a, b, True)
        assert c == want, (len(a), len(b))
    batch.close()


@pytest.mark.parametrize("mode", ["chain", "seq"])
@pytest.mark.parametrize("k", [0, 1, 2, 4, 8])
@pytest.mark.parametrize("hint", [0.0, 0.02, 0.3])
def test_banded_batch_is_exact_for_any_hint(pa, oracle, monkeypatch, k, hint, mode):
    """Diagonal-band DP (pa_batch_create_banded): a hint that is too small only costs re-runs with a wider band, a generous
    one only costs work; the costs are the full-DP costs either way.  Unequal lengths, long indels, unrelated pairs."""
    if k:
        monkeypatch.setenv("PA_STRIP_K", str(k))
    monkeypatch.setenv("PA_BATCH_MODE", mode)
    rows = 2048 * max(k, 1)
    a = rand_seq(5000, seed=31)
    pairs = [gen_pair(n, e, seed=n + int(1000 * e)) for n in (1, 40, 700, 3000, rows + 500, 3 * rows + 77) for e in (0.0, 0.03, 0.12)]
    pairs += [(a, a[:2000] + a[2600:]), (a[:2000] + a[2600:], a), (a, a[:100] + rand_seq(900, seed=5) + a[100:]),
              (rand_seq(900, seed=1), rand_seq(2500, seed=2)), (rand_seq(2500, seed=3), rand_seq(60, seed=4)),
              (b"", b""), (b"ACGT", b""), (b"", b"ACGTA")]
    batch = pa.Batch(pairs, band=hint)
    costs, _ = batch.run()
    for (x, y), c in zip(pairs, costs):
        want = oracle.levenshtein(x, y) if len(x) * len(y) < 4_000_000 else oracle.nw_cost(x, y, True)
        assert c == want, (len(x), len(y), hint, k)
    costs2, _ = batch.run()  # the widened bands are kept: same answers, no further re-runs needed
    assert np.array_equal(costs, costs2)
    batch.close()


def test_batch_invalid_base(pa):
    with pytest.raises(ValueError):
        pa.Batch([(b"ACGTN", b"ACGT")]).run()


def _ones(n, w):
    h, v = np.zeros((n, 2), np.uint64), np.zeros((w, 2), np.uint64)
    h[:, 0] = 1
    v[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return h, v


@pytest.mark.parametrize("n,m", [(300, 200), (1000, 2049), (5000, 4097), (20000, 9000)])
def test_operator_context_block_loop(pa, oracle, n, m):
    """Device-resident operator handle (pa_bp_ctx_*), used the way the reference's engine uses the operators (blocks.rs:719-724):
    256-column blocks, v carried on the host, the horizontal deltas kept on the device.  (1) whole-height blocks, h_mode None;
    (2) every block as two row ranges, the upper one storing its bottom row (Output), the lower one reading it (Input) -- the
    incremental-doubling pattern of blocks.rs:370-469; (3) Update over a stored row.  All against one oracle call on the full
    rectangle."""
    a, b = rand_seq(n, seed=n), rand_seq(m, seed=m + 1)
    w = (m + 63) // 64
    oa, ob = oracle.bitprofile_build(a, b)
    h, v = _ones(n, w)
    h_or, v_or = h.copy().view(oracle.H_DTYPE).reshape(n), v.copy().view(oracle.V_DTYPE).reshape(w)
    want = oracle.simd_compute(oa, ob, h_or, v_or, True)
    ctx = pa.OperatorContext(a, b)
    # (1)
    _, v1 = _ones(n, w)
    total = 0
    for i0 in range(0, n, 256):
        total += ctx.compute(i0, min(n, i0 + 256), 0, w, v1, ctx.H_NONE)
    assert total == want and np.array_equal(v1, _as_u64(v_or))
    # (2)
    if w >= 2:
        cut = w // 2
        _, v2 = _ones(n, w)
        total = 0
        for i0 in range(0, n, 256):
            i1 = min(n, i0 + 256)
            top, bot = v2[:cut].copy(), v2[cut:].copy()
            ctx.compute(i0, i1, 0, cut, top, ctx.H_OUTPUT)
            total += ctx.compute(i0, i1, cut, w, bot, ctx.H_INPUT)
            v2[:cut], v2[cut:] = top, bot
        assert total == want and np.array_equal(v2, _as_u64(v_or))
    # (3) Update: the stored row (from (2): the bottom row of the upper half) in, this range's bottom row out, then Input below
    if w >= 3:
        c1, c2 = w // 3, 2 * w // 3
        _, v3 = _ones(n, w)
        total = 0
        for i0 in range(0, n, 256):
            i1 = min(n, i0 + 256)
            r1, r2, r3 = v3[:c1].copy(), v3[c1:c2].copy(), v3[c2:].copy()
            ctx.compute(i0, i1, 0, c1, r1, ctx.H_OUTPUT)
            ctx.compute(i0, i1, c1, c2, r2, ctx.H_UPDATE)
            total += ctx.compute(i0, i1, c2, w, r3, ctx.H_INPUT)
            v3[:c1], v3[c1:c2], v3[c2:] = r1, r2, r3
        assert total == want and np.array_equal(v3, _as_u64(v_or))
    ctx.close()


def test_operator_context_fill(pa, oracle):
    n, m = 700, 1000
    a, b = rand_seq(n, seed=3), rand_seq(m, seed=4)
    w = (m + 63) // 64
    oa, ob = oracle.bitprofile_build(a, b)
    ctx = pa.OperatorContext(a, b)
    i0, i1 = 256, 512
    h, v = _ones(i1 - i0, w)
    h_or, v_or = h.copy().view(oracle.H_DTYPE).reshape(i1 - i0), v.copy().view(oracle.V_DTYPE).reshape(w)
    want, values_or = oracle.scalar_fill(oa[i0:i1], ob, h_or, v_or)
    vg = v.copy()
    values, hb = ctx.fill(i0, i1, 0, w, vg)
    assert np.array_equal(vg, _as_u64(v_or))
    assert np.array_equal(values.reshape(-1, 2), values_or.reshape(-1).view(np.uint64).reshape(-1, 2))
    assert int(hb.astype(np.int64).sum()) == want
    with pytest.raises(pa.PaError):
        ctx.compute(0, n + 1, 0, w, vg)
    ctx.close()
