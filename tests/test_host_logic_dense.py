"""Independent checks of the host band logic (engine.hpp: j_range / fixed_j_range / compute_next_block bookkeeping) that do NOT go
through the engine's own accessors: the blocks of the last completed pass are dumped as raw arrays (oracle.cpu_align_blocks) and
compared with a dense Levenshtein matrix computed here with numpy.  Rules checked (astarpa2/src/domain.rs:251-350, blocks.rs:245-340,
block.rs:69-122):
  * a block's column, rebuilt from its V words by popcounts written here, starts at top_val and ends at bot_val;
  * every value of the column is an upper bound of the true distance, and EXACT inside the block's fixed_j_range;
  * top_val = (previous column at the new range start) + block width, with +1 per row beyond the previous range;
  * every row of the searched window with g + h <= f_max lies inside the reported fixed range (the reported range may be wider:
    it is united with the ranges of earlier passes), and both ends of the reported range satisfy g + h <= f_max themselves;
  * ranges are rounded out to 64, the next range starts at the previous fixed start, the final cost is D[n][m].
The device-side sweep and the HIP engine are compared with this engine bit for bit elsewhere (tests/test_sweep_emu.py,
tests/test_gpu_sweep.py, tests/test_gpu_engine.py), so these rules pin them too."""
import numpy as np
import pytest

from tests.test_sweep_emu import variants
from tests.util_seq import gen_pair, rand_seq


def dense_dp(a: bytes, b: bytes) -> np.ndarray:
    """D[i][j] = edit distance of a[:i], b[:j]; rows by the running-minimum trick."""
    n, m = len(a), len(b)
    bb = np.frombuffer(b, np.uint8)
    D = np.zeros((n + 1, m + 1), np.int32)
    D[0] = np.arange(m + 1)
    j = np.arange(m + 1, dtype=np.int32)
    for i in range(1, n + 1):
        prev = D[i - 1]
        t = np.empty(m + 1, np.int32)
        t[0] = i
        t[1:] = np.minimum(prev[1:] + 1, prev[:-1] + (bb != a[i - 1]))
        D[i] = np.minimum.accumulate(t - j) + j
    return D


def column(block):
    """Values at rows js..je of the block's right edge, from its V words (bit k of p / m = +1 / -1 between rows 64w+k and 64w+k+1)."""
    vals = [block["top_val"]]
    for p, m in block["v"]:
        for k in range(64):
            vals.append(vals[-1] + ((p >> k) & 1) - ((m >> k) & 1))
    return np.array(vals, np.int64)


def h_of(name, a, b, oracle):
    n, m = len(a), len(b)
    if name in ("simple", "gap_nosparseh", "gap_nodt", "gap_startgap"):
        return lambda i, j: abs((n - i) - (m - j))
    if name == "dijkstra":
        return lambda i, j: 0
    k = 12 if name == "sh12" else 5
    tab = oracle.sh_h(a, b, k)
    return lambda i, j: tab[i]


@pytest.mark.parametrize("name", ["simple", "gap_nosparseh", "dijkstra", "sh5", "gap_startgap"])
def test_blocks_against_dense_dp(oracle, name):
    prm = variants(oracle)[name]
    cases = [gen_pair(n, e, s) for n, e, s in [(300, 0.1, 1), (700, 0.3, 2), (1500, 0.05, 3), (1100, 0.6, 4), (2000, 0.15, 5)]]
    a0, b0 = gen_pair(1200, 0.1, 6)
    cases.append((a0, b0[:500] + rand_seq(300, 9) + b0[500:]))  # a long insertion
    cases.append((a0, b0[:400] + b0[650:]))                      # a long deletion
    for a, b in cases:
        n, m = len(a), len(b)
        D = dense_dp(a, b)
        h = h_of(name, a, b, oracle)
        cost, f_max, blocks = oracle.cpu_align_blocks(a, b, prm)
        assert cost == int(D[n][m])
        assert blocks[0]["i0"] == -1 and blocks[0]["i1"] == 0 and blocks[-1]["i1"] == n
        prev_col, prev = None, None
        for blk in blocks:
            ie, js, je = blk["i1"], blk["js"], blk["je"]
            assert js % 64 == 0 and je % 64 == 0 and js <= blk["ojs"] and blk["oje"] <= je and len(blk["v"]) == (je - js) // 64
            assert js == blk["ojs"] // 64 * 64 and je == -(-blk["oje"] // 64) * 64
            col = column(blk)
            assert col[-1] == blk["bot_val"]
            rows = np.arange(js, min(je, m) + 1)
            true = D[ie][rows]
            got = col[: len(rows)]
            assert (got >= true).all()
            fs, fe = blk["fs"], blk["fe"]
            assert js <= fs <= fe <= min(blk["oje"], m) or prev is None
            assert (got[fs - js: fe - js + 1] == true[fs - js: fe - js + 1]).all(), (ie, fs, fe)
            f = got + np.array([h(ie, int(j)) for j in rows])
            assert f[fs - js] <= f_max and f[fe - js] <= f_max
            if prev is not None:
                width = ie - prev["i1"]
                # top_val: the previous column at the new start (+1 per row beyond the previous range), plus the block's width
                pj = js - prev["js"]
                assert pj >= 0
                pv = prev_col[pj] if js <= prev["je"] else prev_col[-1] + (js - prev["je"])
                assert blk["top_val"] == pv + width
                assert blk["ojs"] <= prev["fs"]  # the new range starts at the previous fixed start (or higher up, from an older pass)
                # every row of the searched window that satisfies g + h <= f_max is inside the reported fixed range
                lo, hi = prev["fs"], min(blk["oje"], m)
                ok = np.nonzero(f[lo - js: hi - js + 1] <= f_max)[0] + lo
                if len(ok):
                    assert fs <= ok[0] and ok[-1] <= fe
            prev_col, prev = col, blk
