"""oracle/astarpa2_restated.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A second, independent restatement of the reference's A*PA2 path (traced, and with trace = False its cost-only arms), written from the Rust text alone and sharing nothing with
csrc/engine.hpp (the product's host logic, which oracle/engine_cpu.cpp instantiates over the CPU kernels) nor with any kernel of this
repository: the DP of a block runs on Python big integers.  Covered: Domain::{Astar, Full, GapStart, GapGap}; the NoCost / GapCost /
SH / GCSH heuristics (GCSH with exact matches, local pruning and Prune::Start, by its definition: class Gcsh -- so the `full`
preset is covered too); band doubling, linear search and no doubling from every DoublingStart; sparse and dense blocks; incremental doubling (2- and 3-range splits over the stored row of horizontal
differences); DT-trace with its x-drop, the re-fill fallback and the parent rules.

    astarpa2/src/lib.rs:122-175          cost_or_align                      band.rs:11-24, 100-182   initial_values, exponential / linear search
    astarpa2/src/domain.rs:90-246        j_range                            domain.rs:251-350        fixed_j_range
    astarpa2/src/domain.rs:356-541       align_for_bounded_dist             ranges.rs:49-80          JRange rounding / union
    astarpa2/src/block.rs:35-140         Block::default / first_col / index / get / get_diff
    astarpa2/src/blocks.rs:146-199       Blocks::init / pop_last_block / reuse_next_block
    astarpa2/src/blocks.rs:205-469       compute_next_block (dense arm, sparse arm, incremental doubling)
    astarpa2/src/blocks.rs:545-660       last_block / next_block_j_range / set_last_block_fixed_j_range / fill_with_blocks
    astarpa2/src/blocks.rs:662-831       HMode, compute_block, init_v_with_overlap, init_v_with_overlap_preserve_fixed
    astarpa2/src/blocks/trace.rs:16-143  Blocks::trace       trace.rs:145-228  parent       trace.rs:231-418  dt_trace_block
    astarpa2/src/blocks/trace.rs:445-500 extend_left / extend_left_simd (the word-at-a-time loop only changes HOW the run is counted)
    pa-bitpacking/src/myers.rs:27-55     compute_block (one Myers step; here on one integer as tall as the block)
    pa-bitpacking/src/scalar.rs:405-425  fill (every column of a block kept)
    pa-affine-types cost_model.rs:387-399, 453-510  unit costs: max_ins/del_for_cost(s) = s, gap_cost = extend_cost = |d|
    pa-heuristic: GapCost h = |(n - i) - (m - j)|; SH with exact matches (sh.rs:47-106, matches/exact.rs, qgrams.rs:30-43)

Purpose: the CIGAR strings and the statistics of these parameter sets used to be pinned only by the reference's acceptance rules
(cost = distance, CIGAR valid) plus engine.hpp agreeing with itself over two kernel back ends.  Two restatements that were written
separately and agree on every field for thousands of random pairs (tests/test_restated_engine.py; one 1 Mbp pair by hand) leave a
common mis-reading of the Rust text as the only shared failure -- `rule-pinned, twice restated` in DESIGN.md 4.  The first
disagreement it found was real: engine.hpp reported block counters after a linear search, the reference does not (lib.rs:132-140
against 158).  The Rust binary itself cannot be built here (no toolchain), so this is still not a reference-generated pin.
"""
from __future__ import annotations

import math
import struct

W = 64
I_MAX = 2**31 - 1
ONE = ((1 << W) - 1, 0)  # V::one(): every vertical difference +1


def _f32(x: float) -> float:
    return struct.unpack("f", struct.pack("f", x))[0]


def _pc(x: int) -> int:
    return bin(x).count("1")


def _round_out(r):  # ranges.rs:71-73
    return (r[0] // W * W, -(-r[1] // W) * W)


def _union(x, y):
    return (min(x[0], y[0]), max(x[1], y[1]))


def _empty(r):
    return r[0] > r[1]


def kmer_key(s: bytes, i: int, k: int) -> bytes:
    """The key the reference's match table uses for the k-mer s[i..i+k): `q as u32` of the 2 k-bit q-gram whose FIRST character is in
    the high bits (matches/exact.rs:47,53,56; qgrams.rs:36-43) -- the low 32 bits are the LAST min(k, 16) characters.  So for k > 16
    two k-mers that agree on their last 16 characters are one key: they match each other in the reference, and here."""
    return s[i + max(0, k - 16):i + k]


def sh_table(a: bytes, b: bytes, k: int, p: int = 0):
    """h(i) = number of seeds of a starting at >= i that have no exact match anywhere in b (seeds: consecutive k-mers from 0).
    p != 0: HeuristicParams.p is the local-pruning length of EVERY heuristic (pa-heuristic/src/cli.rs:168-180), so SH's matches go through
    MatchBuilder::push's local pruning too (matches.rs:205-247; sh.rs:48: find_matches(.., transform_filter = false)) and a seed only counts
    as matched (seed_cost 0: sh_contours.rs:38-46) if one of its matches is kept."""
    n = len(a)
    nseeds = n // k
    if p != 0:
        g = Gcsh(a, b, k, p, False, transform_filter=False)
        kept_seeds = {int(i) // k for i in g.mi.tolist()}
        matched = [s in kept_seeds for s in range(nseeds)]
    else:
        bk = {kmer_key(b, j, k) for j in range(0, len(b) - k + 1)} if len(b) >= k else set()
        matched = [kmer_key(a, s * k, k) in bk for s in range(nseeds)]
    h = [0] * (n + 1)
    unmatched, nxt = 0, nseeds - 1
    for i in range(n, -1, -1):
        if nxt >= 0 and i == nxt * k:
            if not matched[nxt]:
                unmatched += 1
            nxt -= 1
        h[i] = unmatched
    return h


class Gcsh:
    """GCSH with exact matches (r = 1), local pruning p and Prune::Start, by its DEFINITION (pa-heuristic/src):
        seeds           consecutive k-mers of a from 0; potential P(i) = seeds starting at >= i          seeds.rs:34-71, qgrams.rs:99-109
        matches         every (seed start i, j) whose k-mers have ONE u32 key (kmer_key: equal k-mers for k <= 16, equal last 16
                        characters beyond), pushed for j DEcreasing and, per j, i increasing
                                                                                                        matches/exact.rs:15-69, qgrams.rs:81-97
        push filters    T(start) <= T(target), then local pruning (a diagonal-transition look-ahead over the next p seeds)
                                                                                                        matches.rs:205-247, matches/prepruning.rs:95-203
        transform       T(i, j) = (i - j - P(i), j - i - P(i)), compared componentwise                  seeds.rs:140-143
        layers          val(start) = 1 + score(T(end)) over the active matches whose T(end) <= T(target), starts taken in
                        decreasing (i, j); score(q) = max val over starts s with T(s) >= q, 0 if none   contour/hint_contours.rs:213-272
        h(u)            P(u) - score(T(u)), or max(gap distance, P(u)) when the score is 0              heuristic/csh.rs:341-376, seeds.rs:81-86
        prune_block     per seed a window of examined rows: matches of the seed's column whose start row falls into the
                        block's rows are marked (prune.rs:245-292); the layers are re-derived at the start of the next pass
                        (update_contours, csh.rs:498-554; domain.rs:365-371), so h does not move during a pass.
    The reference keeps the layers incrementally; its update re-scores every layer from the lowest touched one upwards, which is the
    state a fresh construction over the still active matches gives -- that state is what this class computes (with numpy)."""

    def __init__(self, a: bytes, b: bytes, k: int, p: int, prune: bool, transform_filter: bool = True):
        import numpy as np

        self.np = np
        self.a, self.b, self.n, self.m, self.k, self.p, self.prune_on = a, b, len(a), len(b), k, p, prune
        n, m = self.n, self.m
        starts = list(range(0, n - k + 1, k))
        self.seed_starts = starts
        self.P = [0] * (n + 1)  # potential
        self.seed_at = [None] * (n + 1)  # index of the seed covering position i (seeds cover [start, end))
        cur, nxt = 0, len(starts) - 1
        for i in range(n, -1, -1):
            if nxt >= 0:
                if i < starts[nxt] + k:
                    self.seed_at[i] = nxt
                if i == starts[nxt]:
                    cur += 1
                    nxt -= 1
            self.P[i] = cur
        self.t_target = self.T(n, m)
        # ---- find the matches: hash of a's seeds, all k-mers of b looked up for decreasing j ----
        table = {}
        for i in starts:
            table.setdefault(kmer_key(a, i, k), []).append(i)
        self.next_match_per_diag = {}
        kept = []
        for j in range(m - k, -1, -1):
            for i in table.get(kmer_key(b, j, k), ()):
                ts = self.T(i, j)
                if transform_filter and not (ts[0] <= self.t_target[0] and ts[1] <= self.t_target[1]):
                    continue
                if p != 0 and not self._preserve(i, j):
                    continue
                if p != 0:
                    d = i - j
                    old = self.next_match_per_diag.get(d, I_MAX)
                    assert old >= i, "Matches should be added in reverse order on each diagonal"
                    self.next_match_per_diag[d] = i
                kept.append((i, j))
        kept = sorted(set(kept))  # by (i, j); dedup of (start, end)
        self.mi = np.array([x[0] for x in kept], dtype=np.int64)
        self.mj = np.array([x[1] for x in kept], dtype=np.int64)
        self.active = np.ones(len(kept), dtype=bool)
        Parr = np.array(self.P, dtype=np.int64)
        self.Parr = Parr
        self.tsx = self.mi - self.mj - Parr[self.mi] if len(kept) else np.zeros(0, np.int64)
        self.tsy = self.mj - self.mi - Parr[self.mi] if len(kept) else np.zeros(0, np.int64)
        pe = Parr[self.mi + k] if len(kept) else np.zeros(0, np.int64)
        self.tex = self.mi - self.mj - pe
        self.tey = self.mj - self.mi - pe
        # the per-seed windows of prune_block (prune.rs:96-102, 166-189): [before0, before1) and, once visited, [after0, after1)
        self.rng = []
        idx = 0
        for s0 in starts:
            b0 = idx
            while idx < len(kept) and kept[idx][0] == s0:
                idx += 1
            self.rng.append([b0, idx, None, None])
        self.dirty = False
        self._build_layers()

    def T(self, i, j):
        pp = self.P[i]
        return (i - j - pp, j - i - pp)

    # ---- matches/prepruning.rs:95-203 ----
    def _extend_right(self, i, j, end_i):  # -> (i, reached end_i); the word-at-a-time version only overshoots when it returns true
        a, b = self.a, self.b
        if j < 0:
            raise AssertionError("local pruning walked above row 0 (the reference would index out of bounds)")
        while i < end_i and j < self.m and i < self.n and a[i] == b[j]:
            i += 1
            j += 1
        return i, i >= end_i

    def _preserve(self, si, sj):
        k, p, P = self.k, self.p, self.P
        e0, e1 = si + k, sj + k
        start_pot = P[si]
        seed_idx = self.seed_at[si]
        last = self.seed_starts[min(seed_idx + p - 1, len(self.seed_starts) - 1)]
        end_i = last + k
        end_pot = P[end_i]
        pd = start_pot - end_pot
        fr = {pd: e0}
        fr[pd], done = self._extend_right(fr[pd], e1, end_i)
        if done:
            return True
        if self.next_match_per_diag.get(e0 - e1, I_MAX) <= fr[pd]:
            return True
        d0, d1 = pd, pd + 1  # the half-open range of diagonal indices alive
        for g in range(1, pd):
            nf = {}
            for d in range(d0, d1):
                v = fr[d]
                nf[d - 1] = max(nf.get(d - 1, -I_MAX), v)
                nf[d] = max(nf.get(d, -I_MAX), v + 1)
                nf[d + 1] = max(nf.get(d + 1, -I_MAX), v + 1)
            fr = nf
            d0, d1 = d0 - 1, d1 + 1
            while d0 < d1 and g + P[fr[d0]] >= start_pot:
                d0 += 1
            while d0 < d1 and g + P[fr[d1 - 1]] >= start_pot:
                d1 -= 1
            if d0 >= d1:
                return False
            for d in range(d0, d1):
                i = fr[d]
                dd = e0 - e1 + (d - pd)
                j = i - dd
                old_i = i
                fr[d], done = self._extend_right(i, j, end_i)
                if done:
                    return True
                nm = self.next_match_per_diag.get(dd, I_MAX)
                if old_i <= nm <= fr[d]:
                    return True
        return False

    # ---- the layers of the active matches ----
    def _build_layers(self):
        np = self.np
        M = len(self.mi)
        self.val = np.zeros(M, dtype=np.int64)
        tx, ty = self.t_target
        usable = self.active & (self.tex <= tx) & (self.tey <= ty)  # arrows: active, T(end) <= T(target)
        # starts in decreasing (i, j); all matches of one start position share one value (r = 1: one match per start)
        done = np.zeros(M, dtype=bool)
        for t in range(M - 1, -1, -1):
            if usable[t]:
                dom = done & (self.tsx >= self.tex[t]) & (self.tsy >= self.tey[t])
                self.val[t] = 1 + (int(self.val[dom].max()) if dom.any() else 0)
                done[t] = True
        self.in_layers = done
        self.dirty = False

    def score(self, q):
        dom = self.in_layers & (self.tsx >= q[0]) & (self.tsy >= q[1])
        return int(self.val[dom].max()) if dom.any() else 0

    def h(self, i, j):
        sc = self.score(self.T(i, j))
        if sc == 0:
            return max(abs((self.n - i) - (self.m - j)), self.P[i])
        return self.P[i] - sc

    # ---- prune.rs:245-292 (both ranges inclusive: columns i0 + 1 ..= i1 come from the caller's i0..i1) ----
    def prune_block(self, i0, i1, j0, j1):
        assert j0 <= j1
        k = self.k
        first = -(-(i0 + 1) // k)  # first seed with start >= i0 + 1
        for sidx in range(max(first, 0), len(self.seed_starts)):
            col = self.seed_starts[sidx]
            if col > i1:
                break
            r = self.rng[sidx]
            if r[2] is None:
                a0 = a1 = r[1]
                while a0 >= r[0] + 1 and self.mj[a0 - 1] > j1:
                    r[1] -= 1
                    a0 -= 1
                r[2], r[3] = a0, a1
            while r[1] > r[0] and self.mj[r[1] - 1] >= j0:
                self.active[r[1] - 1] = False
                self.dirty = True
                r[1] -= 1
            while r[2] < r[3] and self.mj[r[2]] <= j1:
                self.active[r[2]] = False
                self.dirty = True
                r[2] += 1

    def update_contours(self):
        if self.dirty:
            self._build_layers()


class Block:
    __slots__ = ("v", "i_range", "orig", "j_range", "fixed", "offset", "top", "bot", "j_h")

    def __init__(self):  # block.rs:35-49
        self.v = []
        self.i_range = (-1, 0)
        self.orig = (-W, -W)
        self.j_range = (-W, -W)
        self.fixed = None
        self.offset = 0
        self.top = I_MAX
        self.bot = I_MAX
        self.j_h = None

    def copy(self):
        o = Block()
        o.v, o.i_range, o.orig, o.j_range, o.fixed, o.offset, o.top, o.bot = list(self.v), self.i_range, self.orig, self.j_range, self.fixed, self.offset, self.top, self.bot
        o.j_h = self.j_h
        return o

    def index(self, j):  # block.rs:69-121
        js, je = self.j_range
        assert js <= j, f"Cannot index block {self.i_range} with range {self.j_range} by {j}"
        assert js - self.offset >= 0 and je - self.offset <= len(self.v) * W
        if j > je:
            return self.bot + (j - je)
        v, off = self.v, self.offset
        if j - js < je - j:
            val, j0 = self.top, js
            while j0 + W <= j:
                p, m = v[(j0 - off) // W]
                val += _pc(p) - _pc(m)
                j0 += W
            p, m = v[(j0 - off) // W]
            mask = (1 << (j - j0)) - 1
            return val + _pc(p & mask) - _pc(m & mask)
        val, j1 = self.bot, je
        while j1 - W > j:
            p, m = v[(j1 - W - off) // W]
            val -= _pc(p) - _pc(m)
            j1 -= W
        if j1 > j:
            p, m = v[(j1 - W - off) // W]
            mask = ((1 << W) - 1) ^ ((1 << (W - (j1 - j))) - 1)
            val -= _pc(p & mask) - _pc(m & mask)
        return val

    def get(self, j):  # block.rs:125-130
        if j < self.j_range[0] or j > self.j_range[1]:
            return None
        return self.index(j)

    def get_diff(self, j):  # block.rs:133-144
        if j < self.offset:
            return None
        idx = (j - self.offset) // W
        if idx >= len(self.v):
            return None
        bit = (j - self.offset) % W
        p, m = self.v[idx]
        return ((p >> bit) & 1) - ((m >> bit) & 1)


class Restated:
    def __init__(self, a: bytes, b: bytes, heuristic: str = "gap", k: int = 12, sparse_h: bool = True, block_width: int = 256,
                 dt_trace: bool = True, max_g: int = 40, fr_drop: int = 10, domain: str = "astar", sparse: bool = True,
                 doubling: str = "band", start: str = "h0", factor: float = 2.0, delta: float = 1.0, incremental_doubling: bool = False,
                 p: int = 0, prune: bool = False, trace: bool = True):
        assert all(c in b"ACGT" for c in a) and all(c in b"ACGT" for c in b)
        assert domain in ("astar", "full", "gap_start", "gap_gap") and doubling in ("band", "linear", "none") and start in ("zero", "gap", "h0")
        self.a, self.b, self.n, self.m = a, b, len(a), len(b)
        self.domain, self.sparse, self.doubling, self.start, self.factor, self.delta = domain, sparse, doubling, start, factor, delta
        self.kind, self.sparse_h, self.bw = (heuristic if domain == "astar" else "none"), sparse_h, block_width
        self.dt, self.max_g, self.fr_drop = dt_trace, max_g, fr_drop
        self.sh = sh_table(a, b, k, p) if self.kind == "sh" else None
        self.prune = prune and self.kind == "gcsh"
        self.gcsh = Gcsh(a, b, k, p, prune) if self.kind == "gcsh" else None
        self.peq = {c: sum(1 << j for j in range(self.m) if b[j] == c) for c in set(a)}  # rows >= m never match (profile.rs:127-132)
        # Blocks (blocks.rs:87-107)
        self.incremental = incremental_doubling
        self.trace_mode = trace  # false: the cost-only arms of Blocks (blocks.rs:160-171, 252-277) -- one block updated in place
        self.hrow = [(0, 0)] * self.n if incremental_doubling else []  # horizontal differences (p, m) of the row j_h, per column
        self.blocks: list[Block] = []
        self.last = 0
        self.i_range = (-1, 0)
        self.st = dict(num_blocks=0, num_incremental_blocks=0, computed_lanes=0, unique_lanes=0, f_max_tries=0, dt_trace_tries=0,
                       dt_trace_success=0, dt_trace_fallback=0, fill_tries=0, fill_success=0, fill_fallback=0)

    # ---- the heuristic ----
    def h(self, i, j):
        if self.kind == "gap":
            return abs((self.n - i) - (self.m - j))
        if self.kind == "sh":
            return self.sh[i]
        if self.kind == "gcsh":
            return self.gcsh.h(i, j)
        return 0

    # ---- one Myers step per column on an integer `rows` bits tall (myers.rs:27-55) ----
    # hin: None (+1 into the top row of every column) or one (p, m) pair per column.  -> (sum of the bottom row's differences, the new
    # vertical words, every column's words if `keep`, the bottom row's (p, m) per column)
    def _columns(self, i0, i1, w0, vwords, keep, hin=None):
        rows = len(vwords) * W
        full = (1 << rows) - 1
        vp = sum(p << (W * t) for t, (p, _) in enumerate(vwords))
        vm = sum(m << (W * t) for t, (_, m) in enumerate(vwords))
        sh = rows - 1
        m64 = (1 << W) - 1
        hsum, cols, hout = 0, [], []
        for i in range(i0, i1):
            hp0, hm0 = (1, 0) if hin is None else hin[i - i0]
            eq = (self.peq.get(self.a[i], 0) >> (w0 * W)) & full
            xv = eq | vm
            eq |= hm0  # a -1 coming in acts like a match in the top row
            xh = ((((eq & vp) + vp) & full) ^ vp) | eq
            ph = vm | (full & ~(xh | vp))
            mh = vp & xh
            o = ((ph >> sh) & 1, (mh >> sh) & 1)
            hsum += o[0] - o[1]
            hout.append(o)
            ph = ((ph << 1) | hp0) & full
            mh = ((mh << 1) | hm0) & full
            vp = mh | (full & ~(xv | ph))
            vm = ph & xv
            if keep:
                cols.append([((vp >> (W * t)) & m64, (vm >> (W * t)) & m64) for t in range(len(vwords))])
        out = [((vp >> (W * t)) & m64, (vm >> (W * t)) & m64) for t in range(len(vwords))]
        return hsum, out, cols, hout

    def compute_block(self, i_range, v_range, v, lo=0, mode="none"):
        """blocks.rs:686-748: columns i_range over the words v_range, which sit at v[lo : lo + len]; returns the sum of the bottom row's
        horizontal differences.  mode (HMode): none = +1 into the top row, nothing kept; output = +1 in, the bottom row's differences
        stored in self.hrow; input = self.hrow into the top row, nothing kept; update = self.hrow in and replaced."""
        i0, i1 = i_range
        nw = v_range[1] - v_range[0]
        if i1 - i0 > 1:
            self.st["computed_lanes"] += nw
            self.st["num_incremental_blocks"] += 1
        hin = None if mode in ("none", "output") else self.hrow[i0:i1]
        if nw == 0:  # no words: what comes in at the top goes out at the bottom
            hout = [(1, 0)] * (i1 - i0) if hin is None else list(hin)
            hsum = sum(p - m for p, m in hout)
        else:
            hsum, out, _, hout = self._columns(i0, i1, v_range[0], v[lo:lo + nw], False, hin)
            v[lo:lo + nw] = out
        if mode in ("output", "update"):
            self.hrow[i0:i1] = hout
        return hsum

    # ---- Blocks ----
    def last_block(self):
        return self.blocks[self.last]

    def next_block_j_range(self):  # blocks.rs:549-551
        return self.blocks[self.last + 1].j_range if self.last + 1 < len(self.blocks) else None

    def set_last_block_fixed_j_range(self, fixed):  # blocks.rs:554-568
        bl = self.blocks[self.last]
        bl.fixed = _union(bl.fixed, fixed) if (bl.fixed is not None and fixed is not None) else fixed

    def init(self, initial):  # blocks.rs:146-179, trace mode
        assert initial[0] == 0
        self.last = 0
        self.i_range = (-1, 0)
        fixed = initial
        if self.blocks:
            initial = _union(initial, self.blocks[0].j_range)
        initial = _round_out(initial)
        bl = Block()  # Block::first_col (block.rs:53-66)
        assert initial[0] == 0
        bl.v = [ONE] * ((initial[1] - initial[0]) // W)
        bl.i_range, bl.orig, bl.j_range, bl.fixed, bl.offset, bl.top, bl.bot = (-1, 0), fixed, initial, fixed, 0, 0, initial[1] - initial[0]
        if not self.trace_mode:  # one block spanning the entire first column (blocks.rs:160-171)
            bl.v = [ONE] * (-(-self.m // W))
            bl.bot = initial[1]
        if not self.blocks:
            self.blocks.append(bl)
        else:
            self.blocks[0] = bl

    def pop_last_block(self):  # blocks.rs:182-185
        r = self.blocks[self.last].i_range
        assert self.i_range[1] == r[1]
        self.i_range = (self.i_range[0], r[0])
        self.last -= 1

    def _push_i(self, r):
        assert self.i_range[1] == r[0]
        self.i_range = (self.i_range[0], r[1])

    def reuse_next_block(self, i_range, j_range):  # blocks.rs:190-197
        self._push_i(i_range)
        self.last += 1
        bl = self.blocks[self.last]
        assert bl.i_range == i_range and bl.j_range == _round_out(j_range)

    @staticmethod
    def init_v_with_overlap(prev: Block, nxt: Block):  # blocks.rs:753-769
        assert nxt.offset == nxt.j_range[0] and prev.offset == prev.j_range[0]
        pv0 = prev.j_range[0] // W
        v0, v1 = nxt.j_range[0] // W, nxt.j_range[1] // W
        nxt.v = [ONE] * (v1 - v0)
        o0, o1 = max(nxt.j_range[0], prev.j_range[0]) // W, min(nxt.j_range[1], prev.j_range[1]) // W
        assert o0 <= o1, "ranges of consecutive blocks overlap"
        nxt.v[o0 - v0:o1 - v0] = prev.v[o0 - pv0:o1 - pv0]

    def compute_next_block(self, i_range, j_range):  # blocks.rs:205-469 (trace mode)
        self.st["num_blocks"] += 1
        orig = j_range
        jr = _round_out(j_range)
        v_range = (jr[0] // W, jr[1] // W)
        self.st["unique_lanes"] += v_range[1] - v_range[0]
        if self.last + 1 < len(self.blocks):
            old = self.blocks[self.last + 1].j_range
            assert jr[0] <= old[0] and old[1] <= jr[1], "j_range must grow"
            self.st["unique_lanes"] -= (old[1] - old[0]) // W
        if not self.sparse:  # trace && !sparse: every column of the block is kept (blocks.rs:231-240)
            self.fill_with_blocks(i_range, orig)
            return
        self._push_i(i_range)
        prev_top = self.last_block().index(jr[0])
        prev_bot = self.last_block().index(jr[1])
        if not self.trace_mode and not self.incremental:  # blocks.rs:252-277: the single block's v updated in place
            bl = self.blocks[self.last]
            bot = prev_bot + self.compute_block(i_range, v_range, bl.v, v_range[0] - bl.offset // W)
            bl.i_range, bl.orig, bl.j_range = i_range, orig, jr
            bl.top, bl.bot = prev_top + (i_range[1] - i_range[0]), bot
            return
        if self.last + 1 == len(self.blocks):
            self.blocks.append(Block())
        else:
            assert self.blocks[self.last + 1].i_range == i_range
        prev, nxt = self.blocks[self.last], self.blocks[self.last + 1]
        self.last += 1
        old = nxt.copy()  # ("copy settings, but not the vector": the vector stays with nxt)
        # the block is overwritten in place: its v memory and -- note -- its fixed_j_range survive (blocks.rs:303-316)
        nxt.i_range, nxt.orig, nxt.j_range, nxt.offset, nxt.j_h = i_range, orig, jr, jr[0], None
        nxt.top = prev_top + (i_range[1] - i_range[0])
        nxt.bot = prev_bot
        if not self.incremental or prev.fixed is None:
            self.init_v_with_overlap(prev, nxt)
            nxt.bot += self.compute_block(i_range, v_range, nxt.v)
            return
        # ---- incremental doubling (blocks.rs:341-469) ----
        new_j_h = prev.fixed[1] // W * W  # prev_fixed.round_in().1
        nxt.j_h = new_j_h
        off = v_range[0]
        up64 = lambda x: -(-x // W) * W
        if old.j_h is not None and old.fixed is not None and up64(old.fixed[0] - 1) < old.j_h:
            self.init_v_with_overlap_preserve_fixed(prev, old, nxt)
            r0 = _round_out((jr[0], old.fixed[0] - 1))
            vr0 = (r0[0] // W, r0[1] // W)
            assert vr0[0] <= vr0[1]
            assert old.j_h % W == 0 and new_j_h % W == 0 and jr[1] % W == 0
            vr1 = (old.j_h // W, new_j_h // W)
            assert vr1[0] <= vr1[1], "j_h may only increase!"
            vr2 = (new_j_h // W, jr[1] // W)
            assert vr2[0] <= vr2[1]
            self.compute_block(i_range, vr0, nxt.v, vr0[0] - off, "none")
            if vr1[1] > vr1[0]:
                self.compute_block(i_range, vr1, nxt.v, vr1[0] - off, "update")
            nxt.bot += self.compute_block(i_range, vr2, nxt.v, vr2[0] - off, "input")
        else:
            self.init_v_with_overlap(prev, nxt)
            assert jr[0] % W == 0 and new_j_h % W == 0
            vr01 = (jr[0] // W, new_j_h // W)
            assert vr01[0] <= vr01[1]
            vr2 = (new_j_h // W, jr[1] // W)
            assert vr2[0] <= vr2[1]
            self.compute_block(i_range, vr01, nxt.v, vr01[0] - off, "output")
            nxt.bot += self.compute_block(i_range, vr2, nxt.v, vr2[0] - off, "input")

    @staticmethod
    def init_v_with_overlap_preserve_fixed(prev: Block, old: Block, nxt: Block):  # blocks.rs:776-831
        v = nxt.v  # still the old block's words
        assert prev.offset == prev.j_range[0] and old.offset == old.j_range[0] and nxt.offset == nxt.j_range[0]
        assert nxt.j_range[0] <= old.j_range[0] and old.j_range[1] <= nxt.j_range[1]
        pv = (prev.j_range[0] // W, prev.j_range[1] // W)
        ov = (old.j_range[0] // W, old.j_range[1] // W)
        nv = (nxt.j_range[0] // W, nxt.j_range[1] // W)
        assert pv[0] <= nv[0] <= ov[0]
        preserve = (-(-(old.fixed[0] - 1) // W), old.j_h // W)  # JRange(old_fixed.0 - 1, old_j_h).round_in().v_range()
        assert preserve[0] < preserve[1]
        # 1. resize (Vec::resize keeps the front, pads with V::one() or truncates)
        n = nv[1] - nv[0]
        if len(v) < n:
            v.extend([ONE] * (n - len(v)))
        else:
            del v[n:]
        # 2. move the preserved words to where they belong in the new range
        if nv[0] != ov[0]:
            chunk = v[preserve[0] - ov[0]:preserve[1] - ov[0]]
            assert len(chunk) == preserve[1] - preserve[0]
            v[preserve[0] - nv[0]:preserve[0] - nv[0] + len(chunk)] = chunk
        # 3. prefix and suffix from the previous block
        k = preserve[0] - nv[0]
        src = prev.v[nv[0] - pv[0]:preserve[0] - pv[0]]
        assert len(src) == k
        v[:k] = src
        copy_end = min(nv[1], pv[1])
        assert copy_end >= preserve[1]
        src = prev.v[preserve[1] - pv[0]:copy_end - pv[0]]
        assert len(src) == copy_end - preserve[1]
        v[preserve[1] - nv[0]:copy_end - nv[0]] = src
        # 4. the rest: +1
        for t in range(copy_end - nv[0], n):
            v[t] = ONE
        assert len(v) == n

    def fill_with_blocks(self, i_range, original_j_range):  # blocks.rs:571-660
        jr = _round_out(original_j_range)
        self._push_i(i_range)
        v_range = (jr[0] // W, jr[1] // W)
        prev = self.blocks[self.last]
        assert prev.i_range[1] == i_range[0]
        nb = Block()
        nb.i_range, nb.orig, nb.j_range, nb.offset, nb.fixed = (i_range[0], i_range[0]), original_j_range, jr, jr[0], None
        nb.top, nb.bot = prev.index(jr[0]), 0
        self.init_v_with_overlap(prev, nb)
        bot = prev.index(jr[1])
        if v_range[1] > v_range[0]:
            _, _, cols, hpairs = self._columns(i_range[0], i_range[1], v_range[0], nb.v, True)
            hvals = [p - m for p, m in hpairs]
        else:
            cols, hvals = [[] for _ in range(i_range[0], i_range[1])], [1] * (i_range[1] - i_range[0])
        for t, i in enumerate(range(i_range[0], i_range[1])):
            nb.i_range = (i, i + 1)
            nb.top += 1
            self.last += 1
            bl = nb.copy()
            bl.v = cols[t]
            bot += hvals[t]
            bl.bot = bot
            if self.last == len(self.blocks):
                self.blocks.append(bl)
            else:
                self.blocks[self.last] = bl

    # ---- domain.rs:90-246 ----
    def j_range(self, i_range, f_max, prev: Block, old_range):
        is_, ie = i_range
        if f_max is None or self.domain == "full":
            rng = (0, self.m)
            if f_max is None:
                return rng
        elif self.domain == "gap_start":  # unit costs: f_max insertions / deletions at most
            rng = (is_ + 1 - f_max, ie + f_max)
        elif self.domain == "gap_gap":
            d = self.m - self.n
            sres = f_max - abs(self.n - self.m)
            extra = int(sres / 2)  # i32 division truncates towards zero
            rng = (is_ + 1 + min(d, 0) - extra, ie + max(d, 0) + extra)
        if self.domain != "astar":
            if old_range is not None:
                rng = _union(rng, old_range)
            return (max(rng[0], 0), min(rng[1], self.m))
        fixed_start, fixed_end = prev.fixed
        assert fixed_start <= fixed_end, "Fixed range must not be empty"
        u0, u1 = is_, fixed_end
        gu = 0 if is_ < 0 else prev.index(fixed_end)
        v0, v1 = u0, u1

        def f(x, y):
            assert y - u1 >= x - u0
            return gu + abs((x - u0) - (y - u1)) + self.h(x, y)  # extend_cost under unit costs = the gap between the diagonals

        if not self.sparse_h:
            while v0 < ie:
                v0 += 1
                v1 += 1
                v1 += 1
                while v1 <= self.m and f(v0, v1) <= f_max:
                    v1 += 1
                v1 -= 1
        else:
            v0 += 1
            v1 += 1
            v1 += self.bw
            v1 = min(v1, self.m)
            while True:
                if v1 < v0 - u0 + u1:
                    v1 = v0 - u0 + u1
                    break
                fv = f(v0, v1)
                if fv <= f_max:
                    if v1 == self.m:
                        break
                    v1 += 8
                    if v1 >= self.m:
                        v1 = self.m
                else:
                    v0 += -((f_max - fv) // 2)  # (fv - f_max).div_ceil(2)
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
        rng = (fixed_start, v1)
        if old_range is not None:
            rng = _union(rng, old_range)
        return (max(rng[0], 0), min(rng[1], self.m))

    # ---- domain.rs:251-350 ----
    def fixed_j_range(self, i, f_max, prev_fixed, block: Block):
        if self.domain != "astar" or f_max is None:
            return None
        f = lambda j: block.index(j) + self.h(i, j)
        assert block.j_range[0] <= prev_fixed[0]
        start, end = prev_fixed[0], min(block.orig[1], self.m)
        while start <= end:
            fv = f(start)
            if fv <= f_max:
                break
            start += -((f_max - fv) // 2) if self.sparse_h else 1
        while end >= start:
            fv = f(end)
            if fv <= f_max:
                break
            end -= -((f_max - fv) // 2) if self.sparse_h else 1
        fixed = (start, end)
        if block.fixed is not None:
            fixed = block.fixed if _empty(fixed) else _union(fixed, block.fixed)
        return fixed

    # ---- domain.rs:356-541 with trace = true ----
    def align_for_bounded_dist(self, f_max):
        self.st["f_max_tries"] += 1
        if self.prune:
            self.gcsh.update_contours()  # pending prunes take effect now (domain.rs:365-371)
        assert f_max is None or f_max >= 0
        first = Block()
        first.fixed = (-1, -1)
        initial = self.j_range((-1, 0), f_max, first, self.next_block_j_range())
        if _empty(initial) or initial[0] > 0:
            return None
        self.init(initial)
        self.set_last_block_fixed_j_range(initial)
        all_reused = True
        for i in range(0, self.n, self.bw):
            i_range = (i, min(i + self.bw, self.n))
            jr = self.j_range(i_range, f_max, self.last_block(), self.next_block_j_range())
            if _empty(jr):
                assert self.next_block_j_range() is None
                return None
            reuse = self.next_block_j_range() == jr and all_reused  # (a rounded range against the exact new one, as in the Rust)
            all_reused = all_reused and reuse
            prev_fixed = self.last_block().fixed
            if reuse:
                self.reuse_next_block(i_range, jr)
            else:
                self.compute_next_block(i_range, jr)
            nf = self.fixed_j_range(i_range[1], f_max, prev_fixed, self.last_block())
            if nf is not None and _empty(nf):
                return None
            self.set_last_block_fixed_j_range(nf)
            if self.prune:  # matches starting in the columns of this block, rows fixed before AND after it (domain.rs:505-515)
                inter = (max(prev_fixed[0], nf[0]), min(prev_fixed[1], nf[1]))
                if not _empty(inter):
                    self.gcsh.prune_block(i_range[0], i_range[1], inter[0], inter[1])
        dist = self.last_block().get(self.m)
        if dist is None:
            return None
        if self.trace_mode and (f_max is None or dist <= f_max):
            return dist, self.trace((0, 0), (self.n, self.m))
        return dist, None

    def _stats(self):
        """AstarPa2Stats as cost_or_align returns them: only the BandDoubling arm copies the Blocks' own counters into the result
        (`nw.stats.block_stats = blocks.stats`, lib.rs:158); the other arms leave them at their defaults."""
        st = dict(self.st)
        if self.doubling != "band":
            st.update(num_blocks=0, num_incremental_blocks=0, computed_lanes=0, unique_lanes=0)
        return st

    # ---- lib.rs:122-175 + band.rs:11-24, 100-182 ----
    def align(self):
        h0 = self.h(0, 0)
        if self.doubling == "none":
            assert self.domain == "full"
            cost, cigar = self.align_for_bounded_dist(None)
            return cost, cigar, self._stats()
        gap = abs(self.n - self.m)
        start_f, start_inc = {"zero": (0, 1), "gap": (gap, gap), "h0": (h0, 1)}[self.start]
        if self.doubling == "linear":
            last_s, s, maxs = -1, start_f, I_MAX
            step = int(self.delta)  # `delta as Cost`
        else:
            offset = start_f
            last_s, s, maxs = -1, offset + max(start_inc, self.bw), I_MAX
        while True:
            r = self.align_for_bounded_dist(s)
            if r is not None:
                cost, cigar = r
                assert cost <= maxs
                if cost <= s:
                    assert cost > last_s
                    assert h0 <= cost
                    return cost, cigar, self._stats()
                maxs = min(maxs, cost)
            else:
                assert maxs == I_MAX
            last_s = s
            if self.doubling == "linear":
                s = min(s + step, maxs)
            else:
                s = max(int(math.ceil(_f32(_f32(self.factor) * _f32(s - offset)))), 1) + offset
                s = min(s, maxs)

    # ---- blocks/trace.rs ----
    def trace(self, frm, to):  # trace.rs:16-143
        assert self.blocks[-1].i_range[1] == to[0]
        ops = []  # (op, cnt), pushed with merging (pa-affine-types cigar.rs:137-146 has the same rule)

        def push(op, cnt):
            if ops and ops[-1][0] == op:
                ops[-1][1] += cnt
            else:
                ops.append([op, cnt])

        g = [self.blocks[self.last].index(to[1])]
        while to != frm:
            while self.last > 0 and self.blocks[self.last].i_range[0] >= to[0]:
                self.pop_last_block()
            if self.dt and to[0] > 0:
                prev = self.blocks[self.last - 1]
                if prev.i_range[1] < to[0] - 1:
                    self.st["dt_trace_tries"] += 1
                    new_to = self.dt_trace_block(to, g, prev, push)
                    if new_to is not None:
                        self.st["dt_trace_success"] += 1
                        to = new_to
                        continue
                    self.st["dt_trace_fallback"] += 1
            if self.sparse and to[0] > 0:
                block = self.blocks[self.last]
                prev = self.blocks[self.last - 1]
                assert prev.i_range[1] < to[0] <= block.i_range[1]
                if prev.i_range[1] < to[0] - 1 or block.i_range[1] > to[0]:
                    prev_j_range = prev.j_range
                    i_range = (prev.i_range[1], to[0])
                    j_range = (block.j_range[0], to[1])
                    self.pop_last_block()
                    height = min(j_range[1] - j_range[0], (i_range[1] - i_range[0]) * 5 // 4)
                    while True:
                        jr = _round_out((max(j_range[1] - height, prev_j_range[0]), j_range[1]))
                        self.st["fill_tries"] += 1
                        self.fill_with_blocks(i_range, jr)
                        if self.blocks[self.last].index(to[1]) == g[0]:
                            self.st["fill_success"] += 1
                            break
                        self.st["fill_fallback"] += 1
                        assert jr[0] != 0, f"No trace found through block {i_range} {jr}"
                        for _ in range(i_range[0], i_range[1]):
                            self.pop_last_block()
                        height *= 2
            to, (op, cnt) = self.parent(to, g)
            push(op, cnt)
        assert g[0] == 0
        ops.reverse()
        return "".join((str(c) if c != 1 else "") + o for o, c in ops)

    def parent(self, st, g):  # trace.rs:145-228
        block = self.blocks[self.last]
        assert block.i_range[1] == st[0], f"Parent of state {st} but block.i is {block.i_range}"
        i, j = st
        cnt = 0
        while i > 0 and j > 0 and self.a[i - 1] == self.b[j - 1]:
            cnt += 1
            i -= 1
            j -= 1
        if cnt > 0:
            return (i, j), ("=", cnt)
        if block.get_diff(j - 1) == 1:  # vertical delta: an insertion
            g[0] -= 1
            return (i, j - 1), ("I", 1)
        prev = self.blocks[self.last - 1]
        assert prev.i_range[1] == i - 1
        hd = 1 if j < prev.j_range[0] else g[0] - prev.index(j)
        if hd == 1:
            g[0] -= 1
            return (i - 1, j), ("D", 1)
        if j > prev.j_range[1]:
            assert j == prev.j_range[1] + 1
            dd = 1
        else:
            dd = prev.get_diff(j - 1) + hd
        if dd == 1:
            g[0] -= 1
            return (i - 1, j - 1), ("X", 1)
        raise AssertionError(f"ERROR: PARENT OF {st} NOT FOUND IN TRACEBACK")

    def _extend_left(self, i, i0, j):  # trace.rs:445-500 -> (i, j, count)
        cnt = 0
        while i > i0 and j > 0 and self.a[i - 1] == self.b[j - 1]:
            i -= 1
            j -= 1
            cnt += 1
        return i, j, cnt

    def dt_trace_block(self, st, g_st, prev: Block, push):  # trace.rs:231-418
        block_start = prev.i_range[1]
        fr = {(0, 0): [st[0], 0, 0]}  # (g, d) -> [i, ext, parent_d]; the furthest (leftmost) column at distance g on diagonal d

        def extend_and_check(e, j, target_g):
            e[0], j, c = self._extend_left(e[0], prev.i_range[1], j)
            e[1] += c
            return e[0] == prev.i_range[1] and prev.get(j) == target_g

        def emit(g, d):
            new_st = (block_start, st[1] - (st[0] - block_start) - d)
            g_st[0] -= g
            out = []
            while True:
                e = fr[(g, d)]
                if e[1] > 0:
                    out.append(("=", e[1]))
                if g == 0:
                    break
                g -= 1
                d += e[2]
                out.append(({-1: "I", 0: "X", 1: "D"}[e[2]], 1))
            for op, c in reversed(out):
                push(op, c)
            return new_st

        g = 0
        if extend_and_check(fr[(0, 0)], st[1], g_st[0]):
            return emit(0, 0)
        d0, d1 = 0, 0
        while True:
            ng = g + 1
            for d in range(d0 - 1, d1 + 2):
                fr[(ng, d)] = [I_MAX, 0, 0]
            for d in range(d0, d1 + 1):
                e = fr[(g, d)]

                def update(x, y, pd):
                    if y < x[0]:
                        x[0] = y
                        x[2] = pd

                update(fr[(ng, d - 1)], e[0] - 1, 1)
                update(fr[(ng, d)], e[0] - 1, 0)
                update(fr[(ng, d + 1)], e[0], -1)
            g += 1
            d0 -= 1
            d1 += 1
            min_fr, min_i = I_MAX, I_MAX
            for d in range(d0, d1 + 1):
                e = fr[(g, d)]
                if e[0] == I_MAX:
                    continue
                j = st[1] - (st[0] - e[0]) - d
                if extend_and_check(e, j, g_st[0] - g):
                    return emit(g, d)
                min_fr = min(min_fr, 2 * e[0] - d)
                min_i = min(min_i, e[0])
            if g == self.max_g // 2 and min_i > (block_start + st[0]) // 2:
                return None
            if g == self.max_g:
                return None
            if self.fr_drop > 0:
                while d0 < d1 and (fr[(g, d0)][0] <= block_start or 2 * fr[(g, d0)][0] - d0 > min_fr + self.fr_drop):
                    d0 += 1
                while d0 < d1 and (fr[(g, d1)][0] <= block_start or 2 * fr[(g, d1)][0] - d1 > min_fr + self.fr_drop):
                    d1 -= 1
                if d0 > d1:
                    return None


def align(a: bytes, b: bytes, **kw):
    """-> (cost, CIGAR string, statistics) of the restated traced A*PA2 (`simple` family; see the module header)."""
    return Restated(a, b, **kw).align()
