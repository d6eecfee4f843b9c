"""A standalone restatement of the reference's COST-ONLY, non-incremental A*PA2 mode -- AstarPa2Params::simple()-like parameters with
`make_aligner(false)` -- written from the Rust text alone and independent of csrc/engine.hpp:
    astarpa2/src/lib.rs:122-175      cost_or_align            band.rs:100-141     exponential_search
    astarpa2/src/domain.rs:117-350   j_range, fixed_j_range   domain.rs:356-541   align_for_bounded_dist
    astarpa2/src/blocks.rs:146-179   Blocks::init (the `!trace` arm: ONE block over the whole column)
    astarpa2/src/blocks.rs:205-277   compute_next_block (the `!trace && !incremental_doubling` arm: the block's v updated in place)
    astarpa2/src/block.rs:69-131     Block::index / get
Pure Python; the DP of a block runs on Python big integers (one Myers word as tall as the block's rows), so no kernel of this
repository is involved either.  Purpose (review of round 2, weak #2): does the reference's cost-only mode really end on an upper
bound (11 353 for the pair below, whose distance is 11 325), or does engine.hpp mis-restate it?

Usage: python tests/tools/cost_only_restatement.py            (the pair the GPU soak of round 2 found: SH k = 12, 57 373 x 55 438)
Result recorded in DESIGN.md 3a.
"""
import math
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

W = 64
BLOCK = 256


def sh_table(a: bytes, b: bytes, k: int):
    """SH for exact matches (pa-heuristic sh.rs:47-106, matches/exact.rs, qgrams.rs:30-43): h(i) = seeds starting at >= i without a match."""
    n = len(a)
    bits = lambda c: (c >> 1) & 3
    def key(s):
        q = 0
        for c in s:
            q = (q << 2) | bits(c)
        return q & 0xFFFFFFFF
    bk = {key(b[j:j + k]) for j in range(0, len(b) - k + 1)} if len(b) >= k else set()
    nseeds = (n - k) // k + 1 if n >= k else 0
    matched = [key(a[s * k:(s + 1) * k]) in bk for s in range(nseeds)]
    h = [0] * (n + 1)
    unmatched, nxt = 0, nseeds - 1
    for i in range(n, -1, -1):
        if nxt >= 0 and i == nxt * k:
            if not matched[nxt]:
                unmatched += 1
            nxt -= 1
        h[i] = unmatched
    return h


class CostOnly:
    def __init__(self, a: bytes, b: bytes, heuristic: str, k: int = 12, sparse_h: bool = True):
        self.a, self.b, self.n, self.m = a, b, len(a), len(b)
        self.w = (self.m + W - 1) // W
        self.sparse_h = sparse_h
        self.kind = heuristic
        self.sh = sh_table(a, b, k) if heuristic == "sh" else None
        # eq masks of b per character over ALL rows (pad rows never match, profile.rs:127-132)
        self.peq = {c: sum(1 << j for j in range(self.m) if b[j] == c) for c in set(a)}
        self.block = None  # the single block (blocks.rs:160-171)
        self.f_max_tries = 0
        self.computed_lanes = 0

    def h(self, i, j):
        if self.kind == "gap":
            return abs((self.n - i) - (self.m - j))
        if self.kind == "sh":
            return self.sh[i]
        return 0

    # ---- Block::index / get (block.rs:69-131), offset 0, v = list of (p, m) words over the whole column ----
    def index(self, j):
        bl = self.block
        js, je = bl["j_range"]
        assert js <= j, "Cannot index block below its range"
        if j > je:
            return bl["bot"] + (j - je)
        v = bl["v"]
        pc = lambda x: bin(x).count("1")
        if j - js < je - j:
            val, j0 = bl["top"], js
            while j0 + W <= j:
                val += pc(v[j0 // W][0]) - pc(v[j0 // W][1])
                j0 += W
            mask = (1 << (j - j0)) - 1
            return val + pc(v[j0 // W][0] & mask) - pc(v[j0 // W][1] & mask)
        val, j1 = bl["bot"], je
        while j1 - W > j:
            val -= pc(v[(j1 - W) // W][0]) - pc(v[(j1 - W) // W][1])
            j1 -= W
        if j1 > j:
            cnt = j1 - j
            mask = ((1 << W) - 1) ^ ((1 << (W - cnt)) - 1)
            val -= pc(v[(j1 - W) // W][0] & mask) - pc(v[(j1 - W) // W][1] & mask)
        return val

    # ---- compute_block(HMode::None) on words [w0, w1) x columns [i0, i1), in place on the block's v (blocks.rs:686-748) ----
    def compute(self, i0, i1, w0, w1):
        if i1 - i0 > 1:
            self.computed_lanes += w1 - w0
        if w1 == w0:
            return i1 - i0
        v = self.block["v"]
        rows = (w1 - w0) * W
        full = (1 << rows) - 1
        vp = sum(v[w0 + t][0] << (W * t) for t in range(w1 - w0))
        vm = sum(v[w0 + t][1] << (W * t) for t in range(w1 - w0))
        top = 1 << (rows - 1)
        bottom_sum = 0
        for i in range(i0, i1):
            eq = (self.peq.get(self.a[i], 0) >> (w0 * W)) & full
            # Myers / Hyyro step with horizontal input +1 at the top row (myers.rs:27-55)
            xv = eq | vm
            xh = ((((eq & vp) + vp) & full) ^ vp) | eq
            ph = vm | (full & ~(xh | vp))
            mh = vp & xh
            bottom_sum += (1 if ph & top else 0) - (1 if mh & top else 0)
            ph = ((ph << 1) | 1) & full
            mh = (mh << 1) & full
            vp = mh | (full & ~(xv | ph))
            vm = ph & xv
        m64 = (1 << W) - 1
        for t in range(w1 - w0):
            v[w0 + t] = ((vp >> (W * t)) & m64, (vm >> (W * t)) & m64)
        return bottom_sum

    # ---- domain.rs:117-246 (Domain::Astar) ----
    def j_range(self, is_, ie, f_max, prev_fixed, old_range):
        fixed_start, fixed_end = prev_fixed
        assert fixed_start <= fixed_end
        u0, u1 = is_, fixed_end
        gu = 0 if is_ < 0 else self.index(fixed_end)
        v0, v1 = u0, u1
        f = lambda x, y: gu + abs((x - u0) - (y - u1)) + self.h(x, y)
        if not self.sparse_h:
            while v0 < ie:
                v0 += 1
                v1 += 2
                while v1 <= self.m and f(v0, v1) <= f_max:
                    v1 += 1
                v1 -= 1
        else:
            v0 += 1
            v1 += 1
            v1 = min(v1 + BLOCK, self.m)
            while True:
                if v1 < v0 - u0 + u1:
                    v1 = v0 - u0 + u1
                    break
                fv = f(v0, v1)
                if fv <= f_max:
                    if v1 == self.m:
                        break
                    v1 = min(v1 + 8, self.m)
                else:
                    v0 += -((f_max - fv) // 2)  # ceil((fv - f_max) / 2)
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
        s, e = fixed_start, v1
        if old_range is not None:
            s, e = min(s, old_range[0]), max(e, old_range[1])
        return max(s, 0), min(e, self.m)

    # ---- domain.rs:251-350 ----
    def fixed_j_range(self, i, f_max, prev_fixed):
        bl = self.block
        f = lambda j: self.index(j) + self.h(i, j)
        assert bl["j_range"][0] <= prev_fixed[0]
        start, end = prev_fixed[0], min(bl["orig"][1], self.m)
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
        old = bl["fixed"]
        if old is not None:
            fixed = old if fixed[0] > fixed[1] else (min(fixed[0], old[0]), max(fixed[1], old[1]))
        return fixed

    # ---- domain.rs:356-541 with trace = false ----
    def align_for_bounded_dist(self, f_max):
        self.f_max_tries += 1
        rnd = lambda r: (r[0] // W * W, -(-r[1] // W) * W)
        initial = self.j_range(-1, 0, f_max, (-1, -1), None)  # blocks.next_block_j_range(): the Vec holds one block -> None
        if initial[0] > initial[1] or initial[0] > 0:
            return None
        fixed = initial
        rng0 = initial
        if self.block is not None:  # initial_j_range.union(blocks[0].j_range): blocks[0] IS the single block of the previous pass
            rng0 = (min(rng0[0], self.block["j_range"][0]), max(rng0[1], self.block["j_range"][1]))
        rng0 = rnd(rng0)
        self.block = {"v": [((1 << W) - 1, 0)] * self.w, "orig": fixed, "j_range": rng0, "fixed": fixed, "top": 0, "bot": rng0[1]}
        # set_last_block_fixed_j_range(initial): union with itself
        for i in range(0, self.n, BLOCK):
            i0, i1 = i, min(i + BLOCK, self.n)
            jr = self.j_range(i0, i1, f_max, self.block["fixed"], None)
            if jr[0] > jr[1]:
                return None
            prev_fixed = self.block["fixed"]
            # compute_next_block, cost-only arm (blocks.rs:252-277)
            rj = rnd(jr)
            prev_top = self.index(rj[0])
            prev_bot = self.index(rj[1])
            bot = prev_bot + self.compute(i0, i1, rj[0] // W, rj[1] // W)
            self.block.update({"orig": jr, "j_range": rj, "top": prev_top + (i1 - i0), "bot": bot})
            nf = self.fixed_j_range(i1, f_max, prev_fixed)
            if nf[0] > nf[1]:
                return None
            old = self.block["fixed"]  # set_last_block_fixed_j_range: union with the block's own (blocks.rs:556-563)
            self.block["fixed"] = (min(nf[0], old[0]), max(nf[1], old[1])) if old is not None else nf
        js, je = self.block["j_range"]
        if self.m < js or self.m > je:
            return None
        return self.index(self.m)

    # ---- lib.rs:122-175 + band.rs:100-141 (BandDoubling from H0, factor 2) ----
    def cost(self):
        h0 = self.h(0, 0)
        start_f, start_inc = h0, max(1, BLOCK)
        s, last_s, maxs = start_f + start_inc, -1, None
        while True:
            r = self.align_for_bounded_dist(s)
            if r is not None:
                if r <= s:
                    return r
                maxs = r if maxs is None else min(maxs, r)
            nxt = max(int(math.ceil(2.0 * (s - start_f))), 1) + start_f
            last_s, s = s, (nxt if maxs is None else min(nxt, maxs))
            if s <= last_s:
                s = max(int(math.ceil(2.0 * (last_s - start_f))), 1) + start_f


def the_pair():
    """The pair tests/tools/fuzz_sweep.py met at seed 20260927 (profiles/r02_runs/fuzz_sweep_seed20260927.log): SH k = 12,
    gen_pair(57373, 0.2, 705006267) with the soak's own long-indel edit, replayed from the soak's random stream."""
    import oracle
    from tests.test_sweep_emu import variants
    from tests.util_seq import gen_pair, rand_seq

    rng = random.Random(20260927)
    vs = variants(oracle)
    count = 0
    while True:
        rng.choice(list(vs))
        n = rng.choice([rng.randint(1, 600), rng.randint(600, 6000), rng.randint(6000, 60000)])
        e = rng.choice([0.0, 0.005, 0.02, 0.05, 0.1, 0.2, 0.4, 0.8])
        s = rng.randint(1, 10**9)
        mode = rng.random()
        a, b = gen_pair(n, e, s)
        if mode < 0.3 and n > 50:
            cut = rng.randint(0, len(b) - 1)
            ln = rng.randint(1, max(1, min(5000, len(b) // 2)))
            b = b[:cut] + b[cut + ln:] if rng.random() < 0.5 else b[:cut] + rand_seq(ln, s + 1) + b[cut:]
            b = b or b"A"
        elif mode < 0.35:
            b = rand_seq(rng.randint(1, n + 50), s + 2)
        rng.random()
        count += 1
        if s == 705006267:
            return a, b
        if count % 8 == 0:
            for _ in range(12):
                rng.choice([rng.randint(1, 700), rng.randint(700, 9000)])
                rng.choice([0.0, 0.01, 0.05, 0.12, 0.2, 0.35])
                rng.randint(1, 10**9)


if __name__ == "__main__":
    import time

    import oracle

    a, b = the_pair()
    print(f"pair: |a| = {len(a)}, |b| = {len(b)}, edit distance (plain DP oracle) = {oracle.nw_cost(a, b, True)}", flush=True)
    prm = oracle.make_params(domain="astar", heuristic="sh", k=12, doubling="band", start="h0", factor=2.0, block_width=256, sparse=True,
                             incremental_doubling=False, dt_trace=True, max_g=40, fr_drop=10, sparse_h=True)
    c_tr = oracle.cpu_align(a, b, prm, trace=True)
    c_co = oracle.cpu_align(a, b, prm, trace=False)
    print(f"engine.hpp over the CPU kernels: traced {c_tr[0]}, cost-only {c_co[0]} (f_max_tries {c_co[2]['f_max_tries']}, computed_lanes {c_co[2]['computed_lanes']})", flush=True)
    t = time.time()
    st = CostOnly(a, b, "sh", 12, True)
    got = st.cost()
    print(f"standalone restatement of the reference's cost-only mode: {got} (f_max_tries {st.f_max_tries}, computed_lanes {st.computed_lanes}, {time.time() - t:.0f} s)")
