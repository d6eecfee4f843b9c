// apa2_logic.hpp -- A*PA2's whole band search for ONE pair as one program: `AstarPa2::cost_or_align` (astarpa2/src/lib.rs:122-175),
// the search loops of band.rs:100-182 and every `align_for_bounded_dist` pass (domain.rs:356-541) with sparse, non-incremental
// 256-column blocks (blocks.rs:205-340) under Domain::Astar with NoCost / GapCost / SH -- the `simple` preset and its relatives.
//
// What it is for: the many-pair mode.  The reference aligns pairs one after another on one core (pa-bin/src/main.rs:24-35); here
// ONE WAVEFRONT runs this program for one pair, start to finish, with no host round trip: thousands of pairs are in flight on
// the chip and a pair whose pass fails simply goes on with the next bound.  Nothing is shared between wavefronts, so nothing
// polls and nothing can dead-lock (unlike the single-pair sweep of sweep_wave.hpp, which spreads ONE pair over many
// wavefronts and has to speculate).  Everything the program decides with is wavefront-uniform.
//
// The program is written once against a small backend B (the block columns and the DP on them):
//   * apa2_kernel.hpp: the gfx950 backend -- the Myers strips of strip_kernel.hpp, wave-parallel prefix sums and scans;
//   * oracle/apa2_emu.cpp (tests only): the CPU oracle's kernels and the reference's literal scan loops.
// What a backend provides (all results uniform):
//   BlockRec load_rec(k); void store_rec(k, rec)               the persistent record of block k (Blocks::blocks[k] minus v)
//   int32_t index(k, rec, j)                                    Block::index (block.rs:69-122), k = 0: the first column
//   int32_t compute(k, prev, cur, i0, i1)                       init_v_with_overlap (blocks.rs:753-767) + compute_block with
//                                                               HMode::None (blocks.rs:686-748): column k <- DP; the bottom-row sum
//   bool scan_first(k, rec, i, f_max, lo, hi, &j)               first row j in [lo, hi] with index(j) + h(i, j) <= f_max
//   bool scan_last(k, rec, i, f_max, lo, hi, &j)                last such row
//   bool failed()                                               the backend met an error (device: the strips' error word)
//   int32_t uniform(x)                                          x, known to be wavefront-uniform (device: back into a scalar register)
//   uint64_t strip_instructions()                               reporting: modelled VALU instructions of the DP strips so far
//   void mark(slot, value)                                      diagnostics: progress markers (a no-op unless a debug buffer is set)
// The two scans replace the probing loops of domain.rs:306-328: with `sparse_h` the reference jumps ceil((f - f_max) / 2) rows
// after a failed probe; f changes by at most 2 per row (g by 1, the heuristic by at most 1), so a jump never skips a row with
// f <= f_max, and both loops end exactly on the first / last such row of the interval (the emulation backend runs the literal
// loops, so the CPU tests check this equivalence on every block).
#pragma once
#include <math.h>

#include "sweep_logic.hpp"

namespace pa {
namespace apa2 {

using sweep::BlockRec;
using sweep::HeurParams;
using sweep::JRangeOut;
using sweep::NextDecision;
using sweep::kBlockW;
using sweep::kNone;

enum : int32_t { kDoublingBand = 1, kDoublingLinear = 2 };             // engine.hpp DoublingKind
enum : int32_t { kStartZero = 0, kStartGap = 1, kStartH0 = 2 };       // engine.hpp DoublingStart

// Uniform parameters of a launch (AstarPa2Params as far as this path reads them).
struct SearchParams {
    int32_t heur;      // sweep::kHeurNone / kHeurGap / kHeurSH
    int32_t sparse_h;
    int32_t doubling;  // kDoublingBand / kDoublingLinear
    int32_t start;     // kStart*
    float factor;      // BandDoubling
    int32_t delta;     // LinearSearch
};

// Why a pair was handed back to the host engine (the reference would panic there, or the program met a case it leaves alone).
enum : int32_t {
    kOk = 0,
    kErrIndexBelow = 1,      // "Cannot index block below its range" (block.rs:70)
    kErrRangeOrder = 2,      // block.j_range.0 <= prev_fixed.0 (domain.rs:292)
    kErrEmptyWithNext = 3,   // empty j_range with an existing next block (domain.rs:440)
    kErrFixedUnset = 4,      // "With A* Domain, fixed_j_range should always be set" (domain.rs:131)
    kErrH0 = 5,              // "Heuristic at start > final cost" (lib.rs:170)
    kErrTooManyPasses = 6,   // the search did not end within kMaxPasses passes
    kErrDevice = 7,          // a backend failure (strip error word)
    kErrDegenerate = 8,      // |a| == 0 or |b| == 0
    kErrWindow = 9,          // a block's rows left the pair's window of the column store: the pair runs again with full-height slots
};
constexpr int32_t kMaxPasses = 1 << 20;
// A pass whose bound exceeds |a| + |b| covers the whole matrix and must succeed; a search that gets this far beyond it has met a
// backend failure, not a hard pair.
PA_HD bool bound_runaway(int32_t s, int32_t n, int32_t m) { return s < 0 || s > 4 * (n + m) + 8 * kBlockW; }

// Per pair, written once at the end.
struct PairResult {
    int32_t status;  // kOk / kErr*
    int32_t cost;
    int32_t f_max;   // the bound of the pass that succeeded
    uint32_t f_max_tries, sanity_violations;
    uint32_t num_blocks, num_incremental_blocks;
    uint32_t pad0;
    uint64_t computed_lanes, unique_lanes;
    int32_t last_block_idx, blocks_len;
    uint64_t strip_instr;  // VALU instructions of the DP strips this pair ran (model: steps x (11 + 12 K)); reporting only
};
static_assert(sizeof(PairResult) == 64, "PairResult layout");

template <class B>
struct PairProg {
    B& be;
    HeurParams hp;
    SearchParams sp;
    int32_t n, m, nblk;
    // Blocks state that survives a pass (blocks.rs:86-108)
    int32_t last_block_idx = 0, blocks_len = 0;
    // statistics (domain.rs:31-43, blocks.rs:76-84)
    uint32_t f_max_tries = 0, sanity = 0, num_blocks = 0, num_incremental = 0;
    uint64_t computed_lanes = 0, unique_lanes = 0;
    int32_t err = kOk;

    PA_HD PairProg(B& backend, const HeurParams& h, const SearchParams& s) : be(backend), hp(h), sp(s) {
        n = h.n;
        m = h.m;
        nblk = (n + kBlockW - 1) / kBlockW;
    }

    PA_HD static int32_t blk_i1(int32_t k, int32_t n_) {
        const int32_t e = k * kBlockW;
        return e < n_ ? e : n_;
    }

    // One align_for_bounded_dist (domain.rs:356-541).  true: Some(dist) in *dist; false: None.  Check `err` afterwards.
    PA_HD bool pass(int32_t f_max, int32_t* dist) {
        f_max_tries += 1;
        be.mark(1, (uint32_t)f_max);
        be.mark(2, f_max_tries);
        // ---- the first column (domain.rs:395-413, blocks.rs:146-179) ----
        BlockRec stale;
        stale.js = kNone;
        stale.je = kNone;
        if (last_block_idx + 1 < blocks_len) stale = be.load_rec(last_block_idx + 1);  // next_block_j_range() before init()
        JRangeOut jr0;
        if (!sweep::next_j_range(hp, -1, 0, -1, -1, 0, f_max, sp.sparse_h, stale.js, stale.je, &jr0) || jr0.ojs > 0) return false;
        BlockRec prev;
        prev.ojs = jr0.ojs;
        prev.oje = jr0.oje;
        prev.js = 0;
        prev.je = jr0.je;
        if (blocks_len > 0) {  // initial_j_range.union(blocks[0].j_range), rounded
            const BlockRec old0 = be.load_rec(0);
            if (old0.je > prev.je) prev.je = old0.je;
        }
        prev.fs = jr0.ojs;
        prev.fe = jr0.oje;
        prev.top_val = 0;
        prev.bot_val = prev.je;
        be.store_rec(0, prev);
        if (blocks_len < 1) blocks_len = 1;
        last_block_idx = 0;

        bool all_reused = true;
        for (int32_t k = 1; k <= nblk; ++k) {
            const int32_t i0 = (k - 1) * kBlockW, i1 = blk_i1(k, n);
            if (prev.fs == kNone) {
                err = kErrFixedUnset;
                return false;
            }
            // ---- j_range of block k (domain.rs:117-246) and the reuse test (domain.rs:449-455) ----
            be.mark(3, (uint32_t)k);
            be.mark(0, 10);
            const int32_t gu = be.index(k - 1, prev, prev.fe);
            be.mark(0, 11);
            BlockRec old_next;
            old_next.js = kNone;
            old_next.je = kNone;
            old_next.fs = kNone;
            old_next.fe = kNone;
            if (k < blocks_len) old_next = be.load_rec(k);
            const NextDecision d = sweep::decide_next(hp, f_max, sp.sparse_h, i0, i1, prev.fs, prev.fe, gu, old_next, all_reused);
            be.mark(0, 12);
            if (!d.ok) {
                if (old_next.js != kNone) err = kErrEmptyWithNext;
                return false;
            }
            const bool reuse = (d.flags & 1) != 0;
            all_reused = all_reused && reuse;
            BlockRec cur;
            if (reuse) {  // blocks.rs:190-197: the block stays as the older pass left it
                cur = old_next;
            } else {      // blocks.rs:205-340
                num_blocks += (uint32_t)d.d_num_blocks;
                unique_lanes += d.d_unique_add;
                unique_lanes -= d.d_unique_sub;
                computed_lanes += d.d_computed;
                num_incremental += (uint32_t)d.d_incremental;
                cur.js = d.jr.js;
                cur.je = d.jr.je;
                cur.ojs = d.jr.ojs;
                cur.oje = d.jr.oje;
                cur.fs = old_next.js != kNone ? old_next.fs : kNone;  // fixed_j_range is kept (blocks.rs:308)
                cur.fe = old_next.js != kNone ? old_next.fe : kNone;
                if (cur.js < prev.js) {
                    err = kErrIndexBelow;
                    return false;
                }
                cur.top_val = be.index(k - 1, prev, cur.js) + (i1 - i0);
                const int32_t prev_bot = be.index(k - 1, prev, cur.je);
                if (k == blocks_len) blocks_len += 1;
                be.mark(0, 13);
                be.mark(4, (uint32_t)cur.js);
                be.mark(5, (uint32_t)cur.je);
                cur.bot_val = prev_bot + be.compute(k, prev, cur, i0, i1);
                if (be.failed()) {
                    err = kErrDevice;
                    return false;
                }
                be.store_rec(k, cur);
            }
            last_block_idx = k;
            // ---- fixed_j_range of block k (domain.rs:251-350) ----
            if (cur.js > prev.fs) {
                err = kErrRangeOrder;
                return false;
            }
            const int32_t lo = prev.fs, hi = cur.oje < m ? cur.oje : m;
            int32_t fs = 0, fe = -1;
            bool found = false;
            be.mark(0, 14);
            if (lo <= hi) found = be.scan_first(k, cur, i1, f_max, lo, hi, &fs);
            be.mark(0, 15);
            if (found) be.scan_last(k, cur, i1, f_max, fs, hi, &fe);
            be.mark(0, 16);
            if (cur.fs != kNone) {  // union with what the older passes fixed (domain.rs:332-341)
                if (!found) {
                    fs = cur.fs;
                    fe = cur.fe;
                } else {
                    fs = fs < cur.fs ? fs : cur.fs;
                    fe = fe > cur.fe ? fe : cur.fe;
                }
                found = true;
            }
            if (!found || fs > fe) return false;  // domain.rs:483-489
            cur.fs = fs;
            cur.fe = fe;
            be.store_rec(k, cur);
            prev = cur;
        }
        // ---- domain.rs:520-523 ----
        if (m < prev.js || m > prev.je) return false;
        *dist = be.index(nblk, prev, m);
        return true;
    }

    PA_HD int32_t next_bound(int32_t s, int32_t offset) const {
        if (sp.doubling == kDoublingLinear) return s + sp.delta;
        const float x = ceilf(sp.factor * (float)(s - offset));  // band.rs:138, f32 arithmetic
        const int32_t c = be.uniform((int32_t)x);               // (the float unit is a vector unit: keep the bound scalar)
        return (c > 1 ? c : 1) + offset;
    }

    // lib.rs:122-175 + band.rs:100-182 (engine.hpp cost_or_align / band_search).
    PA_HD void run(PairResult* out) {
        const int32_t h0 = sweep::heur_h(hp, 0, 0);
        int32_t start_f = 0, start_inc = 1;  // band.rs:13-23
        if (sp.start == kStartGap) {
            start_f = start_inc = sweep::iabs32(n - m);
        } else if (sp.start == kStartH0) {
            start_f = h0;
            start_inc = 1;
        }
        int32_t s, offset = start_f;
        if (sp.doubling == kDoublingLinear) {
            s = start_f;
        } else {
            if (start_inc < kBlockW) start_inc = kBlockW;  // lib.rs:142
            s = offset + start_inc;
        }
        int32_t last_s = -1, maxs = INT32_MAX, cost = 0, f_ok = 0;
        bool done = false;
        for (int32_t it = 0; it < kMaxPasses && !done && err == kOk; ++it) {
            int32_t dist = 0;
            const bool some = pass(s, &dist);
            if (err != kOk) break;
            if (some) {
                if (dist > maxs) sanity += 1;  // band.rs:118-121
                if (dist <= s) {
                    if (dist <= last_s) sanity += 1;  // band.rs:123-126
                    cost = dist;
                    f_ok = s;
                    done = true;
                    break;
                }
                if (dist < maxs) maxs = dist;
            } else if (maxs != INT32_MAX) {
                sanity += 1;  // band.rs:132-135
            }
            const int32_t before = s;
            last_s = s;
            const int32_t nx = next_bound(s, offset);
            s = nx < maxs ? nx : maxs;
            if (s <= before) s = next_bound(before, offset);  // never stall (engine.hpp band_search)
            if (bound_runaway(s, n, m)) break;
        }
        if (err == kOk && !done) err = kErrTooManyPasses;
        if (err == kOk && h0 > cost) err = kErrH0;
        out->status = err;
        out->cost = cost;
        out->f_max = f_ok;
        out->f_max_tries = f_max_tries;
        out->sanity_violations = sanity;
        out->num_blocks = num_blocks;
        out->num_incremental_blocks = num_incremental;
        out->pad0 = 0;
        out->computed_lanes = computed_lanes;
        out->unique_lanes = unique_lanes;
        out->last_block_idx = last_block_idx;
        out->blocks_len = blocks_len;
        out->strip_instr = be.strip_instructions();
    }
};

}  // namespace apa2
}  // namespace pa
