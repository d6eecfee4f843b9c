// sweep_logic.hpp -- the scalar band logic of the device-side A*PA2 sweep, shared by host and device.
//
// The sweep kernel (sweep_wave.hpp) runs one `align_for_bounded_dist` pass (astarpa2/src/domain.rs:356-541) as ONE
// persistent launch: every 256-column block's `j_range` (domain.rs:117-246) and `fixed_j_range` (domain.rs:251-350) is
// decided by the wavefronts that hold the block's right-edge column, so no host round trip per block is left.  The
// functions here are those decisions for the heuristics whose h() is closed-form or a per-column table (NoCost, GapCost,
// SH); they are plain integer code, compiled both by hipcc (device) and by g++ (host set-up of block 0/1, the CPU
// emulation of the kernel under tests/tools, and unit tests against engine.hpp's literal loops).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PA_HD __host__ __device__ __forceinline__
#else
#define PA_HD inline
#endif

namespace pa {
namespace sweep {

constexpr int32_t kBlockW = 256;      // AstarPa2Params::block_width of every preset (params.rs:46-128); the sweep requires it
constexpr int32_t kLanes = 64;        // wavefront lanes
constexpr int32_t kLaneRows = 32;     // DP rows per lane (K = 1 subword)
constexpr int32_t kStripRows = 2048;  // rows per strip = one wavefront
constexpr int32_t kNone = INT32_MIN;  // "no value" for optional fields
constexpr int32_t kMaxShrink = 1024;  // rows the band's bottom edge may move UP per block before the pass is handed back to the
                                      // host engine (strips above the bottom run ahead of the decision under this bound)

enum : int32_t { kHeurNone = 0, kHeurGap = 1, kHeurSH = 2 };

// Result / abort codes of one pass.
enum : uint32_t {
    kStRunning = 0,
    kStDone = 1,        // reached the last block: `value` = last_block.get(|b|)               (domain.rs:520)
    kStNoPath = 2,      // empty j_range / fixed_j_range, or |b| outside the last block          (domain.rs:411,439,483,522)
    kStAbort = 3,       // the kernel met a case it does not handle: the host engine redoes the pass (`value` = reason)
    kStTimeout = 4,     // a bounded spin expired
};
enum : int32_t {
    kAbortNonMonotone = 1,   // the band's bottom edge moved up between blocks (speculation rule violated)
    kAbortOldAbove = 2,      // an older pass's range starts above this pass's fixed start
    kAbortWindow = 3,        // band outside the diagonal window the buffers were sized for
    kAbortRing = 4,          // a ring slot was overwritten before its reader got there
    kAbortMismatch = 5,      // top-edge and bottom-edge logic disagree (internal consistency check)
    kAbortScanAbove = 6,     // a bottom-up scan left the stored columns
};

struct HeurParams {
    int32_t kind;         // kHeurNone / kHeurGap / kHeurSH
    int32_t n, m;         // |a|, |b|
    const int32_t* sh_h;  // kHeurSH: h(i, *) for i = 0..n (engine.hpp SeedHeuristicH::h_by_i)
};

// ---- the block-column store of the batched A*PA2 kernels: band-proportional slots --------------------------------------------
// The reference keeps, per block, the V words of the block's rows only (astarpa2/src/block.rs:8-21, blocks.rs:280-340).  The batch
// kernels keep slot k of a pair as a WINDOW of `win` words around the main diagonal -- absolute words [off, off + win), off from the
// block's column by a fixed-point slope -- so that a pair costs (blocks x win) words instead of (blocks x all rows); words are still
// addressed by their absolute index (the slot pointer is moved back by `off`).  A block whose rows leave the window makes the pair
// run again with full-height slots (pa_batch_align's second round).  win >= words of b: off = 0, the full column.
struct SlotGeom {
    int32_t n, m;      // |a|, |b|
    int32_t win;       // words per slot
    uint32_t ratio;    // floor(m * 2^20 / n): rows per column of the main diagonal
};
PA_HD int32_t slot_off(const SlotGeom& g, int32_t k) {
    const int32_t wtot = (g.m + 63) >> 6;
    if (g.win >= wtot) return 0;
    const int64_t col = (int64_t)k * 256 < (int64_t)g.n ? (int64_t)k * 256 : (int64_t)g.n;
    const int32_t dw = (int32_t)(((uint64_t)col * (uint64_t)g.ratio) >> 26);  // row of the diagonal / 64
    int32_t off = dw - g.win / 2;
    if (off > wtot - g.win) off = wtot - g.win;
    if (off < 0) off = 0;
    return off;
}
PA_HD bool slot_holds(const SlotGeom& g, int32_t k, int32_t w0, int32_t w1) {  // words [w0, w1) inside slot k's window
    const int32_t off = slot_off(g, k);
    return w0 >= off && w1 <= off + g.win;
}

PA_HD int32_t iabs32(int32_t x) { return x < 0 ? -x : x; }
PA_HD int32_t imin32(int32_t a, int32_t b) { return a < b ? a : b; }
PA_HD int32_t imax32(int32_t a, int32_t b) { return a > b ? a : b; }
PA_HD int32_t div_ceil_pos(int32_t a, int32_t b) { return (a + b - 1) / b; }
PA_HD int32_t floor64(int32_t x) { return x >= 0 ? (x & ~63) : -(((-x) + 63) & ~63); }
PA_HD int32_t ceil64(int32_t x) { return x >= 0 ? ((x + 63) & ~63) : -((-x) & ~63); }

PA_HD int32_t heur_h(const HeurParams& hp, int32_t i, int32_t j) {  // pa-heuristic distances.rs:131-168, sh.rs:88-106
    if (hp.kind == kHeurGap) return iabs32((hp.n - i) - (hp.m - j));
    if (hp.kind == kHeurSH) return hp.sh_h[i];
    return 0;
}

// f(v) of j_range (domain.rs:153-158): gu + extend_cost(u, v) + h(v), unit costs (extend_cost = gap cost).
PA_HD int32_t jr_f(const HeurParams& hp, int32_t gu, int32_t u0, int32_t u1, int32_t x, int32_t y) {
    return gu + iabs32((x - u0) - (y - u1)) + heur_h(hp, x, y);
}

// End of the next block's j_range for Domain::Astar (domain.rs:160-235), given u = (is, fixed_end) and g(u).
// `sparse_h` selects the reference's two probing schedules.  For fixed x, f is non-decreasing in y below the diagonal of
// u for all three heuristics, so runs of the reference's "+8 while f <= f_max" / "+1 while f <= f_max" steps are
// fast-forwarded by a galloping search to the same stopping point (checked against the literal loops in
// tests/test_sweep_logic.py); everything else is the literal control flow.
PA_HD int32_t jr_end_astar(const HeurParams& hp, int32_t is, int32_t ie, int32_t fixed_end, int32_t gu, int32_t f_max,
                           int32_t sparse_h) {
    const int32_t u0 = is, u1 = fixed_end, blen = hp.m;
    int32_t v0 = u0, v1 = u1;
    if (!sparse_h) {  // domain.rs:171-181
        while (v0 < ie) {
            v0 += 1;
            v1 += 2;
            // while v1 <= blen && f(v0, v1) <= f_max: v1 += 1   -- largest t with f(v1 + t - 1) <= f_max, v1 + t - 1 <= blen
            if (v1 <= blen && jr_f(hp, gu, u0, u1, v0, v1) <= f_max) {
                int32_t lo = 0, hi = 1;  // f(v1 + lo) ok; find first not ok (or beyond blen)
                while (v1 + hi <= blen && jr_f(hp, gu, u0, u1, v0, v1 + hi) <= f_max) {
                    lo = hi;
                    hi = hi * 2;
                }
                // invariant: v1 + lo ok, v1 + hi not ok (or > blen)
                while (hi - lo > 1) {
                    const int32_t mid = lo + (hi - lo) / 2;
                    if (v1 + mid <= blen && jr_f(hp, gu, u0, u1, v0, v1 + mid) <= f_max) lo = mid;
                    else hi = mid;
                }
                v1 += lo + 1;
            }
            v1 -= 1;
        }
        return v1;
    }
    // domain.rs:182-233
    v0 += 1;
    v1 += 1;
    v1 += kBlockW;
    v1 = imin32(v1, blen);
    for (;;) {
        if (v1 < v0 - u0 + u1) {
            v1 = v0 - u0 + u1;
            break;
        }
        const int32_t fv = jr_f(hp, gu, u0, u1, v0, v1);
        if (fv <= f_max) {
            if (v1 == blen) break;
            // literal: v1 += 8 (clamped to blen), loop.  Fast-forward over the steps that stay <= f_max.
            // positions p_t = min(v1 + 8t, blen); find the largest t >= 1 such that all p_1..p_{t-1} are ok, i.e. the
            // first t >= 1 with p_t not ok or p_t == blen (the loop then handles p_t itself).
            int32_t lo = 0, hi = 1;  // p_lo ok and != blen
            // (32-bit: v1 <= blen < 2^30 and the search stops at the first p >= blen, so p < 2 * blen + 8)
            for (;;) {
                const int32_t p = v1 + 8 * hi;
                if (p >= blen || jr_f(hp, gu, u0, u1, v0, p) > f_max) break;
                lo = hi;
                hi *= 2;
            }
            while (hi - lo > 1) {
                const int32_t mid = lo + (hi - lo) / 2;
                const int32_t p = v1 + 8 * mid;
                if (p < blen && jr_f(hp, gu, u0, u1, v0, p) <= f_max) lo = mid;
                else hi = mid;
            }
            // p_hi is the first position that is not ok or reaches blen: the literal loop arrives there next.
            const int32_t p = v1 + 8 * hi;
            v1 = p >= blen ? blen : p;
        } else {
            v0 += div_ceil_pos(fv - f_max, 2);
            if (v0 > ie) {
                v0 = ie;
                break;
            }
        }
    }
    v0 = ie;
    for (;;) {
        if (v1 < v0 - u0 + u1) {
            v1 = v0 - u0 + u1;
            break;
        }
        const int32_t fv = jr_f(hp, gu, u0, u1, v0, v1);
        if (fv <= f_max) break;
        v1 -= div_ceil_pos(fv - f_max, 2);
    }
    return v1;
}

// One persistent record per 256-column block, carried from pass to pass (what the reference keeps in Blocks::blocks[k]:
// j_range, original_j_range, fixed_j_range; blocks.rs:86-108, block.rs:8-45).  Index 0 is the first column (i = 0).
struct BlockRec {
    int32_t js, je;      // j_range rounded out to 64 (kNone: the block does not exist yet)
    int32_t ojs, oje;    // original_j_range
    int32_t fs, fe;      // fixed_j_range (kNone: not set)
    int32_t top_val, bot_val;
};

// The j_range of a block from its predecessor's fixed range (domain.rs:117-246 for Domain::Astar), including the union
// with the older pass's range and the crop to [0, |b|].  Returns false when the range is empty.
struct JRangeOut {
    int32_t ojs, oje;  // original (unrounded) range
    int32_t js, je;    // rounded out
};
PA_HD bool next_j_range(const HeurParams& hp, int32_t is, int32_t ie, int32_t prev_fs, int32_t prev_fe, int32_t gu, int32_t f_max,
                        int32_t sparse_h, int32_t old_js, int32_t old_je, JRangeOut* out) {
    int32_t s = prev_fs;
    int32_t e = jr_end_astar(hp, is, ie, prev_fe, gu, f_max, sparse_h);
    if (old_js != kNone) {  // range.union(old_range)
        s = imin32(s, old_js);
        e = imax32(e, old_je);
    }
    s = imax32(s, 0);  // intersection with [0, |b|]
    e = imin32(e, hp.m);
    out->ojs = s;
    out->oje = e;
    out->js = floor64(s);
    out->je = ceil64(e);
    return s <= e;
}

// The decision made after block k is complete (domain.rs:432-455, blocks.rs:205-230): the next block's range, whether
// the block of an older pass is reused, and what the pass statistics gain.  Shared by the kernel's bottom-edge logic and
// by the host, which decides block 1 from the first column.
struct NextDecision {
    bool ok;           // false: empty range, the pass ends (domain.rs:439-443)
    JRangeOut jr;
    int32_t flags;     // bit0: reused; bit1: every block so far reused
    uint64_t d_num_blocks, d_unique_add, d_unique_sub, d_computed, d_incremental;
};
PA_HD NextDecision decide_next(const HeurParams& hp, int32_t f_max, int32_t sparse_h, int32_t is, int32_t ie_next, int32_t fs, int32_t fe,
                               int32_t gu, const BlockRec& old_next, bool all_reused) {
    NextDecision d;
    d.flags = 0;
    d.d_num_blocks = d.d_unique_add = d.d_unique_sub = d.d_computed = d.d_incremental = 0;
    d.ok = next_j_range(hp, is, ie_next, fs, fe, gu, f_max, sparse_h, old_next.js, old_next.je, &d.jr);
    if (!d.ok) return d;
    // `blocks.next_block_j_range() == Some(j_range)`: the unrounded new range against the rounded old one (domain.rs:449-455)
    const bool reuse = all_reused && old_next.js != kNone && old_next.js == d.jr.ojs && old_next.je == d.jr.oje;
    if (reuse) {
        d.jr.ojs = old_next.ojs;  // a reused block keeps its old original_j_range (blocks.rs:190-197)
        d.jr.oje = old_next.oje;
        d.flags = 3;
    } else {  // blocks.rs:211-226, 704-712
        d.d_num_blocks = 1;
        d.d_unique_add = (uint64_t)((d.jr.je - d.jr.js) / 64);
        if (old_next.js != kNone) d.d_unique_sub = (uint64_t)((old_next.je - old_next.js) / 64);
        if (ie_next - is > 1) {
            d.d_computed = (uint64_t)((d.jr.je - d.jr.js) / 64);
            d.d_incremental = 1;
        }
    }
    return d;
}

// ---- self-validating 8-byte words ("the data is the flag"): {tag:32 | value:32} ------------------------------------
PA_HD uint64_t tw_make(uint32_t tag, int32_t value) { return ((uint64_t)tag << 32) | (uint32_t)value; }
PA_HD uint32_t tw_tag(uint64_t w) { return (uint32_t)(w >> 32); }
PA_HD int32_t tw_val(uint64_t w) { return (int32_t)(uint32_t)w; }
// tag of per-block words: pass id (12 bits, never 0) and block index (20 bits)
PA_HD uint32_t blk_tag(uint32_t pass, int32_t k) { return ((pass & 0xFFFu) << 20) | ((uint32_t)k & 0xFFFFFu); }

// Records in device memory, all fields tagged words.
struct TRec {  // top-edge record of block k, written by the top-edge logic of block k-1
    uint64_t state;    // kTDesc / kTCont / kTEmpty
    uint64_t js;       // DESC: rounded start of block k's j_range
    uint64_t top_val;  // DESC: block k's top_val = index_{k-1}(js) + width_k
    uint64_t fs_prev;  // DESC: final fixed start of block k-1
    uint64_t lim;      // DESC: lower limit for the bottom-up scan of block k-1 (the row the top-down scan reached)
    uint64_t found;    // DESC: 1 if the top-down scan of block k-1 found a row with f <= f_max itself
    uint64_t cont_j;   // CONT: row at which the strip below continues the scan of block k-1
    uint64_t pad;
};
enum : int32_t { kTDesc = 1, kTCont = 2, kTEmpty = 3 };

struct BRec {  // bottom-edge record of block k
    uint64_t js, je, ojs, oje;  // written when block k's range is decided (by the bottom-edge logic of block k-1)
    uint64_t flags;             // same time: bit0 = reused block, bit1 = all blocks so far reused
    uint64_t fs, fe, bot_val;   // written when block k itself is complete
    uint64_t top_val;           // idem (copied from the top-edge record)
    uint64_t smax;              // with the range: highest strip started so far in this pass
    uint64_t specmax;           // with the range: rows below this may have been treated as inside the band by strips running ahead
    uint64_t pad[5];
};
static_assert(sizeof(TRec) == 64 && sizeof(BRec) == 128, "record layout");

struct PassStats {  // BlockStats of one pass (blocks.rs:76-84), accumulated by the bottom-edge logic
    uint64_t num_blocks, num_incremental_blocks, computed_lanes, unique_lanes;
};

struct Status {
    uint32_t state;     // kSt*
    int32_t value;      // cost / abort reason
    int32_t k_end;      // last block whose range was committed
    int32_t k_fixed;    // last block whose fixed range was committed
    PassStats stats;
    uint32_t dbg[8];
};

}  // namespace sweep
}  // namespace pa
