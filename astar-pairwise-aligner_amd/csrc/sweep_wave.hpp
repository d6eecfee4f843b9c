// sweep_wave.hpp -- one pass of A*PA2's align_for_bounded_dist (astarpa2/src/domain.rs:356-541) as ONE persistent launch.
//
// What it replaces: the per-block loop `for i in (0..|a|).step_by(256) { j_range; compute_next_block; fixed_j_range }`
// for the sparse, non-incremental block engine (blocks.rs:280-340) under Domain::Astar with NoCost / GapCost / SH.
//
// MI355X-first design (nothing like the CPU's block-at-a-time loop):
//  * The band is cut into ROW strips of 2048 rows aligned to absolute rows; a wavefront owns a strip for as long as the band
//    covers it and walks the columns CONTINUOUSLY across block boundaries (lane l = 32 rows, anti-diagonal skew inside the
//    wave as in strip_kernel.hpp, bottom row handed to the strip below through 8-byte granules).  Time per pass is
//    (columns x one Myers step) + pipeline depth instead of blocks x (256 + skew + launch).
//  * Band decisions are made where the data is, without stopping the pipeline:
//      - top edge: the strip holding the band's first row runs the `fixed_j_range` start scan (domain.rs:306-316) while
//        its lanes cross the block boundary one by one, and publishes the next block's (start row, top value).  A lane
//        that may become the first row of the next block forces its incoming horizontal delta to +1 (blocks.rs:730-734)
//        speculatively; lanes above the new start simply compute values nobody reads.
//      - bottom edge: the strip holding the band's last row runs the end scan (domain.rs:318-328) and the next block's
//        `j_range` (domain.rs:160-235) one block LATER than the data is available: lanes below the band compute garbage and
//        are reset to V::one() when they cross a block boundary outside the band (blocks.rs:753-767), so the decision is off
//        the critical path.  Strips between the edges never wait for any decision: they run ahead under the rule "the
//        band's bottom edge never moves up", which the bottom-edge logic verifies for every block (else the pass aborts
//        and the host engine redoes it).
//  * Everything shared between wavefronts is a self-validating 8-byte word {tag | value} written with ONE agent-scope
//    store and polled with agent-scope loads (MI355X_MICROARCH.md hand-off R2); column payloads are 8-byte write-through
//    stores whose flag (the strip's prefix word) is published after the wave's stores have drained (R1).
//
// The program is written once against a small wave policy W (vector type, cross-lane ops, memory scopes): the device
// policy maps to gfx950 builtins (sweep_kernel.hpp), the host policy emulates a wavefront with 64-element arrays and runs
// every strip on its own thread (tests/tools/sweep_emu) so that the protocol is testable without a GPU.
#pragma once
#include "sweep_logic.hpp"

#if defined(__clang__)
#define PA_NOUNROLL _Pragma("nounroll")
#else
#define PA_NOUNROLL  // (host emulation under gcc: its unroll pragma does not apply to while loops, and nothing depends on it there)
#endif

// Phase clocks (diagnostics): compiled in with -DPA_SWEEP_PHASE_TIMERS only -- eight 64-bit accumulators are 16 scalar registers
// the strip state cannot spare.
#ifdef PA_SWEEP_PHASE_TIMERS
#define PA_CLK(W) W::clock()
#define PA_CLK_ADD(acc, expr) acc += (expr)
#else
#define PA_CLK(W) 0ull
#define PA_CLK_ADD(acc, expr) do { } while (0)
#endif

#ifdef PA_SWEEP_TRACE
#include <cstdio>
#define PA_TRACE(...) std::fprintf(stderr, __VA_ARGS__)
#else
#define PA_TRACE(...) do { } while (0)
#endif

namespace pa {
namespace sweep {

struct Ctx {
    // the pair
    const uint32_t* a_codes;  // 2-bit codes, 16 per u32, padded by >= 8 words
    const uint32_t* b_prof;   // BitProfile words of b (u32 view of (nb0:u64, nb1:u64))
    int32_t n, m, nblk;       // |a|, |b|, number of 256-column blocks
    int32_t wtot;             // ceil(m / 64)
    // the pass
    int32_t f_max;
    uint32_t pass;            // 1..4095
    int32_t heur, sparse_h;
    const int32_t* sh_h;
    int32_t store_cols;       // 1: keep every block's column (traceback); 0: ring of col_ring blocks
    // state
    const BlockRec* d_old;    // [nblk + 2], merged result of the earlier passes (valid once the previous pass is over: *prev_done)
    // Passes of one band search run PIPELINED: the pass with the next f_max is launched while this one runs and follows one or two
    // blocks behind.  What a pass needs from its predecessor is tiny -- the fixed range of block kc and the range of block kc + 1
    // (the unions of domain.rs:181-183, 332-341; blocks.rs:190-197) -- and the predecessor's bottom-edge records ARE that, as
    // self-validating words: read them while the predecessor runs, the merged array once it is over.
    const BRec* prev_brec;    // the previous pass's records (nullptr: no previous pass, or it is over: use d_old)
    uint32_t prev_pass;       // its pass id (tags)
    const uint64_t* prev_done;  // *prev_done == prev_pass once it is over and d_old holds its merged records
    const uint64_t* cancel;   // *cancel == pass: the host gave this pass up (a pass before it succeeded)
    BRec* brec;               // [nblk + 2]
    TRec* trec;               // [nblk + 2]
    uint64_t* bprog;          // one word: {tag(pass, k) | oje_k} of the last decided block
    uint64_t* strip_start;    // [nstrips]: {pass | first block}
    uint64_t* pring;          // [nstrips][pr_stride]: {tag(pass, k) | index_k(end of strip)}, blocks in the strip's window
    int64_t pr_stride;
    uint64_t* gran;           // [nstrips][gran_stride]
    int64_t gran_stride;      // granules per strip boundary
    int32_t win;              // rows: the band stays within |row - column| <= win (window of the buffers)
    uint64_t* col;            // V words: [slot][col_stride] x (p, m)
    int64_t col_stride;       // V words per block column
    int32_t col_ring;         // blocks in the column ring (store_cols == 0), power of two
    Status* status;
    uint32_t* ticket;
    int32_t nstrips, nwaves;
    uint32_t spin_limit;      // polls before a wait gives up
    uint64_t* timing;         // nullptr, or 8 accumulators of phase clocks (diagnostics)
};

// NOTE on integer widths: everything a wavefront decides with is uniform and must stay on the scalar unit.  gfx950's SALU has no
// ordered 64-bit compare, so one int64 `<` sends the value (and everything computed from it: the chunk counter, the block
// index, every branch on them) to the VALU, where uniform branches turn into EXEC-mask code.  All comparisons here are
// 32-bit; lengths are < 2^30 (sweep_supported), so k * 256, row and column numbers fit.
PA_HD int32_t blk_end(const Ctx& c, int32_t k) {  // E_k: one past the last column of block k (block 0 = column 0)
    const int32_t e = k * kBlockW;
    return e < c.n ? e : c.n;
}
PA_HD int32_t col_base_word(const Ctx& c, int32_t k) {  // first V word of block k's column slot
    const int32_t lo = (k - 1) * kBlockW - c.win;
    return lo <= 0 ? 0 : (lo >> 6);
}
PA_HD int64_t col_slot(const Ctx& c, int32_t k) { return c.store_cols ? (int64_t)k : (int64_t)(k & (c.col_ring - 1)); }
PA_HD uint64_t* pr_word(const Ctx& c, int32_t r, int32_t k) {  // prefix word of strip r for block k (nullptr: outside the window)
    const int32_t lo = (r * kStripRows - c.win) / kBlockW;
    const int32_t idx = k - (lo > 0 ? lo : 0);
    if (idx < 0 || idx >= (int32_t)c.pr_stride) return nullptr;
    return c.pring + (int64_t)r * c.pr_stride + idx;
}
PA_HD int32_t gran_base(const Ctx& c, int32_t r) {  // first granule (32-column chunk) of the boundary below strip r
    const int32_t lo = (r + 1) * kStripRows - c.win;
    return lo <= 0 ? 0 : (lo >> 5);
}

// ----------------------------------------------------------------------------------------------------------------------
// Flags of the strip state are full ints: byte-sized members next to ints get merged into overlapping wider loads, which
// keeps that slice of the state in scratch memory -- and whatever is loaded from scratch counts as divergent.
typedef int32_t flag_t;

template <class W>
struct StripProg {
    using vec = typename W::vec;
    const Ctx& c;
    HeurParams hp;
    int32_t r, row0, rowE;
    // lane state
    vec lane, lrow0, vp, vm, nb0, nb1, X, acc_lo, acc_hi, snap_p, snap_m, andm, orm, basev, fpend;
    // block state
    int32_t kc;                          // the block lane 0 is in
    int32_t js_c, top_c, fsprev_c;       // top-edge record of block kc
    int32_t je_c, oje_c;                 // bottom-edge record of block kc (valid at its crossing unless bot_interior)
    flag_t bot_interior;
    int32_t js_n, top_n, fs_n, lim_n, found_n;  // top-edge record of block kc + 1
    flag_t have_n;
    flag_t is_top;                         // the band's first row is (or was) inside this strip: FORCE variant
    flag_t gran_on;                      // lane 0 takes its horizontal deltas from the strip above
    // top-down scan of block kc
    flag_t sc_active;
    int32_t sc_j, sc_e, sc_base0;
    int32_t old_fs_c, old_js_n;          // older passes: fixed start of block kc, range start of block kc + 1
    flag_t alive;                        // false: stop (pass over, abort, timeout)
    // deferred flag publications (after the payload stores have drained)
    uint64_t st_num_blocks = 0, st_unique = 0, st_computed = 0, st_incremental = 0;  // my share of the pass's BlockStats
    uint64_t t_cross2 = 0, t_probe = 0, t_flush = 0;
    uint64_t t_begin = 0, t_cross = 0, t_end = 0, t_bottom = 0, t_plain = 0, t_wait_gran = 0;  // phase clocks (W::clock ticks)
    int32_t pend_k, pend_p, pend_cont_j;  // deferred publications (see flush_deferred)
    flag_t pend_cont;

    PA_HD StripProg(const Ctx& ctx) : c(ctx) {
        hp.kind = c.heur;
        hp.n = c.n;
        hp.m = c.m;
        hp.sh_h = c.sh_h;
        pend_k = -1;
        pend_p = pend_cont_j = 0;
        pend_cont = false;
        fin_state = fin_value = fin_k_end = fin_k_fixed = 0;
        alive = true;
    }

    // ---- status ------------------------------------------------------------------------------------------------------
    PA_HD bool pass_over() { return W::load_u32(&c.status->state) != kStRunning || W::load_u64(c.cancel) == (uint64_t)c.pass; }
    // The end of the pass as this strip sees it: recorded here, written to the status block once, by wave_main (this is
    // inlined at every wait and every consistency check: it must be a handful of scalar moves, the kernel has to fit the
    // instruction cache).
    int32_t fin_state, fin_value, fin_k_end, fin_k_fixed;
    PA_HD void finish(uint32_t state, int32_t value, int32_t k_end, int32_t k_fixed) {
        PA_TRACE("pass %u strip %d block %d: finish state=%u value=%d k_end=%d\n", c.pass, r, kc, state, value, k_end);
        fin_state = (int32_t)state;
        fin_value = value;
        fin_k_end = k_end;
        fin_k_fixed = k_fixed;
        alive = false;
    }
    PA_HD void commit_finish() {
        if (fin_state != 0 && W::cas_u32(&c.status->state, kStRunning, (uint32_t)fin_state)) {
            c.status->value = fin_value;
            c.status->k_end = fin_k_end;
            c.status->k_fixed = fin_k_fixed;
        }
    }
    PA_HD void abort_pass(int32_t reason) { finish(kStAbort, reason, 0, 0); }

    // Poll a tagged word until its tag matches.  Returns false when the pass is over / timed out.
    PA_HD bool wait_word(const uint64_t* p, uint32_t tag, int32_t* out, bool no_timeout = false) {
        uint32_t spins = 0;
        for (;;) {
            const uint64_t w = W::load_u64(p);
            if (tw_tag(w) == tag) {
                *out = tw_val(w);
                return true;
            }
            W::nap(spins);
            if ((++spins & 63u) == 0) {
                if (pass_over()) {
                    alive = false;
                    return false;
                }
                if (spins == c.spin_limit / 2)
                    PA_TRACE("pass %u strip %d block %d: wait_word slow: tag want %x have %llx  off trec %ld brec %ld pring %ld start %ld\n", c.pass, r, kc, tag,
                             (unsigned long long)W::load_u64(p), (long)((const char*)p - (const char*)c.trec), (long)((const char*)p - (const char*)c.brec),
                             (long)((const char*)p - (const char*)c.pring), (long)((const char*)p - (const char*)c.strip_start));
                if (spins > c.spin_limit && !no_timeout) {
                    PA_TRACE("pass %u strip %d block %d: wait_word timeout: tag want %x have %llx  off trec %ld brec %ld pring %ld start %ld\n", c.pass, r, kc, tag,
                             (unsigned long long)W::load_u64(p), (long)((const char*)p - (const char*)c.trec), (long)((const char*)p - (const char*)c.brec),
                             (long)((const char*)p - (const char*)c.pring), (long)((const char*)p - (const char*)c.strip_start));
                    finish(kStTimeout, 0, 0, 0);
                    return false;
                }
            }
        }
    }
    PA_HD uint32_t btag(int32_t k) const { return blk_tag(c.pass, k); }
    PA_HD bool wait_pr(int32_t rr, int32_t k, int32_t* out) {  // strip rr's prefix word of block k
        const uint64_t* pw = pr_word(c, rr, k);
        if (!pw) {
            abort_pass(kAbortWindow);
            return false;
        }
        return wait_word(pw, btag(k), out);
    }

    // ---- helpers over the crossed column (snapshots) -------------------------------------------------------------------
    PA_HD static int32_t popdiff(uint32_t p, uint32_t mm) { return (int32_t)W::popc(p) - (int32_t)W::popc(mm); }
    PA_HD static int32_t prefix_of(uint32_t p, uint32_t mm, int32_t rows) {  // sum of the first `rows` (0..32) deltas
        const uint32_t mask = rows >= 32 ? 0xFFFFFFFFu : ((1u << rows) - 1u);
        return popdiff(p & mask, mm & mask);
    }
    PA_HD int32_t lane_of(int32_t j) const { return (j - row0) >> 5; }

    // index_kc(j) (block.rs:69-122) for a row inside this strip, all lanes crossed, basev valid.
    PA_HD int32_t index_local(int32_t j, int32_t p_end) {
        if (j >= rowE) return p_end + (j - rowE);  // only used with j == rowE, or beyond the band (+1 per row)
        const int32_t l = lane_of(j);
        return W::readlane_i(basev, l) + prefix_of(W::readlane(snap_p, l), W::readlane(snap_m, l), j - (row0 + 32 * l));
    }

    // index_kc(j) for a row ABOVE this strip, from the stored columns: strip rr's prefix word (or the block's top value)
    // plus the V words between.  One vector load of <= 32 words.
    PA_HD bool index_above(int32_t k, int32_t js_k, int32_t top_k, int32_t j, int32_t* out) {
        const int32_t rr = j / kStripRows;
        const int32_t rr0 = rr * kStripRows;
        int32_t base, from;
        if (js_k >= rr0) {
            base = top_k;
            from = js_k;
        } else {
            if (!wait_pr(rr - 1, k, &base)) return false;
            from = rr0;
        }
        if (j < from) {
            abort_pass(kAbortScanAbove);
            return false;
        }
        // strip rr must have stored block k's column: its own prefix word says so
        int32_t dummy;
        if (!wait_pr(rr, k, &dummy)) return false;
        const int32_t w0 = from >> 6, w1 = j >> 6;  // whole words [w0, w1), then j & 63 rows of word w1
        const uint64_t* colk = c.col + (col_slot(c, k) * c.col_stride - col_base_word(c, k)) * 2;
        // lane l handles word w0 + l (l < 33)
        const vec widx = lane + (uint32_t)w0;
        const typename W::mask inr = W::le_u(widx, (uint32_t)w1);
        vec plo, phi, mlo, mhi;
        W::load_v_words(colk, widx, inr, plo, phi, mlo, mhi);
        // rows of this word that count: all 64 for w < w1, j & 63 for w == w1
        const typename W::mask lastw = W::eq_u(widx, (uint32_t)w1);
        const int32_t rem = j & 63;
        const uint32_t mlo_mask = rem >= 32 ? 0xFFFFFFFFu : ((1u << rem) - 1u);
        const uint32_t mhi_mask = rem <= 32 ? 0u : ((1u << (rem - 32)) - 1u);
        const vec klo = W::select(lastw, W::splat(mlo_mask), W::splat(0xFFFFFFFFu));
        const vec khi = W::select(lastw, W::splat(mhi_mask), W::splat(0xFFFFFFFFu));
        vec val = W::popc_v(plo & klo) + W::popc_v(phi & khi) - W::popc_v(mlo & klo) - W::popc_v(mhi & khi);
        val = W::select(inr, val, W::splat(0u));
        *out = base + (int32_t)W::reduce_add(val);
        return true;
    }

    // ---- start of a strip ---------------------------------------------------------------------------------------------
    PA_HD bool fetch_trec_desc(int32_t k) {  // wait for block k's top-edge record (DESC or CONT); fills the *_n fields
        const TRec* t = c.trec + k;
        int32_t st;
        if (!wait_word(&t->state, btag(k), &st)) return false;
        if (st == kTEmpty) {  // the pass ends at block k - 1; whoever found out has set the status
            alive = false;
            return false;
        }
        if (st == kTCont) {
            int32_t cj, base;
            if (!wait_word(&t->cont_j, btag(k), &cj)) return false;
            // the strip above has published its prefix word for block k - 1 before the CONT record
            if (!wait_pr(r - 1, k - 1, &base)) return false;
            sc_active = true;
            sc_j = cj;
            sc_e = 0;
            sc_base0 = base;
            have_n = false;
            is_top = true;
            return true;
        }
        // (locals, not &member: a member whose address is taken stays in scratch, and scratch loads count as divergent)
        int32_t a_js, a_top, a_fs, a_lim, a_found;
        if (!wait_word(&t->js, btag(k), &a_js) || !wait_word(&t->top_val, btag(k), &a_top) || !wait_word(&t->fs_prev, btag(k), &a_fs) ||
            !wait_word(&t->lim, btag(k), &a_lim) || !wait_word(&t->found, btag(k), &a_found))
            return false;
        js_n = a_js;
        top_n = a_top;
        fs_n = a_fs;
        lim_n = a_lim;
        found_n = a_found;
        have_n = true;
        return true;
    }

    PA_HD void publish_trec_desc(int32_t k) {
        TRec* t = c.trec + k;
        W::store_u64(&t->js, tw_make(btag(k), js_n));
        W::store_u64(&t->top_val, tw_make(btag(k), top_n));
        W::store_u64(&t->fs_prev, tw_make(btag(k), fs_n));
        W::store_u64(&t->lim, tw_make(btag(k), lim_n));
        W::store_u64(&t->found, tw_make(btag(k), found_n));
        W::store_u64(&t->state, tw_make(btag(k), kTDesc));
    }

    // ---- the top-down scan (domain.rs:306-316) ------------------------------------------------------------------------
    // Stop: the scan has its answer.  e1 = lane holding js_n, base_e1 = index_kc(first row of that lane).
    PA_HD void scan_finish(int32_t fs_final, int32_t found, int32_t lim, int32_t base_at_js) {
        fs_n = fs_final;
        found_n = found;
        lim_n = lim;
        js_n = floor64(fs_final);
        if (old_js_n != kNone && old_js_n < js_n) {  // an older pass started this block higher up: not handled here
            abort_pass(kAbortOldAbove);
            return;
        }
        top_n = base_at_js + (blk_end(c, kc + 1) - blk_end(c, kc));
        have_n = true;
        sc_active = false;
        publish_trec_desc(kc + 1);
    }
    PA_HD void scan_empty() {  // no row of block kc has f <= f_max and no older fixed range exists (domain.rs:483-489)
        sc_active = false;
        W::store_u64(&c.trec[kc + 1].state, tw_make(btag(kc + 1), kTEmpty));
        finish(kStNoPath, 0, kc, kc - 1);
    }
    // The scan probes rows of lane `cl`, which is leaving block kc in this very step: its V is the block's final one, the lanes
    // above it have left already (their final V is in the snapshot registers).  Called at the few steps per block where the
    // scanned row's lane crosses -- everything else in a crossing chunk stays on the unrolled step code.
    PA_HD void scan_probe(int32_t cl) {
        // index_kc(first row of lane cl) = value at the scan's first lane + the column sums in between
        const typename W::mask between = W::and_m(W::ge_i(lane, sc_e), W::lt_i(lane, cl));
        const int32_t base = sc_base0 + (int32_t)W::reduce_add(W::select(between, W::popc_v(snap_p) - W::popc_v(snap_m), W::splat(0u)));
        const int32_t base_prev = cl > sc_e ? base - popdiff(W::readlane(snap_p, cl - 1), W::readlane(snap_m, cl - 1)) : base;
        const int32_t l0 = row0 + 32 * cl;
        const int32_t ie = blk_end(c, kc);
        while (sc_active && sc_j < l0 + 32) {
            const bool past_end = !bot_interior && sc_j > imin32(oje_c, c.m);
            if ((old_fs_c != kNone && sc_j >= old_fs_c) || (past_end && old_fs_c != kNone)) {
                // the union with the older fixed range decides (domain.rs:332-341): fs = old start
                const int32_t e1 = lane_of(floor64(old_fs_c));
                if (e1 != cl && e1 != cl - 1) {
                    abort_pass(kAbortOldAbove);
                    return;
                }
                scan_finish(old_fs_c, 0, sc_j, e1 == cl ? base : base_prev);
                return;
            }
            if (past_end) {
                scan_empty();
                return;
            }
            const int32_t f = base + prefix_of(W::readlane(vp, cl), W::readlane(vm, cl), sc_j - l0) + heur_h(hp, ie, sc_j);
            if (f <= c.f_max) {
                const int32_t e1 = lane_of(floor64(sc_j));
                scan_finish(sc_j, 1, sc_j, e1 == cl ? base : base_prev);
                return;
            }
            sc_j += c.sparse_h ? div_ceil_pos(f - c.f_max, 2) : 1;
            if (old_fs_c != kNone && sc_j > old_fs_c) sc_j = old_fs_c;
        }
        // the scan moves on to a later lane: the lanes up to it leave the block with the scan still open
        if (sc_active) force_pending(cl, lane_of(sc_j));
    }
    // Even lanes in (lo, hi] may hold the first row of block kc + 1 (its range starts at a multiple of 64): when they cross they
    // start forcing their incoming horizontal delta to +1 (blocks.rs:730-734).  Lanes that turn out to lie above the new first
    // row compute values nobody reads.
    PA_HD void force_pending(int32_t lo, int32_t hi) {
        const typename W::mask m = W::and_m(W::and_m(W::gt_i(lane, lo), W::le_i(lane, hi)), W::eq_u(lane & 1u, 0u));
        fpend = W::select(m, W::splat(1u), fpend);
    }

    // ---- one Myers step for all lanes ------------------------------------------------------------------------------
    // (always with the +1-forcing op, a no-op in lanes that do not hold the band's first row: one more VALU per step, but one
    //  copy of each unrolled chunk variant instead of two -- the kernel has to fit the instruction cache)
    PA_HD void step(uint32_t s_x, bool hi_half) { W::template myers<true>(s_x, X, vp, vm, nb0, nb1, hi_half ? acc_hi : acc_lo, andm, orm); }

    // ---- block boundary, part 1: lane 0 is about to leave block kc ------------------------------------------------------
    // Everything this boundary needs from other wavefronts is fetched with ONE round of vector loads (lane l reads word l
    // of a record): the pass status, the bottom-edge progress word, block kc's bottom-edge record, block kc + 1's top-edge
    // record and the older passes' records of blocks kc and kc + 1.  A second round happens only when something is not
    // there yet.
    vec brv_lo, brv_hi;  // BRec[kc], lane l = word l (kept for the bottom-edge logic)
    vec oldv;            // d_old[kc], d_old[kc + 1] as 16 ints
    vec pbv_lo, pbv_hi, pnv_lo, pnv_hi, pdn_lo, pdn_hi;  // the previous pass's BRec[kc], BRec[kc + 1] and its done word
    flag_t prev_over;    // the older records come from d_old (no previous pass, or it is over and merged)
    flag_t old_prev;     // this boundary's older records are the previous pass's own (pbv / pnv)
    flag_t pfo_valid;    // pbv / pnv / pdn (or oldv) were fetched a chunk ahead
    vec pfb_st, pfb_bp_lo, pfb_bp_hi, pfb_tr_lo, pfb_tr_hi;  // the same loads issued one chunk ahead of the boundary (in flight
    flag_t pfb_valid;                                        // during the chunk; the boundary only waits if something is missing)
    vec pfp_lo, pfp_hi;  // the strip above's prefix word of block kc, issued one chunk ahead of boundary_end
    flag_t pfp_valid;

    PA_HD void boundary_loads(vec& st_lo, vec& bp_lo, vec& bp_hi, vec& tr_lo, vec& tr_hi) {
        vec dummy;
        W::load_words(reinterpret_cast<const uint64_t*>(c.status) - 1, 2, st_lo, dummy);  // [cancel word | {state, value}]: adjacent
        W::load_words(c.bprog, 1, bp_lo, bp_hi);
        W::load_words(reinterpret_cast<const uint64_t*>(c.brec + kc), 11, brv_lo, brv_hi);
        W::load_words(reinterpret_cast<const uint64_t*>(c.trec + (kc + 1)), 7, tr_lo, tr_hi);
    }
    PA_HD void older_loads() {
        if (prev_over) {
            W::load_i32s(reinterpret_cast<const int32_t*>(c.d_old + kc), kc + 1 <= c.nblk ? 16 : 8, oldv);
        } else {
            W::load_words(reinterpret_cast<const uint64_t*>(c.prev_brec + kc), 7, pbv_lo, pbv_hi);
            W::load_words(reinterpret_cast<const uint64_t*>(c.prev_brec + (kc + 1 <= c.nblk ? kc + 1 : kc)), 4, pnv_lo, pnv_hi);
            W::load_words(c.prev_done, 1, pdn_lo, pdn_hi);
        }
    }
    PA_HD void prefetch_boundary() {  // at the start of the last chunk of block kc
        older_loads();
        pfo_valid = true;
        boundary_loads(pfb_st, pfb_bp_lo, pfb_bp_hi, pfb_tr_lo, pfb_tr_hi);
        pfb_valid = true;
    }
    // The older passes' records of blocks kc and kc + 1.  While the previous pass runs they are its own bottom-edge records
    // (final as soon as their tags are there: it decides block kc + 1's range right after block kc's fixed range); a record it
    // never writes (the pass ended before) is waited for until the pass is over, then everything comes from the merged array.
    PA_HD bool resolve_older() {
        if (!pfo_valid) older_loads();
        pfo_valid = false;
        old_prev = false;
        if (prev_over) return true;
        const uint32_t pt = blk_tag(c.prev_pass, kc), ptn = blk_tag(c.prev_pass, kc + 1);
        uint32_t spins = 0;
        for (;;) {
            const bool fixed_ok = W::readlane(pbv_hi, kBfs) == pt && W::readlane(pbv_hi, kBfe) == pt;
            const bool range_ok = kc + 1 > c.nblk || (W::readlane(pnv_hi, kBjs) == ptn && W::readlane(pnv_hi, kBje) == ptn &&
                                                      W::readlane(pnv_hi, kBojs) == ptn && W::readlane(pnv_hi, kBoje) == ptn);
            if (fixed_ok && range_ok) {
                old_prev = true;
                return true;
            }
            if (W::readlane(pdn_lo, 0) == c.prev_pass) {  // over and merged: d_old from here on
                prev_over = true;
                older_loads();
                return true;
            }
            W::nap(spins);
            if ((++spins & 63u) == 0) {
                if (pass_over()) {
                    alive = false;
                    return false;
                }
                if (spins > c.spin_limit) {
                    finish(kStTimeout, 4, 0, 0);
                    return false;
                }
            }
            older_loads();
        }
    }
    enum { kBjs = 0, kBje = 1, kBojs = 2, kBoje = 3, kBflags = 4, kBfs = 5, kBfe = 6, kBbot = 7, kBtop = 8, kBsmax = 9, kBspec = 10 };
    enum { kTstate = 0, kTjs = 1, kTtop = 2, kTfs = 3, kTlim = 4, kTfound = 5, kTcont = 6 };
    // field f of the older record of block kc (rec 0: fs = 4, fe = 5) / kc + 1 (rec 1: js, je, ojs, oje = 0..3), BlockRec order
    PA_HD int32_t old_field(int rec, int f) const {
        if (old_prev) {  // the previous pass's BRec: fs, fe are words 5, 6; the range words are 0..3 in both layouts
            const int32_t v = rec == 0 ? W::readlane_i(pbv_lo, f + 1) : W::readlane_i(pnv_lo, f);
            return (rec == 1 && kc + 1 > c.nblk) ? kNone : v;
        }
        return W::readlane_i(oldv, rec * 8 + f);
    }

    PA_HD bool boundary_begin() {
        PA_TRACE("pass %u strip %d boundary_begin block %d js=%d fsprev=%d\n", c.pass, r, kc, js_c, fsprev_c);
        have_n = false;
        sc_active = false;
        bot_interior = false;
        fpend = W::splat(0u);
        const bool scan_owner = fsprev_c >= row0 && fsprev_c < rowE && js_c >= row0;
        const bool dead = !scan_owner && rowE <= js_c;
        const bool need_t = !scan_owner && !dead;
        bool bot_done = false, t_done = !need_t;
        const uint32_t tk = btag(kc), tn = btag(kc + 1);
        uint32_t spins = 0;
        for (;;) {
            vec st_lo, bp_lo, bp_hi, tr_lo, tr_hi;
            if (pfb_valid) {  // first round: the loads issued a chunk ago
                st_lo = pfb_st;
                bp_lo = pfb_bp_lo;
                bp_hi = pfb_bp_hi;
                tr_lo = pfb_tr_lo;
                tr_hi = pfb_tr_hi;
                pfb_valid = false;
            } else {
                boundary_loads(st_lo, bp_lo, bp_hi, tr_lo, tr_hi);
            }
            if (W::readlane(st_lo, 1) != kStRunning || W::readlane(st_lo, 0) == c.pass) {  // over, or given up by the host
                alive = false;
                return false;
            }
            if (!bot_done) {
                // A strip well above the last decided bottom edge does not wait for the decision of block kc: the bottom edge
                // moves up by at most kMaxShrink rows per block in the sense checked by the bottom-edge logic (see there), so
                // after `kc - qb` undecided blocks it is still below my last row.
                const uint32_t bt = W::readlane(bp_hi, 0);
                const int32_t qb = (bt >> 20) == (c.pass & 0xFFFu) ? (int32_t)(bt & 0xFFFFFu) : INT32_MAX;
                if (qb < kc && kc - qb < (1 << 19) && rowE + kMaxShrink * (kc - qb) < W::readlane_i(bp_lo, 0)) {
                    bot_interior = true;
                    bot_done = true;
                } else if (W::readlane(brv_hi, kBje) == tk && W::readlane(brv_hi, kBoje) == tk) {
                    je_c = W::readlane_i(brv_lo, kBje);
                    oje_c = W::readlane_i(brv_lo, kBoje);
                    if (rowE < oje_c) bot_interior = true;  // band rows below my strip in block kc: every lane of mine is inside
                    bot_done = true;
                }
            }
            if (!t_done && W::readlane(tr_hi, kTstate) == tn) {
                const int32_t st = W::readlane_i(tr_lo, kTstate);
                if (st == kTEmpty) {  // the pass ends at block kc; whoever found out has set the status
                    alive = false;
                    return false;
                }
                if (st == kTCont) {
                    if (W::readlane(tr_hi, kTcont) == tn) {
                        int32_t base;  // the strip above has published its prefix word for block kc before the CONT record
                        if (!wait_pr(r - 1, kc, &base)) return false;
                        sc_active = true;
                        sc_j = W::readlane_i(tr_lo, kTcont);
                        sc_e = 0;
                        sc_base0 = base;
                        is_top = true;
                        t_done = true;
                        force_pending(0, lane_of(sc_j));
                    }
                } else if (W::readlane(tr_hi, kTjs) == tn && W::readlane(tr_hi, kTtop) == tn && W::readlane(tr_hi, kTfs) == tn &&
                           W::readlane(tr_hi, kTlim) == tn && W::readlane(tr_hi, kTfound) == tn) {
                    js_n = W::readlane_i(tr_lo, kTjs);
                    top_n = W::readlane_i(tr_lo, kTtop);
                    fs_n = W::readlane_i(tr_lo, kTfs);
                    lim_n = W::readlane_i(tr_lo, kTlim);
                    found_n = W::readlane_i(tr_lo, kTfound);
                    have_n = true;
                    t_done = true;
                }
            }
            if (bot_done && t_done) break;
            W::nap(spins);
            if ((++spins & 63u) == 0 && spins > c.spin_limit) {
                PA_TRACE("pass %u strip %d block %d: boundary timeout bot=%d t=%d\n", c.pass, r, kc, (int)bot_done, (int)t_done);
                finish(kStTimeout, 3, 0, 0);
                return false;
            }
        }
        if (!resolve_older()) return false;
        old_fs_c = old_field(0, 4);
        old_js_n = kc + 1 <= c.nblk ? old_field(1, 0) : kNone;
        if (scan_owner) {  // the top-down scan of block kc starts in my rows
            sc_active = true;
            sc_j = fsprev_c;
            sc_e = lane_of(js_c);
            sc_base0 = top_c;
            is_top = true;
            if (old_fs_c != kNone && sc_j > old_fs_c) sc_j = old_fs_c;  // cannot happen for a growing band; keeps the scan sane
            force_pending(sc_e, lane_of(sc_j));
        } else if (dead) {  // a strip the band has left, finishing its last crossing
            have_n = true;
            js_n = js_c;
            top_n = 0;
            fs_n = fsprev_c;
            lim_n = 0;
            found_n = 0;
        }
        // lane 0's input in block kc + 1
        if (kc < c.nblk) {
            const int32_t jsn = have_n ? js_n : rowE;  // scan owner: the next block starts at or below my first row
            gran_on = jsn < row0;
        }
        return true;
    }

    // ---- block boundary, part 2: every lane has left block kc -----------------------------------------------------------
    PA_HD void boundary_end() {
        const int32_t jeb = bot_interior ? INT32_MAX : je_c;
        const int32_t mrows = c.wtot * 64;
        const typename W::mask act = W::and_m(W::and_m(W::ge_i(lrow0, js_c), W::lt_i(lrow0, jeb)), W::lt_i(lrow0, mrows));
        // column of block kc (V words, sparse blocks of the traceback: blocks.rs:322-339)
        if (!c.store_cols) {  // ring of columns: the slot's previous block must be behind the bottom-edge logic, its only reader
            uint32_t spins = 0;
            for (;;) {
                const uint64_t bp = W::load_u64(c.bprog);
                if ((tw_tag(bp) >> 20) == (c.pass & 0xFFFu) && (int32_t)(tw_tag(bp) & 0xFFFFFu) > kc - c.col_ring) break;
                W::nap(spins);
                if ((++spins & 63u) == 0) {
                    if (pass_over()) {
                        alive = false;
                        return;
                    }
                    if (spins > c.spin_limit) {
                        finish(kStTimeout, 2, 0, 0);
                        return;
                    }
                }
            }
        }
        {
            uint64_t* colk = c.col + (col_slot(c, kc) * c.col_stride - col_base_word(c, kc)) * 2;
            const int32_t wlo = col_base_word(c, kc), whi = wlo + (int32_t)c.col_stride;
            const int32_t wfirst = imax32(row0, js_c) >> 6, wlast = (imin32(imin32(rowE, jeb), mrows) >> 6);
            if (wfirst < wlast && (wfirst < wlo || wlast > whi)) {
                abort_pass(kAbortWindow);
                return;
            }
            W::store_v_halves(colk, (uint32_t)(row0 >> 6), lane, act, snap_p, snap_m);
        }
        const vec val = W::select(act, W::popc_v(snap_p) - W::popc_v(snap_m), W::splat(0u));
        int32_t pbase;
        if (js_c >= row0) {
            pbase = top_c;
        } else if (pfp_valid && W::readlane(pfp_hi, 0) == btag(kc)) {  // fetched during the last crossing chunk
            pbase = W::readlane_i(pfp_lo, 0);
        } else if (!wait_pr(r - 1, kc, &pbase)) {
            return;
        }
        pfp_valid = false;
        // index_kc at every lane's first row is only needed where a scan looks at the column: in the strips at the band's edges
        const bool edge = sc_active || (!bot_interior && je_c > row0 && je_c <= rowE);
        int32_t p_end;
        if (edge) {
            const vec excl = W::prefix_excl(val);
            basev = excl + (uint32_t)pbase;
            p_end = pbase + (int32_t)(W::readlane(excl, 63) + W::readlane(val, 63));
        } else {
            p_end = pbase + (int32_t)W::reduce_add(val);
        }
        if (!pr_word(c, r, kc)) {
            abort_pass(kAbortWindow);
            return;
        }
        flush_deferred();  // (nothing pending in practice: the previous boundary's words went out a block ago)
        pend_k = kc;
        pend_p = p_end;

        // the scan ran off my last lane
        if (sc_active) {
            const int32_t ie = blk_end(c, kc);
            for (;;) {
                const bool past_end = !bot_interior && sc_j > imin32(oje_c, c.m);
                if ((old_fs_c != kNone && sc_j >= old_fs_c) || (past_end && old_fs_c != kNone)) {
                    const int32_t jo = floor64(old_fs_c);
                    if (jo < imax32(row0, js_c) || jo > rowE) {
                        abort_pass(kAbortOldAbove);
                        return;
                    }
                    scan_finish(old_fs_c, 0, sc_j, index_local(jo, p_end));
                    break;
                }
                if (past_end) {
                    scan_empty();
                    return;
                }
                if (sc_j != rowE || bot_interior || je_c > rowE) break;  // continues in the strip below
                const int32_t f = p_end + heur_h(hp, ie, sc_j);        // row rowE itself, the band ends here
                if (f <= c.f_max) {
                    scan_finish(sc_j, 1, sc_j, index_local(floor64(sc_j), p_end));
                    break;
                }
                sc_j += c.sparse_h ? div_ceil_pos(f - c.f_max, 2) : 1;
                if (old_fs_c != kNone && sc_j > old_fs_c) sc_j = old_fs_c;
            }
            if (!alive) return;
            if (sc_active) {  // hand the scan to the strip below (it reads my prefix word first)
                sc_active = false;
                pend_cont = true;
                pend_cont_j = sc_j;
                have_n = true;  // for me: the next block starts below my rows
                js_n = rowE;
                top_n = 0;
                fs_n = sc_j;
            }
        }
        if (!bot_interior && je_c > row0 && je_c <= rowE) {
            [[maybe_unused]] const uint64_t tq0 = PA_CLK(W);
            bottom_edge(p_end);
            PA_CLK_ADD(t_bottom, W::clock() - tq0);
        }
    }

    // ---- the bottom-edge logic of block kc (domain.rs:318-350, 117-246, 449-455; blocks.rs:205-230) ----------------------
    PA_HD bool index_any(int32_t j, int32_t p_end, int32_t bot_val, int32_t* out) {
        if (j > je_c) {
            *out = bot_val + (j - je_c);  // block.rs:97-99
            return true;
        }
        if (j >= imax32(row0, js_c)) {
            *out = index_local(j, p_end);
            return true;
        }
        return index_above(kc, js_c, top_c, j, out);
    }
    // a word of BRec[kc]: from the record fetched at the start of the boundary, or (written a moment later) from memory
    PA_HD bool brec_word(int idx, const uint64_t* p, int32_t* out) {
        if (W::readlane(brv_hi, idx) == btag(kc)) {
            *out = W::readlane_i(brv_lo, idx);
            return true;
        }
        return wait_word(p, btag(kc), out);
    }
    PA_HD void bottom_edge(int32_t p_end) {
        PA_TRACE("pass %u strip %d bottom edge of block %d: js=%d je=%d oje=%d\n", c.pass, r, kc, js_c, je_c, oje_c);
        const int32_t bot_val = index_local(je_c, p_end);
        // top-edge results of this block
        if (!have_n) {
            // the scan of block kc has not reported yet (it may still run in a strip above/below): wait for its record
            if (!fetch_trec_desc(kc + 1)) return;
            if (!have_n) {  // a CONT record addressed to a strip that is not mine cannot reach the bottom strip
                abort_pass(kAbortMismatch);
                return;
            }
        }
        const int32_t ie = blk_end(c, kc);
        const int32_t end = imin32(oje_c, c.m);
        const int32_t old_fe = old_field(0, 5);
        // bottom-up scan (domain.rs:318-328): the last row >= lim with f <= f_max
        int32_t fe_scan = kNone;
        {
            int32_t e = end;
            while (e >= lim_n) {
                int32_t g;
                if (!index_any(e, p_end, bot_val, &g)) return;
                const int32_t f = g + heur_h(hp, ie, e);
                if (f <= c.f_max) {
                    fe_scan = e;
                    break;
                }
                e -= c.sparse_h ? div_ceil_pos(f - c.f_max, 2) : 1;
            }
        }
        int32_t fs_final = fs_n, fe_final;
        if (fe_scan == kNone) {
            if (old_fs_c == kNone) {  // empty and nothing older: the pass fails here
                finish(kStNoPath, 0, kc, kc - 1);
                return;
            }
            fs_final = old_fs_c;
            fe_final = old_fe;
        } else {
            fe_final = old_fs_c != kNone ? imax32(fe_scan, old_fe) : fe_scan;
        }
        if (fs_final != fs_n) {
            abort_pass(kAbortMismatch);
            return;
        }
        BRec* b = c.brec + kc;
        W::store_u64(&b->fs, tw_make(btag(kc), fs_final));
        W::store_u64(&b->fe, tw_make(btag(kc), fe_final));
        W::store_u64(&b->bot_val, tw_make(btag(kc), bot_val));
        W::store_u64(&b->top_val, tw_make(btag(kc), top_c));
        if (kc == c.nblk) {  // domain.rs:520-523
            if (js_c <= c.m && c.m <= je_c) {
                int32_t dist;
                if (!index_any(c.m, p_end, bot_val, &dist)) return;
                finish(kStDone, dist, kc, kc);
            } else {
                finish(kStNoPath, 0, kc, kc);
            }
            return;
        }
        // the next block's range
        int32_t gu;
        if (!index_any(fe_final, p_end, bot_val, &gu)) return;
        const int32_t kn = kc + 1;
        int32_t flags_c;
        if (!brec_word(kBflags, &b->flags, &flags_c)) return;
        BlockRec old_next;
        old_next.js = old_field(1, 0);
        old_next.je = old_field(1, 1);
        old_next.ojs = old_field(1, 2);
        old_next.oje = old_field(1, 3);
        old_next.fs = old_next.fe = kNone;  // (not looked at: decide_next only needs the range)
        old_next.top_val = old_next.bot_val = 0;
        const NextDecision nd = decide_next(hp, c.f_max, c.sparse_h, ie, blk_end(c, kn), fs_final, fe_final, gu, old_next, (flags_c & 2) != 0);
        if (!nd.ok) {
            finish(kStNoPath, 0, kc, kc);
            return;
        }
        const JRangeOut jr = nd.jr;
        // (the owner of this logic moves from strip to strip: every strip sums its own share and adds it once, when it ends)
        st_num_blocks += nd.d_num_blocks;
        st_unique += nd.d_unique_add - nd.d_unique_sub;
        st_computed += nd.d_computed;
        st_incremental += nd.d_incremental;
        // The speculation rule of the strips above me: a strip that saw block q's end `oje_q` ran ahead through block k without
        // waiting if its last row was < oje_q - kMaxShrink * (k - q).  Block kn breaks that promise iff a strip boundary lies
        // in [oje_kn, max_q(oje_q - kMaxShrink * (kn - q))): such a strip treated rows as inside the band that are not (and the
        // strip that should run this logic for block kn may be among them) -- the host engine redoes the pass.
        int32_t spec_c;
        if (!brec_word(kBspec, &b->specmax, &spec_c)) return;
        const int32_t spec_n = imax32(spec_c, oje_c) - kMaxShrink;
        if (jr.js < js_c || (jr.oje < spec_n && ((spec_n - 1) / kStripRows) * kStripRows >= jr.oje)) {
            abort_pass(kAbortNonMonotone);
            return;
        }
        if (jr.js != js_n) {
            abort_pass(kAbortMismatch);
            return;
        }
        if (jr.je - blk_end(c, kn) > c.win || ie - jr.js > c.win) {
            abort_pass(kAbortWindow);
            return;
        }
        BRec* bn = c.brec + kn;
        W::store_u64(&bn->js, tw_make(btag(kn), jr.js));
        W::store_u64(&bn->ojs, tw_make(btag(kn), jr.ojs));
        W::store_u64(&bn->oje, tw_make(btag(kn), jr.oje));
        W::store_u64(&bn->flags, tw_make(btag(kn), nd.flags));
        W::store_u64(&bn->je, tw_make(btag(kn), jr.je));
        // strips the band covers for the first time start at block kn
        int32_t smax;
        if (!brec_word(kBsmax, &b->smax, &smax)) return;
        const int32_t mrows = c.wtot * 64;
        const int32_t last_new = (imin32(jr.je, mrows) - 1) / kStripRows;
        for (int32_t rr = smax + 1; rr <= last_new && rr < c.nstrips; ++rr) W::store_u64(c.strip_start + rr, tw_make(c.pass, kn));
        W::store_u64(&bn->smax, tw_make(btag(kn), imax32(smax, last_new)));
        W::store_u64(&bn->specmax, tw_make(btag(kn), spec_n));
        W::store_u64(c.bprog, tw_make(btag(kn), jr.oje));
    }

    // ---- deferred publications ------------------------------------------------------------------------------------------
    // The strip's prefix word of block pend_k (and, if the top-down scan runs on into the strip below, the CONT record of
    // block pend_k + 1) go out one chunk after the column stores they vouch for, behind an s_waitcnt vmcnt(0).  Plain integers
    // (no pointers, no arrays): this state must stay in scalar registers -- anything that lands in scratch comes back as a
    // "divergent" value and drags the whole loop onto the vector unit.
    PA_HD void flush_deferred() {
        if (pend_k < 0) return;
        W::drain_stores();
        W::store_u64(pr_word(c, r, pend_k), tw_make(btag(pend_k), pend_p));
        if (pend_cont) {
            TRec* t = c.trec + (pend_k + 1);
            W::store_u64(&t->cont_j, tw_make(btag(pend_k + 1), pend_cont_j));
            W::store_u64(&t->state, tw_make(btag(pend_k + 1), kTCont));
        }
        pend_k = -1;
        pend_cont = false;
    }

    // ---- the strip --------------------------------------------------------------------------------------------------------
    uint64_t* gin;
    uint64_t* gout;
    int32_t gin_base, gout_base;
    uint32_t pf_lo, pf_hi;
    vec pf_glo, pf_ghi;  // the prefetched granule, not yet broadcast (so that the load stays in flight during the chunk)
    flag_t pf_has;

    PA_HD void prefetch_inputs(int32_t q) {
        W::load_codes2(c.a_codes, q, pf_lo, pf_hi);
        const int32_t idx = q - gin_base;
        pf_has = gran_on && 32 * q < c.n && idx >= 0 && idx < (int32_t)c.gran_stride;
        if (pf_has) W::load_words(gin + idx, 1, pf_glo, pf_ghi);
    }
    // Lane j (< 32) of XS = packed pipeline input of column 32q + j: its 2-bit code and the delta coming in from above.
    PA_HD bool decode_inputs(int32_t q, vec& XS) {
        const bool want = gran_on && 32 * q < c.n;
        uint32_t glo = 0, ghi = 0;
        if (want) {
            const int32_t idx = q - gin_base;
            if (idx < 0 || idx >= (int32_t)c.gran_stride) {
                abort_pass(kAbortWindow);
                return false;
            }
            uint64_t g = pf_has ? (((uint64_t)W::readlane(pf_ghi, 0) << 32) | W::readlane(pf_glo, 0)) : W::load_u64(gin + idx);
            uint32_t spins = 0;
            while ((uint32_t)g == 0u) {  // the strip above is not there yet
                W::nap(spins);
                g = W::load_u64(gin + idx);
                if ((++spins & 63u) == 0) {
                    if (pass_over()) {
                        alive = false;
                        return false;
                    }
                    if (spins == c.spin_limit / 2) PA_TRACE("pass %u strip %d block %d: granule slow q=%d\n", c.pass, r, kc, q);
                    if (spins > c.spin_limit) {
                        PA_TRACE("pass %u strip %d block %d: granule timeout q=%d\n", c.pass, r, kc, q);
                        finish(kStTimeout, 1, 0, 0);
                        return false;
                    }
                }
            }
            glo = (uint32_t)g - 0x55555555u;
            ghi = (uint32_t)(g >> 32) - 0x55555555u;
        }
        const typename W::mask upper = W::ne_u(lane & 16u, 0u);
        const vec sh = (lane & 15u) * 2u;
        const vec cw = W::select(upper, W::splat(pf_hi), W::splat(pf_lo));
        const vec code = W::shr_v(cw, sh) & 3u;
        const vec hin2 = want ? (W::shl_v(W::select(upper, W::splat(ghi), W::splat(glo)), sh) & 0xC0000000u) : W::splat(0x80000000u);
        XS = code | hin2;
        return true;
    }

    PA_HD void run(int32_t strip) {
        r = strip;
        row0 = r * kStripRows;
        rowE = row0 + kStripRows;
        lane = W::lane_ids();
        lrow0 = lane * 32u + (uint32_t)row0;
        // wait until the band reaches my rows
        int32_t k0;
        if (!wait_word(c.strip_start + r, c.pass, &k0, true)) return;  // may take the whole pass: only the end of the pass ends it
        kc = k0;
        PA_TRACE("pass %u strip %d starts at block %d\n", c.pass, r, k0);
        bot_interior = false;
        is_top = false;
        sc_active = false;
        have_n = false;
        je_c = oje_c = 0;
        if (!fetch_trec_desc(kc)) return;
        if (!have_n) {
            abort_pass(kAbortMismatch);
            return;
        }
        js_c = js_n;
        top_c = top_n;
        fsprev_c = fs_n;
        gran_on = js_c < row0;
        if (js_c >= row0) is_top = true;
        // profile words and V::one()
        W::load_profile(c.b_prof, (uint32_t)(row0 >> 6), c.wtot, lane, nb0, nb1);
        vp = W::splat(0xFFFFFFFFu);
        vm = W::splat(0u);
        X = W::splat(0u);
        acc_lo = acc_hi = W::splat(0u);
        snap_p = snap_m = basev = W::splat(0u);
        andm = W::splat(0xFFFFFFFFu);
        orm = W::splat(0u);
        fpend = W::splat(0u);
        pfb_valid = false;
        pfp_valid = false;
        pfo_valid = false;
        old_prev = false;
        prev_over = c.prev_brec == nullptr;
        if (is_top && js_c > row0) {  // the band's first row is inside my strip: that lane forces +1 from the start
            const typename W::mask first = W::eq_u(lrow0, (uint32_t)js_c);
            andm = W::select(first, W::splat(3u), andm);
            orm = W::select(first, W::splat(0x80000000u), orm);
        }
        gin = r > 0 ? c.gran + (int64_t)(r - 1) * c.gran_stride : c.gran;
        gout = c.gran + (int64_t)r * c.gran_stride;
        gin_base = r > 0 ? gran_base(c, r - 1) : 0;
        gout_base = gran_base(c, r);
        if (r == 0) gran_on = false;
        const bool has_below = rowE < c.wtot * 64;

        int32_t q = blk_end(c, kc - 1) >> 5;
        const int32_t q_first = q;
        const int32_t q_fin = ((c.n - 1) >> 5) + 3;  // two chunks after the one holding column n - 1, lane 63's accumulators hold the
                                                     // last granule; the iteration after that closes the last block
        bool crossing = false;
        int32_t cx = 0;
        prefetch_inputs(q);
        for (;; ++q) {
            const int32_t t0 = 32 * q;
            // ---- a block boundary starts: lane 0 leaves block kc with this chunk, or (last block) every lane has frozen ----
            const bool fin = kc == c.nblk && !crossing && q >= q_fin;
            if (fin || (kc < c.nblk && t0 == blk_end(c, kc))) {
                // (last block: the strip below needs my last granule to finish ITS block, which my boundary may wait for)
                if (fin && has_below && q - q_first >= 3) publish_granule(q - 3);
                [[maybe_unused]] const uint64_t tb0 = PA_CLK(W);
                const bool okb = boundary_begin();
                PA_CLK_ADD(t_begin, W::clock() - tb0);
                if (!okb) return;
                crossing = true;
                cx = t0;
            }
            bool block_done = false;
            if (fin) {
                PA_TRACE("pass %u strip %d final boundary block %d\n", c.pass, r, kc);
                snap_p = vp;
                snap_m = vm;
                PA_NOUNROLL
                while (sc_active && sc_j < rowE) {
                    scan_probe(lane_of(sc_j));
                    if (!alive) return;
                }
                block_done = true;
            } else {
                vec XS;
                {
                    [[maybe_unused]] const uint64_t tg0 = PA_CLK(W);
                    const bool okd = decode_inputs(q, XS);
                    PA_CLK_ADD(t_wait_gran, W::clock() - tg0);
                    if (!okd) return;
                }
                if (q - q_first >= 3 && has_below) publish_granule(q - 3);  // completed two chunks ago (lane 63 lags 64 steps)
                {
                    [[maybe_unused]] const uint64_t tf0 = PA_CLK(W);
                    flush_deferred();  // (before the prefetches: its drain must not wait for loads issued just now)
                    PA_CLK_ADD(t_flush, W::clock() - tf0);
                }
                prefetch_inputs(q + 1);
                // the next boundary's records / the strip above's prefix word: in flight during this chunk's steps
                if (!crossing && kc < c.nblk && t0 + 32 == blk_end(c, kc)) prefetch_boundary();
                if (crossing && t0 + 31 >= cx + 63 && js_c < row0) {
                    const uint64_t* pw = pr_word(c, r - 1, kc);
                    if (pw) {
                        W::load_words(pw, 1, pfp_lo, pfp_hi);
                        pfp_valid = true;
                    }
                }
                // ---- 32 steps ----
                const bool tail = t0 + 31 >= c.n;   // some lane runs past the last column: its V freezes there
                const bool head = q - q_first < 2;  // lanes whose column is still left of the strip's first column do not move
                [[maybe_unused]] const uint64_t tc0 = PA_CLK(W);
                const int32_t jeb = bot_interior ? INT32_MAX : je_c;
                const int32_t cl0 = t0 - cx;  // the lane that crosses in step 0 of this chunk (crossing chunks)
                if (head && !tail && !crossing) {
                    // The strip's first two chunks: lane l takes up its first column in step 32 q_first + l.  The lanes run freely (what a
                    // lane computes before its first column is garbage nobody reads) and restart from V::one() in the step they begin --
                    // the crossing chunk's reset, one lane per step -- instead of 32 single steps with every lane's V saved and restored
                    // (round 6: 6.5 us per chunk, 7 % of a 10 kbp pass).
                    vec dsp = snap_p, dsm = snap_m;
                    W::template chunk_cross<true, true>(XS, X, vp, vm, nb0, nb1, acc_lo, acc_hi, andm, orm, lane, t0 - 32 * q_first, dsp, dsm, W::splat(1u), W::splat(0u));
                    PA_CLK_ADD(t_cross, W::clock() - tc0);
                } else if (tail || head) {
                    const vec resetm = W::select(W::ge_i(lrow0, jeb), W::splat(1u), W::splat(0u));  // below the band in block kc
                    // strip start / last columns (rare): the general loop, one step at a time
                    PA_NOUNROLL
                    for (int32_t j = 0; j < 32; ++j) {
                        const int32_t t = t0 + j;
                        const int32_t cl = t - cx;
                        if (crossing && cl >= 0 && cl < 64) {
                            if (sc_active && cl == lane_of(sc_j)) {
                                scan_probe(cl);
                                if (!alive) return;
                            }
                            const typename W::mask me = W::eq_u(lane, (uint32_t)cl);
                            snap_p = W::select(me, vp, snap_p);
                            snap_m = W::select(me, vm, snap_m);
                            const typename W::mask mf = W::and_m(me, W::ne_u(fpend, 0u));
                            andm = W::select(mf, W::splat(3u), andm);
                            orm = W::select(mf, W::splat(0x80000000u), orm);
                            const typename W::mask mr = W::and_m(me, W::ne_u(resetm, 0u));
                            vp = W::select(mr, W::splat(0xFFFFFFFFu), vp);
                            vm = W::select(mr, W::splat(0u), vm);
                        }
                        const vec op = vp, om = vm;
                        step(W::readlane(XS, j), j >= 16);
                        // column t - lane in [first column of the strip, n)
                        const typename W::mask live = W::and_m(W::gt_i(lane, t - c.n), W::le_i(lane, t - 32 * q_first));
                        vp = W::select(live, vp, op);
                        vm = W::select(live, vm, om);
                    }
                    PA_CLK_ADD(t_cross, W::clock() - tc0);
                } else if (crossing) {
                    // The lanes cross one per step (lane cl0 + j in step j): snapshot of their block-final V, V::one() below the
                    // band, pending +1 forcing -- all inside the unrolled step code.  The top-down scan interrupts it only at the
                    // steps where the lane holding the scanned row crosses (one to three per block).
                    const bool probe_here = sc_active && lane_of(sc_j) - cl0 < 32;
                    if (!probe_here) {
                        // straight-line variants: (force op in the step) x (lanes that start forcing / reset when they cross)
                        const bool extra = !bot_interior || (is_top && W::any(fpend));
                        const vec resetm = W::select(W::ge_i(lrow0, jeb), W::splat(1u), W::splat(0u));  // below the band in block kc
                        if (extra) W::template chunk_cross<true, true>(XS, X, vp, vm, nb0, nb1, acc_lo, acc_hi, andm, orm, lane, cl0, snap_p, snap_m, resetm, fpend);
                        else W::template chunk_cross<true, false>(XS, X, vp, vm, nb0, nb1, acc_lo, acc_hi, andm, orm, lane, cl0, snap_p, snap_m, resetm, fpend);
                    } else {
                        const vec resetm = W::select(W::ge_i(lrow0, jeb), W::splat(1u), W::splat(0u));
                        int32_t j = 0;
                        PA_NOUNROLL
                        while (j < 32) {
                            int32_t stop = 32;
                            if (sc_active) {
                                const int32_t jh = lane_of(sc_j) - cl0;  // the step in which the scanned row's lane crosses
                                if (jh == j) {
                                    [[maybe_unused]] const uint64_t tp0 = PA_CLK(W);
                                    scan_probe(cl0 + j);
                                    PA_CLK_ADD(t_probe, W::clock() - tp0);
                                    if (!alive) return;
                                    continue;
                                }
                                if (jh > j && jh < 32) stop = jh;
                            }
                            W::chunk_cross_range(XS, X, vp, vm, nb0, nb1, acc_lo, acc_hi, andm, orm, lane, cl0, snap_p, snap_m, resetm, fpend, j, stop);
                            j = stop;
                        }
                    }
                    PA_CLK_ADD(t_cross2, W::clock() - tc0);
                } else {
                    W::template chunk<true>(XS, X, vp, vm, nb0, nb1, acc_lo, acc_hi, andm, orm);
                    PA_CLK_ADD(t_plain, W::clock() - tc0);
                }
                block_done = crossing && t0 + 31 >= cx + 63;  // every lane has left block kc
            }
            // ---- the plain stretch: chunks with no lane crossing and no boundary work ahead run in a loop of their own, so that
            //      the register allocator keeps only what THEY need in registers (the block state lives in spill lanes meanwhile)
            if (!crossing && !block_done && !fin) {
                // chunks q + 1 .. qs - 1: before the chunk that prefetches the next boundary's records, left of the last column
                int32_t qs = (blk_end(c, kc) >> 5) - 1;
                if (kc == c.nblk) qs = c.n >> 5;  // (chunks from there on hold columns past the end: the general loop)
                int32_t qq = q + 1;
                PA_NOUNROLL
                while (qq < qs && qq - q_first >= 2) {
                    vec XS;
                    [[maybe_unused]] const uint64_t tg0 = PA_CLK(W);
                    if (!decode_inputs(qq, XS)) return;
                    [[maybe_unused]] const uint64_t tg1 = PA_CLK(W);
                    PA_CLK_ADD(t_wait_gran, tg1 - tg0);
                    if (qq - q_first >= 3 && has_below) publish_granule(qq - 3);
                    prefetch_inputs(qq + 1);
                    W::stretch_marker();  // (keeps this copy of the chunk from being merged back into the general loop's)
                    W::template chunk<true>(XS, X, vp, vm, nb0, nb1, acc_lo, acc_hi, andm, orm);
                    PA_CLK_ADD(t_plain, W::clock() - tg1);
                    ++qq;
                }
                q = qq - 1;
            }
            if (block_done) {
                [[maybe_unused]] const uint64_t te0 = PA_CLK(W);
                boundary_end();
                PA_CLK_ADD(t_end, W::clock() - te0);
                if (!alive) return;
                crossing = false;
                if (kc == c.nblk) {  // the pass's last block
                    flush_deferred();
                    return;
                }
                kc += 1;
                js_c = js_n;
                top_c = top_n;
                fsprev_c = fs_n;
                if (rowE <= js_c) {  // the band has moved below my rows: retire
                    PA_TRACE("pass %u strip %d retires before block %d (js=%d)\n", c.pass, r, kc, js_c);
                    if (has_below) publish_granule(q - 2);
                    flush_deferred();
                    return;
                }
            }
        }
    }

    // Granule g (columns 32g .. 32g+31 of my bottom row) from lane 63's lagged accumulators.
    PA_HD void publish_granule(int32_t g) {
        if (g < 0 || 32 * g >= c.n) return;
        const int32_t idx = g - gout_base;
        if (idx < 0 || idx >= (int32_t)c.gran_stride) return;  // outside the window: the strip below cannot be there
        const uint32_t vlo = W::readlane(acc_lo, 63), vhi = W::readlane(acc_hi, 63);
        W::store_u64(gout + idx, (((uint64_t)vhi << 32) | (uint64_t)vlo) + 0x5555555555555555ull);
    }
};

// The wave program: claim strips by ticket (earlier wavefronts take upper strips, so a consumer's producer has always
// started) and run them one after the other.
template <class W>
PA_HD void wave_main(const Ctx& c) {
    const uint32_t w = W::ticket(c.ticket);
    for (int32_t strip = (int32_t)w; strip < c.nstrips; strip += c.nwaves) {
        StripProg<W> prog(c);
        prog.run(strip);
        prog.commit_finish();
        if (prog.st_num_blocks) {
            PassStats& st = c.status->stats;
            W::add_u64(&st.num_blocks, prog.st_num_blocks);
            W::add_u64(&st.unique_lanes, prog.st_unique);
            W::add_u64(&st.computed_lanes, prog.st_computed);
            W::add_u64(&st.num_incremental_blocks, prog.st_incremental);
        }
        if (c.timing) {  // phase clocks of all strips, summed (diagnostics: PA_SWEEP_TIMING)
            W::add_u64(c.timing + 0, prog.t_begin);
            W::add_u64(c.timing + 1, prog.t_cross);
            W::add_u64(c.timing + 2, prog.t_end - prog.t_bottom);
            W::add_u64(c.timing + 3, prog.t_bottom);
            W::add_u64(c.timing + 4, prog.t_plain);
            W::add_u64(c.timing + 5, prog.t_wait_gran + (prog.t_flush << 32));  // (diagnostics: the drain of the deferred publications rides in the upper half)
            W::add_u64(c.timing + 6, prog.t_cross2);
            W::add_u64(c.timing + 7, prog.t_probe);
        }
        if (W::load_u32(&c.status->state) != kStRunning || W::load_u64(c.cancel) == (uint64_t)c.pass) return;
    }
}

}  // namespace sweep
}  // namespace pa
