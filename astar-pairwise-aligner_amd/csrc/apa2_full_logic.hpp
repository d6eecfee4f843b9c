// apa2_full_logic.hpp -- the per-pair band search of the WHOLE A*PA2 family as one flat program: any heuristic behind `h(i, j)`
// (NoCost, GapCost, SH, GCSH), incremental doubling with its stored row of horizontal differences, pruning of matches between blocks.
//
// Backends: csrc/apa2_full_kernel.hpp (gfx950: pa_batch_create_params with AstarPa2Params::full() and its relatives, round 4) and
// oracle/apa2_full_emu.cpp (tests only: the CPU kernels and csrc/gcsh.hpp, compared with the host engine and the second restatement in
// tests/test_apa2_full_emu.py).  apa2_logic.hpp is the program of the `simple` family (wave-parallel scans instead of probes); this one
// covers the whole family: flat per-block records, block columns addressed by absolute word in per-block slots (so "keep the words an
// older pass fixed" is "leave them where they are"), compute calls over word ranges with the stored row of horizontal differences read
// at the top and / or tapped on the way, and the heuristic / pruning behind four backend calls.
//
// What it restates:
//   domain.rs:117-246   j_range for Domain::Astar with the literal probing loops (GCSH is not monotone along a column, so the
//                       galloping of sweep_logic.hpp does not apply)
//   domain.rs:251-350   fixed_j_range (literal jumping loops)
//   domain.rs:356-541   align_for_bounded_dist: reuse test, pruning of the block's matches, update of the contours per pass
//   blocks.rs:146-197   init / reuse_next_block            blocks.rs:205-469   compute_next_block incl. incremental doubling
//   blocks.rs:753-831   init_v_with_overlap / init_v_with_overlap_preserve_fixed (as word ranges of the slot)
//   lib.rs:122-175, band.rs:100-182   the band search around it
//
// Backend B:
//   FullRec load_rec(k) / void store_rec(k, rec)
//   int32_t index(k, rec, j)                       Block::index of slot k
//   void init_plain(k, prev, cur)                  slot k := prev's words where the ranges overlap, V::one() elsewhere
//   void init_preserve(k, prev, cur, p0, p1, prev_w1)   words [p0, p1) of slot k stay; [cur.js/64, p0) and [p1, copy_end) come from
//                                                  slot k - 1, the rest is V::one()   (copy_end = min(cur.je/64, prev_w1))
//   int32_t compute2(k, i0, i1, w0, wt, w1, hin, tap)   columns [i0, i1) over words [w0, w1) of slot k as ONE range; the top row is the
//                                                  stored row of horizontal differences (hin) or +1; with `tap` the differences of row
//                                                  64 wt (w0 <= wt <= w1) replace the stored row on the way (wt == w0: the top row
//                                                  itself).  Returns the bottom row's sum.  In the reference's terms (blocks.rs:662-748):
//                                                  (!hin, !tap) = None; (hin, !tap) = Input; (!hin, tap) = Output over [w0, wt) then
//                                                  Input over [wt, w1); (hin, tap) = Update over [w0, wt) then Input over [wt, w1)
//   int32_t h(i, j); void prune_block(i0, i1, j0, j1); void update_contours()
//   bool failed()                                  the backend could not compute the last block (the search stops; device only)
//   int32_t uniform(x)                             x, known to be the same in every lane (device: back into a scalar register)
#pragma once
#include <math.h>
#include <stdint.h>

#include "sweep_logic.hpp"

namespace pa {
namespace apa2 {

enum : int32_t { kHNone = 0, kHInput = 1, kHUpdate = 2, kHOutput = 3 };  // blocks.rs:662-668 (engine.hpp HMode)

struct FullRec {
    int32_t js, je;      // j_range rounded out (kNone: no such block yet)
    int32_t ojs, oje;    // original_j_range
    int32_t fs, fe;      // fixed_j_range (kNone: unset)
    int32_t top_val, bot_val;
    int32_t j_h;         // row of the stored horizontal differences (kNone: none)
    int32_t pad[3];
};
static_assert(sizeof(FullRec) == 48, "FullRec layout");

struct FullParams {
    int32_t sparse_h, prune, incremental;
    int32_t doubling;   // 1 band doubling, 2 linear search (engine.hpp DoublingKind)
    int32_t start;      // 0 zero, 1 gap, 2 h0
    float factor;
    int32_t delta;
};

struct FullResult {
    int32_t status, cost, f_max;
    uint32_t f_max_tries, sanity_violations, num_blocks, num_incremental_blocks;
    uint64_t computed_lanes, unique_lanes;
    int32_t last_block_idx, blocks_len;
};

enum : int32_t { kFullOk = 0, kFullErrOrder = 1, kFullErrRange = 2, kFullErrPasses = 3, kFullErrH0 = 4, kFullErrSplit = 5, kFullErrBackend = 6 };

template <class B>
struct PairProgFull {
    static constexpr int32_t kNone = sweep::kNone;
    static constexpr int32_t kBlockW = sweep::kBlockW;
    B& be;
    FullParams sp;
    int32_t n, m, nblk;
    int32_t last_block_idx = 0, blocks_len = 0;
    uint32_t f_max_tries = 0, sanity = 0, num_blocks = 0, num_incremental = 0;
    uint64_t computed_lanes = 0, unique_lanes = 0;
    int32_t err = kFullOk;

    PA_HD PairProgFull(B& backend, const FullParams& p, int32_t n_, int32_t m_) : be(backend), sp(p), n(n_), m(m_) { nblk = (n + kBlockW - 1) / kBlockW; }

    PA_HD static int32_t up64(int32_t x) { return sweep::ceil64(x); }
    PA_HD static int32_t dn64(int32_t x) { return sweep::floor64(x); }

    // ---- domain.rs:117-246 (Astar): the last row of the next block's range (its first row is the previous block's fixed start) ----
    PA_HD int32_t j_range_end(int32_t is, int32_t ie, int32_t fixed_end, int32_t gu, int32_t f_max) {
        const int32_t u0 = is, u1 = fixed_end;
        int32_t v0 = u0, v1 = u1;
        auto f = [&](int32_t x, int32_t y) { return gu + sweep::iabs32((x - u0) - (y - u1)) + be.h(x, y); };
        if (!sp.sparse_h) {
            while (v0 < ie) {
                v0 += 1;
                v1 += 2;
                while (v1 <= m && f(v0, v1) <= f_max) v1 += 1;
                v1 -= 1;
            }
            return v1;
        }
        v0 += 1;
        v1 += 1;
        v1 += kBlockW;
        if (v1 > m) v1 = m;
        for (;;) {
            if (v1 < v0 - u0 + u1) {
                v1 = v0 - u0 + u1;
                break;
            }
            const int32_t fv = f(v0, v1);
            if (fv <= f_max) {
                if (v1 == m) break;
                v1 += 8;
                if (v1 >= m) v1 = m;
            } else {
                v0 += sweep::div_ceil_pos(fv - f_max, 2);
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
            const int32_t fv = f(v0, v1);
            if (fv <= f_max) break;
            v1 -= sweep::div_ceil_pos(fv - f_max, 2);
        }
        return v1;
    }

    // The statistics of one compute_block call (blocks.rs:686-748).
    PA_HD void count_range(int32_t i0, int32_t i1, int32_t w0, int32_t w1) {
        if (i1 - i0 > 1) {
            computed_lanes += (uint64_t)(w1 - w0);
            num_incremental += 1;
        }
    }

    // blocks.rs:205-469 for the sparse traced engine.  `old` = the block an older pass left at this index (js == kNone: none).
    PA_HD bool compute_next_block(int32_t k, const FullRec& prev, const FullRec& old, int32_t s, int32_t e, int32_t i0, int32_t i1, FullRec* out) {
        num_blocks += 1;
        FullRec cur;
        cur.ojs = s;
        cur.oje = e;
        cur.js = dn64(s);
        cur.je = up64(e);
        unique_lanes += (uint64_t)((cur.je - cur.js) / 64);
        if (old.js != kNone) {
            if (!(cur.js <= old.js && old.je <= cur.je)) {  // "j_range must grow"
                err = kFullErrRange;
                return false;
            }
            unique_lanes -= (uint64_t)((old.je - old.js) / 64);
        }
        if (cur.js < prev.js) {
            err = kFullErrOrder;
            return false;
        }
        cur.fs = old.js != kNone ? old.fs : kNone;  // the block's fixed range survives its re-computation (blocks.rs:308)
        cur.fe = old.js != kNone ? old.fe : kNone;
        cur.top_val = be.index(k - 1, prev, cur.js) + (i1 - i0);
        cur.bot_val = be.index(k - 1, prev, cur.je);
        cur.j_h = kNone;
        cur.pad[0] = cur.pad[1] = cur.pad[2] = 0;
        if (k == blocks_len) blocks_len += 1;
        const int32_t w0 = cur.js / 64, w1 = cur.je / 64;
        if (!sp.incremental || prev.fs == kNone) {
            be.init_plain(k, prev, cur);
            count_range(i0, i1, w0, w1);
            cur.bot_val += be.compute2(k, i0, i1, w0, w0, w1, false, false);  // HMode::None
            *out = cur;
            return true;
        }
        // ---- incremental doubling (blocks.rs:341-469) ----
        const int32_t new_j_h = dn64(prev.fe);  // prev_fixed.round_in().1
        cur.j_h = new_j_h;
        if (new_j_h < cur.js || new_j_h > cur.je) {
            err = kFullErrSplit;
            return false;
        }
        if (old.js != kNone && old.j_h != kNone && old.fs != kNone && up64(old.fs - 1) < old.j_h) {
            // 3 ranges: above the old fixed part (plain), old j_h .. new j_h (the stored row is updated), below (the stored row is input);
            // the words between stay as the older pass left them
            const int32_t p0 = up64(old.fs - 1) / 64, p1 = old.j_h / 64;
            if (old.j_h > new_j_h) {  // "j_h may only increase"
                err = kFullErrSplit;
                return false;
            }
            const int32_t wt = new_j_h / 64;
            be.init_preserve(k, prev, cur, p0, p1, prev.je / 64);
            count_range(i0, i1, w0, p0);
            be.compute2(k, i0, i1, w0, w0, p0, false, false);  // HMode::None
            // HMode::Update over [p1, wt) (only if not empty) and HMode::Input over [wt, w1): one range from the stored row at the
            // top, the deltas of row 64 wt stored on the way
            if (wt > p1) count_range(i0, i1, p1, wt);
            count_range(i0, i1, wt, w1);
            cur.bot_val += be.compute2(k, i0, i1, wt > p1 ? p1 : wt, wt, w1, true, wt > p1);
        } else {
            const int32_t wt = new_j_h / 64;
            be.init_plain(k, prev, cur);
            // HMode::Output over [w0, wt) (runs even if empty: it sets the stored row) and HMode::Input over [wt, w1): one range
            count_range(i0, i1, w0, wt);
            count_range(i0, i1, wt, w1);
            cur.bot_val += be.compute2(k, i0, i1, w0, wt, w1, false, true);
        }
        *out = cur;
        return true;
    }

    // One align_for_bounded_dist (domain.rs:356-541).  true: Some(dist); false: None.  Check `err` afterwards.
    PA_HD bool pass(int32_t f_max, int32_t* dist) {
        f_max_tries += 1;
        if (sp.prune && !external_update) be.update_contours();
        FullRec none;
        none.js = none.je = none.ojs = none.oje = none.fs = none.fe = none.j_h = kNone;
        none.top_val = none.bot_val = 0;
        none.pad[0] = none.pad[1] = none.pad[2] = 0;
        // ---- the first column: j_range((-1, 0), prev.fixed = (-1, -1)) united with what blocks.next_block_j_range() holds BEFORE init ----
        FullRec stale = none;
        if (last_block_idx + 1 < blocks_len) stale = be.load_rec(last_block_idx + 1);
        int32_t s0 = -1, e0 = j_range_end(-1, 0, -1, 0, f_max);
        if (stale.js != kNone) {
            s0 = s0 < stale.js ? s0 : stale.js;
            e0 = e0 > stale.je ? e0 : stale.je;
        }
        if (s0 < 0) s0 = 0;
        if (e0 > m) e0 = m;
        if (s0 > e0 || s0 > 0) return false;
        FullRec prev = none;
        prev.ojs = s0;
        prev.oje = e0;
        prev.js = 0;
        prev.je = up64(e0);
        if (blocks_len > 0) {
            const FullRec old0 = be.load_rec(0);
            if (old0.je > prev.je) prev.je = old0.je;
        }
        prev.fs = s0;
        prev.fe = e0;
        prev.top_val = 0;
        prev.bot_val = prev.je;
        be.store_rec(0, prev);
        if (blocks_len < 1) blocks_len = 1;
        last_block_idx = 0;

        bool all_reused = true;
        for (int32_t k = 1; k <= nblk; ++k) {
            const int32_t i0 = (k - 1) * kBlockW, i1 = k * kBlockW < n ? k * kBlockW : n;
            FullRec old = none;
            if (k < blocks_len) old = be.load_rec(k);
            // ---- j_range (domain.rs:117-246) ----
            const int32_t gu = be.index(k - 1, prev, prev.fe);
            int32_t s = prev.fs, e = j_range_end(i0, i1, prev.fe, gu, f_max);
            if (old.js != kNone) {
                s = s < old.js ? s : old.js;
                e = e > old.je ? e : old.je;
            }
            if (s < 0) s = 0;
            if (e > m) e = m;
            if (s > e) return false;
            const bool reuse = all_reused && old.js != kNone && old.js == s && old.je == e;  // (the exact new range against the rounded old one)
            all_reused = all_reused && reuse;
            const int32_t pfs = prev.fs, pfe = prev.fe;
            FullRec cur;
            if (reuse) {
                cur = old;
            } else {
                if (!compute_next_block(k, prev, old, s, e, i0, i1, &cur)) return false;
                if (be.failed()) {  // (device: a strip error, or the block left the pair's window of the column store)
                    err = kFullErrBackend;
                    return false;
                }
                be.store_rec(k, cur);  // (the block is overwritten in place: a pass that ends at this block's fixed range leaves it like this)
            }
            last_block_idx = k;
            // ---- fixed_j_range (domain.rs:251-350) ----
            if (cur.js > pfs) {
                err = kFullErrOrder;
                return false;
            }
            int32_t start = pfs, end = cur.oje < m ? cur.oje : m;
            while (start <= end) {
                const int32_t fv = be.index(k, cur, start) + be.h(i1, start);
                if (fv <= f_max) break;
                start += sp.sparse_h ? sweep::div_ceil_pos(fv - f_max, 2) : 1;
            }
            while (end >= start) {
                const int32_t fv = be.index(k, cur, end) + be.h(i1, end);
                if (fv <= f_max) break;
                end -= sp.sparse_h ? sweep::div_ceil_pos(fv - f_max, 2) : 1;
            }
            int32_t fs = start, fe = end;
            if (cur.fs != kNone) {
                if (fs > fe) {
                    fs = cur.fs;
                    fe = cur.fe;
                } else {
                    fs = fs < cur.fs ? fs : cur.fs;
                    fe = fe > cur.fe ? fe : cur.fe;
                }
            }
            if (fs > fe) return false;
            cur.fs = fs;
            cur.fe = fe;
            be.store_rec(k, cur);
            // ---- prune the matches of this block that start in rows fixed before and after it (domain.rs:505-515) ----
            if (sp.prune) {
                const int32_t a0 = pfs > fs ? pfs : fs, a1 = pfe < fe ? pfe : fe;
                if (a0 <= a1) be.prune_block(i0, i1, a0, a1);
            }
            prev = cur;
        }
        if (m < prev.js || m > prev.je) return false;
        *dist = be.index(nblk, prev, m);
        return true;
    }

    PA_HD int32_t next_bound(int32_t s, int32_t offset) const {
        if (sp.doubling == 2) return s + sp.delta;
        const int32_t c = be.uniform((int32_t)ceilf(sp.factor * (float)(s - offset)));  // band.rs:138, f32 (a vector unit on the GPU)
        return (c > 1 ? c : 1) + offset;
    }

    // ---- lib.rs:122-175 + band.rs:100-182, one pass per step ----
    // The search is resumable: everything it carries from pass to pass is in SearchState (plus the block records and columns the
    // backend keeps), so a device can run ONE pass per launch and let the host re-derive the contours of the pairs that need another
    // pass in between (with `external_update` the program does not call update_contours itself).
    struct SearchState {
        int32_t h0, offset, s, last_s, maxs, it;
        int32_t cost, f_ok;
        int32_t done;  // 0 running, 1 found, 2 gave up (err / bound ran away / too many passes)
        // what PairProgFull itself carries
        int32_t last_block_idx, blocks_len, err;
        uint32_t f_max_tries, sanity, num_blocks, num_incremental;
        uint64_t computed_lanes, unique_lanes;
    };
    bool external_update = false;

    PA_HD void save(SearchState* st) const {
        st->last_block_idx = last_block_idx;
        st->blocks_len = blocks_len;
        st->err = err;
        st->f_max_tries = f_max_tries;
        st->sanity = sanity;
        st->num_blocks = num_blocks;
        st->num_incremental = num_incremental;
        st->computed_lanes = computed_lanes;
        st->unique_lanes = unique_lanes;
    }
    PA_HD void load(const SearchState& st) {
        last_block_idx = st.last_block_idx;
        blocks_len = st.blocks_len;
        err = st.err;
        f_max_tries = st.f_max_tries;
        sanity = st.sanity;
        num_blocks = st.num_blocks;
        num_incremental = st.num_incremental;
        computed_lanes = st.computed_lanes;
        unique_lanes = st.unique_lanes;
    }

    PA_HD void begin(SearchState* st) {
        st->h0 = be.h(0, 0);
        int32_t start_f = 0, start_inc = 1;
        if (sp.start == 1) {
            start_f = start_inc = sweep::iabs32(n - m);
        } else if (sp.start == 2) {
            start_f = st->h0;
            start_inc = 1;
        }
        st->offset = start_f;
        if (sp.doubling == 2) {
            st->s = start_f;
        } else {
            if (start_inc < kBlockW) start_inc = kBlockW;
            st->s = st->offset + start_inc;
        }
        st->last_s = -1;
        st->maxs = INT32_MAX;
        st->it = 0;
        st->cost = st->f_ok = 0;
        st->done = 0;
        save(st);
    }

    // One pass with the bound st->s; afterwards st->done says whether another step is needed (then st->s is its bound).
    PA_HD void step(SearchState* st) {
        load(*st);
        if (st->done == 0 && (st->it >= 4000 || err != kFullOk)) st->done = 2;
        if (st->done != 0) return;
        int32_t dist = 0;
        const bool some = pass(st->s, &dist);
        st->it += 1;
        if (err != kFullOk) {
            st->done = 2;
        } else {
            bool found = false;
            if (some) {
                if (dist > st->maxs) sanity += 1;
                if (dist <= st->s) {
                    if (dist <= st->last_s) sanity += 1;
                    st->cost = dist;
                    st->f_ok = st->s;
                    found = true;
                } else if (dist < st->maxs) {
                    st->maxs = dist;
                }
            } else if (st->maxs != INT32_MAX) {
                sanity += 1;
            }
            if (found) {
                st->done = 1;
            } else {
                const int32_t before = st->s;
                st->last_s = st->s;
                const int32_t nx = next_bound(st->s, st->offset);
                st->s = nx < st->maxs ? nx : st->maxs;
                if (st->s <= before) st->s = next_bound(before, st->offset);
                if (st->s < 0 || st->s > 4 * (n + m) + 8 * kBlockW) st->done = 2;
            }
        }
        save(st);
    }

    PA_HD void finish(const SearchState& st, FullResult* out) {
        load(st);
        if (err == kFullOk && st.done != 1) err = kFullErrPasses;
        if (err == kFullOk && st.h0 > st.cost) err = kFullErrH0;
        out->status = err;
        out->cost = st.cost;
        out->f_max = st.f_ok;
        out->f_max_tries = f_max_tries;
        out->sanity_violations = sanity;
        out->num_blocks = num_blocks;
        out->num_incremental_blocks = num_incremental;
        out->computed_lanes = computed_lanes;
        out->unique_lanes = unique_lanes;
        out->last_block_idx = last_block_idx;
        out->blocks_len = blocks_len;
    }

    // The whole search in one go (what a single launch per pair would run).
    PA_HD void run(FullResult* out) {
        SearchState st;
        begin(&st);
        while (st.done == 0) step(&st);
        finish(st, out);
    }
};

}  // namespace apa2
}  // namespace pa
