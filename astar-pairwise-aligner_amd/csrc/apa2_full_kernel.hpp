// apa2_full_kernel.hpp -- the gfx950 backend of apa2_full_logic.hpp: ONE WAVEFRONT runs the whole band search of one pair for the
// WHOLE A*PA2 family -- AstarPa2Params::full() (GCSH k = 12 with local pruning p = 14, pruning of matches between blocks, incremental
// doubling with the stored row of horizontal differences; astarpa2/src/params.rs:98-128) and every other Domain::Astar parameter set
// over sparse 256-column blocks that apa2_kernel.hpp does not take.
//
// What it replaces: `for (a, b) in pairs { aligner.align(a, b) }` (pa-bin/src/main.rs:24-35) over astarpa2/src/lib.rs:122-175,
// band.rs:100-182, domain.rs:117-541, blocks.rs:146-469 and, per probe / block / pass, pa-heuristic's csh.rs:341-376 (h),
// prune.rs:245-292 (prune_block) and csh.rs:497-554 + contour/hint_contours.rs:213-272 (contours re-derived).
//
// MI355X-first shape (next to what apa2_kernel.hpp already does: persistent grid, pairs by ticket, strips of strip_kernel.hpp, block
// columns in HBM at their absolute words, wave-parallel Block::index):
//  * The heuristic lives on the GPU.  A contour layer is a short linked list whose newest point sits inline in the layer's 16-byte
//    record (gcsh_dev.hpp); one probe of h(i, j) tests 64 LAYERS AT ONCE, one per lane, in a window around the previous answer --
//    layers are nested, so the score is the highest lane that says yes; a window that misses is followed by 64-ary bracketing.
//    The reference's hinted linear probe (hint_contours.rs:283-344) becomes one load round.
//  * The contours are (re-)derived by the same wavefront: matches from the last start to the first, one wave-parallel score each
//    (hint_contours.rs:213-255); between two passes the pruned matches are simply left out (csh.rs:525-545 reaches the same layers).
//  * prune_block: one lane per seed of the block (at most 64 seeds per 256 columns for k >= 4) runs the two-pointer windows of
//    prune.rs:245-292.
//  * Incremental doubling WITHOUT extra strips: the rows above and below the stored row j_h -- HMode::Output + HMode::Input, or
//    HMode::Update + HMode::Input (blocks.rs:406-468), two operator calls in the reference -- run as ONE strip whose lane at row j_h
//    "taps" its outgoing horizontal deltas into the stored row (run_strip<.., TAP>): a block of `full` (about ten 64-row words) is one
//    half-wave strip, as in the `simple` kernel.
//  * The probing loops of j_range / fixed_j_range stay literal (GCSH with local pruning is not consistent: a jump may skip rows that
//    would pass, and the reference's results depend on where the probes land).
#pragma once
#include "apa2_jobs.hpp"
#include "apa2_full_logic.hpp"
#include "apa2_kernel.hpp"
#include "gcsh_dev.hpp"
#include "strip2_kernel.hpp"
#include "strip_kernel.hpp"

namespace pa {
namespace apa2 {


typedef int32_t pa_i32x4 __attribute__((ext_vector_type(4)));

struct FullDevBackend {
    FullJob job;
    GcshDev g;
    uint32_t* err;
    uint32_t* dbg;
    int lane;
    int32_t hint = 0;      // the last score (the next probe's window is centred on it)
    bool dirty = false;    // matches were pruned since the contours were derived
    mutable uint32_t strip_units = 0;
    // the rendezvous of half-wave blocks (strip2_kernel.hpp, round 5): a block of `full` is about ten words -- twenty lanes -- so two pairs'
    // blocks run as ONE strip whenever two wavefronts of the workgroup reach theirs within each other's patience
    RdvLds rdv{nullptr, 0};
    int wave = 0;
    RdvParams rp{0u, 0u, 0u, 0u};
    mutable rdv::Counters rdv_cnt;
    int32_t my_prio = 0;  // this wavefront's issue priority while it runs this pair (RdvParams::prio)
    uint32_t n_probe = 0, n_round = 0;  // diagnostics: h probes and the load rounds they took
    bool timing = false;                // diagnostics (PA_APA2_PROBE_STATS): phase clocks, 100 MHz ticks
    mutable uint64_t t_build = 0, t_dp = 0, t_h = 0, t_index = 0, t_prune = 0, t_init = 0;
    __device__ __forceinline__ uint64_t tick() const { return timing ? wall_clock64() : 0; }

    __device__ __forceinline__ FullDevBackend(const FullJob& j, uint32_t* e, uint32_t* d) : err(e), dbg(d) {
        lane = (int)(threadIdx.x & 63);
        job.a_codes = own_sgpr(j.a_codes);
        job.b_prof = own_sgpr(j.b_prof);
        job.rec = own_sgpr(j.rec);
        job.jh = own_sgpr(j.jh);
        job.col = own_sgpr(j.col);
        job.col_stride = own_sgpr(j.col_stride);
        job.hrow = own_sgpr(j.hrow);
        job.sh_h = own_sgpr(j.sh_h);
        job.gran = own_sgpr(j.gran);
        job.sum = own_sgpr(j.sum);
        job.result = own_sgpr(j.result);
        job.n = own_sgpr(j.n);
        job.m = own_sgpr(j.m);
        job.heur = own_sgpr(j.heur);
        job.slot_ratio = own_sgpr(j.slot_ratio);
        g.mi = own_sgpr(j.g.mi);
        g.mj = own_sgpr(j.g.mj);
        g.active = own_sgpr(j.g.active);
        g.win = own_sgpr(j.g.win);
        g.lrec = own_sgpr(j.g.lrec);
        g.cell = own_sgpr(j.g.cell);
        g.nmatch = own_sgpr(j.g.nmatch);
        g.nlayers = 1;
        g.n = job.n;
        g.m = job.m;
        g.k = own_sgpr(j.g.k);
        g.nseeds = own_sgpr(j.g.nseeds);
        g.prune = own_sgpr(j.g.prune);
        g.pad = 0;
        if (g.k >= 1 && g.k < 32 && g.n < (1 << 26) - 64) div_m = (uint32_t)((1ull << 31) / (uint64_t)g.k) + 1u;
    }
    __device__ __forceinline__ uint64_t strip_instructions() const { return (uint64_t)strip_units << 5; }
    __device__ __forceinline__ int32_t uniform(int32_t x) const { return (int32_t)rfl((uint32_t)x); }
    mutable bool win_fail = false;  // a block left the window of the column store: the pair runs again with full-height slots
    __device__ __forceinline__ bool failed() const {
        return win_fail || rfl(__hip_atomic_load((const PA_GLOBAL uint32_t*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != PA_ERR_NONE;
    }
    __device__ __forceinline__ sweep::SlotGeom geom() const { return sweep::SlotGeom{job.n, job.m, (int32_t)job.col_stride, job.slot_ratio}; }
    // slot k, addressed by ABSOLUTE word (the pointer is moved back by the window's first word)
    // (a block's logic asks for slots k and k - 1 a dozen times: the last two answers are kept)
    mutable int32_t sk0 = -1, sk1 = -1;
    mutable gu32 sp0 = nullptr, sp1 = nullptr;
    __device__ __forceinline__ gu32 slot(int32_t k) const {
        if (k == sk0) return sp0;
        if (k == sk1) return sp1;
        const gu32 q = (gu32)job.col + ((int64_t)k * job.col_stride - (int64_t)sweep::slot_off(geom(), k)) * 4;
        sk1 = sk0;
        sp1 = sp0;
        sk0 = k;
        sp0 = q;
        return q;
    }
    __device__ __forceinline__ void sync_mem() const { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }  // same wavefront writes, then reads

    // ---- block records --------------------------------------------------------------------------------------------------------
    mutable bool rec_dirty = false;  // records stored since the last fence
    __device__ __forceinline__ FullRec load_rec(int32_t k) const {
        if (rec_dirty) {
            sync_mem();
            rec_dirty = false;
        }
        const PA_GLOBAL int32_t* p = (const PA_GLOBAL int32_t*)job.rec + (size_t)k * 8;
        const PA_GLOBAL int32_t* q = (const PA_GLOBAL int32_t*)job.jh + k;
        const int32_t x = lane < 8 ? p[lane] : (lane == 8 ? q[0] : 0);
        FullRec r;
        r.js = __builtin_amdgcn_readlane(x, 0);
        r.je = __builtin_amdgcn_readlane(x, 1);
        r.ojs = __builtin_amdgcn_readlane(x, 2);
        r.oje = __builtin_amdgcn_readlane(x, 3);
        r.fs = __builtin_amdgcn_readlane(x, 4);
        r.fe = __builtin_amdgcn_readlane(x, 5);
        r.top_val = __builtin_amdgcn_readlane(x, 6);
        r.bot_val = __builtin_amdgcn_readlane(x, 7);
        r.j_h = __builtin_amdgcn_readlane(x, 8);
        r.pad[0] = r.pad[1] = r.pad[2] = 0;
        return r;
    }
    __device__ __forceinline__ void store_rec(int32_t k, const FullRec& r) const {
        // every lane stores the same bytes (see apa2_kernel.hpp: no lane-dependent select chain, no divergent branch)
        PA_GLOBAL pa_i32x4* p = (PA_GLOBAL pa_i32x4*)((PA_GLOBAL int32_t*)job.rec + (size_t)k * 8);
        const pa_i32x4 lo = {r.js, r.je, r.ojs, r.oje}, hi = {r.fs, r.fe, r.top_val, r.bot_val};
        p[0] = lo;
        p[1] = hi;
        ((PA_GLOBAL int32_t*)job.jh)[k] = r.j_h;
        rec_dirty = true;  // (records are read back by a later pass, not by this block)
    }

    // ---- Block::index (block.rs:69-122) -----------------------------------------------------------------------------------------
    // The column of the block computed last stays in registers (lane l: word cw0 + l), when it has at most 64 words: the probes of
    // fixed_j_range and the three look-ups of the next block then cost a masked popcount and a wave sum, no memory round trip.
    mutable int32_t ck = -1, cw0 = 0, cw1 = 0;
    mutable uint64_t cvp = 0, cvm = 0;
    __device__ __forceinline__ void cache_column(int32_t k, int32_t w_from, int32_t w_end) const {
        ck = -1;
        if (w_end - w_from > 64 || w_end <= w_from) return;
        const gcu32 c = (gcu32)slot(k);
        const int32_t wi = w_from + lane;
        cvp = ~0ull;
        cvm = 0ull;
        if (wi < w_end) {
            cvp = (uint64_t)c[(size_t)wi * 4 + 0] | ((uint64_t)c[(size_t)wi * 4 + 1] << 32);
            cvm = (uint64_t)c[(size_t)wi * 4 + 2] | ((uint64_t)c[(size_t)wi * 4 + 3] << 32);
        }
        ck = k;
        cw0 = w_from;
        cw1 = w_end;
    }
    __device__ __forceinline__ int32_t prefix(int32_t k, int32_t w_from, int32_t w_end, int32_t j) const {
        if (k != ck || w_from != cw0 || w_end != cw1) cache_column(k, w_from, w_end);
        if (k == ck) {
            const int32_t full = j >> 6, rem = j & 63;
            const int32_t wi = w_from + lane;
            int32_t acc = 0;
            if (wi < full || (wi == full && rem != 0)) {  // (words at or beyond w_end read as +1 per row: cvp / cvm hold that)
                const uint64_t mask = wi < full ? ~0ull : ((1ull << rem) - 1ull);
                acc = __builtin_popcountll(cvp & mask) - __builtin_popcountll(cvm & mask);
            }
            return wsum(acc);
        }
        const gcu32 c = (gcu32)slot(k);
        const int32_t full = j >> 6, rem = j & 63;
        int32_t acc = 0;
        for (int32_t base = w_from; base <= full; base += 64) {
            const int32_t wi = base + lane;
            if (wi < full || (wi == full && rem != 0)) {
                uint64_t p = ~0ull, mm = 0ull;
                if (wi < w_end) {
                    p = (uint64_t)c[(size_t)wi * 4 + 0] | ((uint64_t)c[(size_t)wi * 4 + 1] << 32);
                    mm = (uint64_t)c[(size_t)wi * 4 + 2] | ((uint64_t)c[(size_t)wi * 4 + 3] << 32);
                }
                const uint64_t mask = wi < full ? ~0ull : ((1ull << rem) - 1ull);
                acc += __builtin_popcountll(p & mask) - __builtin_popcountll(mm & mask);
            }
        }
        return wsum(acc);
    }
    __device__ __forceinline__ int32_t index(int32_t k, const FullRec& r, int32_t j) const {
        if (k == 0) return j;
        if (j > r.je) return r.bot_val + (j - r.je);
        const uint64_t t0 = tick();
        const int32_t v = r.top_val + prefix(k, r.js >> 6, r.je >> 6, j);
        t_index += tick() - t0;
        return v;
    }

    // ---- the left edge of a block (blocks.rs:753-831) -----------------------------------------------------------------------------
    __device__ __forceinline__ void put_word(gu32 dst, gcu32 src, int32_t wi, bool copy) const {
        uint32_t x0 = 0xFFFFFFFFu, x1 = 0xFFFFFFFFu, x2 = 0u, x3 = 0u;
        if (copy) {
            x0 = src[(size_t)wi * 4 + 0];
            x1 = src[(size_t)wi * 4 + 1];
            x2 = src[(size_t)wi * 4 + 2];
            x3 = src[(size_t)wi * 4 + 3];
        }
        dst[(size_t)wi * 4 + 0] = x0;
        dst[(size_t)wi * 4 + 1] = x1;
        dst[(size_t)wi * 4 + 2] = x2;
        dst[(size_t)wi * 4 + 3] = x3;
    }
    // init_v_with_overlap (blocks.rs:753-767) is not a pass over memory here: the block's strip reads its left edge straight from the
    // previous block's column (run_strip<.., TAP> with StripJob::values as the source) -- one store / fence / load round trip less.
    mutable int32_t lazy_k = -1, lazy_pw0 = 0, lazy_pw1 = 0;
    __device__ __forceinline__ void init_plain(int32_t k, const FullRec& prev, const FullRec&) const {
        lazy_k = k;
        lazy_pw0 = k > 1 ? prev.js >> 6 : 0;  // (the first column is all +1)
        lazy_pw1 = k > 1 ? prev.je >> 6 : 0;
    }
    // words [p0, p1) of slot k stay as the older pass left them; [w0, p0) and [p1, min(w1, prev_w1)) come from the previous block
    // (the first column is all +1), the rest is V::one()
    __device__ __forceinline__ void init_preserve(int32_t k, const FullRec&, const FullRec& cur, int32_t p0, int32_t p1, int32_t prev_w1) const {
        const int32_t w0 = cur.js >> 6, w1 = cur.je >> 6;
        const int32_t copy_end = w1 < prev_w1 ? w1 : prev_w1;
        const gu32 dst = slot(k);
        const gcu32 src = (gcu32)slot(k > 0 ? k - 1 : 0);
        for (int32_t wi = w0 + lane; wi < w1; wi += 64) {
            if (wi >= p0 && wi < p1) continue;
            put_word(dst, src, wi, k > 1 && (wi < p0 || wi < copy_end));
        }
        sync_mem();
    }

    // ---- the DP of a block: one range of rows, see apa2_full_logic.hpp `compute2` --------------------------------------------------
    __device__ __forceinline__ int32_t row_sum(int32_t i0, int32_t i1) const {
        const gcu8 hr = (gcu8)job.hrow;
        int32_t acc = 0;
        for (int32_t i = i0 + lane; i < i1; i += 64) {
            const uint32_t b = hr[i];
            acc += (int32_t)(b & 1u) - (int32_t)((b >> 1) & 1u);
        }
        return wsum(acc);
    }
    __device__ __forceinline__ int32_t compute2(int32_t k, int32_t i0, int32_t i1, int32_t w0, int32_t wt, int32_t w1, bool hin, bool tap) const {
        const int32_t words = w1 - w0;
        if (tap && !hin && wt == w0) {  // an empty HMode::Output range: the stored row becomes the +1 row it was given (blocks.rs:443-455)
            for (int32_t i = i0 + lane; i < i1; i += 64) ((gu8)job.hrow)[i] = 1;
            sync_mem();
        }
        if (words <= 0) return hin ? row_sum(i0, i1) : i1 - i0;  // no rows: the bottom row is the top row
        if (win_fail || !sweep::slot_holds(geom(), k, w0, w1)) {
            win_fail = true;
            return 0;
        }
        const bool tap_inside = tap && wt > w0;
        const uint64_t t0 = tick();
        const bool from_prev = lazy_k == k;
        lazy_k = -1;
        int32_t done = 0;
        for (int32_t st = 0; done < words; ++st) {
            const int32_t left = words - done;
            const int32_t kk = left > 32 ? 2 : 1;  // lane = 32 or 64 rows: the tap can sit on any 64-row boundary
            const int32_t take = left < 32 * kk ? left : 32 * kk;
            const bool last = done + take >= words;
            const int32_t sw0 = w0 + done;
            StripJob j;
            j.a_codes = job.a_codes;
            j.b_prof = job.b_prof;
            j.v = (uint32_t*)slot(k);
            j.hin_gran = st > 0 ? job.gran + (size_t)((st - 1) & 1) * 8 : nullptr;
            j.hin_arr = (st == 0 && hin) ? job.hrow : nullptr;
            j.hout_gran = last ? nullptr : job.gran + (size_t)(st & 1) * 8;
            j.hout_arr = job.hrow;  // (TAP: written only when tap_lane >= 0)
            j.values = from_prev ? (uint32_t*)slot(k > 0 ? k - 1 : 0) : nullptr;  // (TAP: the source of the left edge)
            j.sum_out = last ? job.sum : nullptr;
            j.n = i1 - i0;
            j.word0 = sw0;
            j.nlanes = 2 * take;
            j.fill_stride = lazy_pw1;
            j.fill_word0 = lazy_pw0;
            j.exact_tail = last ? 0 : 1;
            j.flags = 0;
            j.col0 = i0;
            j.tail_rows = -1;
            j.k = kk;
            j.ckpt = nullptr;
            j.ckpt_stride = 0;
            j.hin_n = 0;
            j.vsum_out = nullptr;
            // the tap: the deltas leaving the lane whose last row is 64 wt - 1, if that row is in this strip
            int tl = -1;
            if (tap_inside && wt > sw0 && wt <= sw0 + take) tl = kk == 2 ? (wt - sw0) - 1 : 2 * (wt - sw0) - 1;
            if (kk == 1 && rp.enabled && dual_ok(j) && rdv_strip<true>(rdv, wave, rp, j, tl, err, &rdv_cnt, &strip_units, my_prio)) {
                sync_mem();
                ck = -1;
                break;  // (a strip that qualifies is the block's only one)
            }
            if (kk == 2) run_strip<2, false, false, false, true, false, false, true, true>(j, err, 0, tl);
            else if (j.nlanes <= 32) run_strip<1, false, false, false, true, false, true, true, true>(j, err, 0, tl);
            else run_strip<1, false, false, false, true, false, false, true, true>(j, err, 0, tl);
            sync_mem();
            ck = -1;  // (the column changed)
            strip_units += (uint32_t)((((i1 - i0 + 31) >> 5) + ((kk == 1 && j.nlanes <= 32) ? 1 : 2)) * (11 + 12 * kk));
            done += take;
        }
        const int32_t ret = (int32_t)rfl((uint32_t)__hip_atomic_load((const PA_GLOBAL int32_t*)job.sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        t_dp += tick() - t0;
        return ret;
    }

    // ---- the heuristic -------------------------------------------------------------------------------------------------------------
    // x / k for 0 <= x < 2^26 on the scalar unit: k is a launch constant, M = floor(2^31 / k) + 1 (Granlund-Montgomery: exact for
    // dividends below 2^26 and k < 2^5); longer sequences take the hardware's (vector) division.
    uint32_t div_m = 0;
    __device__ __forceinline__ int32_t div_k(int32_t x) const {
        if (div_m != 0u) return (int32_t)(((uint64_t)(uint32_t)x * (uint64_t)div_m) >> 31);
        return (int32_t)rfl((uint32_t)(x / g.k));
    }
    __device__ __forceinline__ int32_t pot(int32_t i) const {  // gd_potential with the quotient in a scalar register
        if (i < 0 || i > g.n) return 0;
        const int32_t before = div_k(i + g.k - 1);
        return before < g.nseeds ? g.nseeds - before : 0;
    }
    __device__ __forceinline__ bool contains_lane(int32_t v, int32_t qx, int32_t qy) const {  // per lane: its own layer v >= 1
        const PA_GLOBAL pa_i32x4* lr = (const PA_GLOBAL pa_i32x4*)g.lrec;
        const PA_GLOBAL pa_i32x4* cl = (const PA_GLOBAL pa_i32x4*)g.cell;
        pa_i32x4 c = lr[v];
        bool hit = c.x >= qx && c.y >= qy;
        int32_t nx = hit ? -1 : c.z;
        for (int32_t guard = g.nmatch; nx >= 0 && guard > 0; --guard) {  // (a list is at most nmatch long)
            c = cl[nx];
            hit = c.x >= qx && c.y >= qy;
            nx = hit ? -1 : c.z;
        }
        return hit;
    }
    // ---- the layer window: lane l keeps the record of layer wb + l in registers (a write-through cache of lrec) ---------------------------
    // The contour build appends layer after layer at the top and asks, match after match, for the score just below it: almost every
    // probe of the build finds the boundary between "holds a point >= q" and "does not" among the 64 layers of the window and never
    // touches memory.  A probe that does not (a match off the alignment's chain) searches memory (score_mem).
    int32_t wb = 0;           // the window's first layer
    bool wvalid = false;
    bool building = false;    // contour build: the window follows the top layers, a miss does not move it
    int32_t wx = 0, wy = 0, wn = -1;  // per lane: lrec[wb + lane]
    uint32_t n_wmiss = 0;

    // The highest layer that holds a point >= (qx, qy) (hint_contours.rs:258-272), searched in memory: lo is known to hold one (layer 0
    // holds everything), hi is known not to (or is past the last layer); 64 layers per round.
    __device__ __forceinline__ int32_t score_mem(int32_t qx, int32_t qy) {
        const int32_t nl = g.nlayers;
        int32_t lo = 0, hi = nl;
        int32_t stride = 1;
        int32_t base = hint - 31;
        if (base > nl - 64) base = nl - 64;
        if (base < 1) base = 1;
        while (hi - lo > 1) {
            n_round += 1;
            const int32_t v = base + lane * stride;
            const bool valid = v > lo && v < hi;
            const bool c = valid && contains_lane(v, qx, qy);
            const uint64_t mv = __ballot(valid), mt = __ballot(c);
            const uint64_t mf = mv & ~mt;
            if (mt) lo = base + (63 - __builtin_clzll(mt)) * stride;
            if (mf) {
                const int32_t f = base + __builtin_ctzll(mf) * stride;
                if (f < hi) hi = f;
            }
            const int32_t span = hi - lo - 1;
            if (span <= 0) break;
            stride = (span + 63) >> 6;
            base = lo + stride;
        }
        return lo;
    }
    // ---- two layer windows for the SEARCH (round 5) -----------------------------------------------------------------------------------
    // The probes of a band search sit at the band's two edges, 50-150 layers apart, and alternate between them; at either edge the
    // score moves by about a layer per seed, a dozen layers per 256-column block.  ONE window was re-centred by every other probe
    // (round 4: slower than none); TWO windows of 64 layers -- lane l holds the record of layer base + l -- hold both edges for
    // several blocks: a probe tests its 64 layers against registers, the least recently used window is re-filled (one load round, what
    // every probe cost before) when neither holds the boundary.  Within a pass the layers do not change (prune_block only marks matches).
    int32_t sb0 = 0, sx0 = 0, sy0 = 0, sn0 = -1, sb1 = 0, sx1 = 0, sy1 = 0, sn1 = -1;
    bool sv0 = false, sv1 = false, s_last0 = false;
    bool search_windows = true;
    uint32_t n_shit = 0;  // diagnostics: probes answered from a window
    // 1: answered (*ans); 0: the boundary is not among the window's layers
    __device__ __forceinline__ int window_probe(int32_t base, int32_t x, int32_t y, int32_t nx0, int32_t qx, int32_t qy, int32_t* ans) const {
        const int32_t v = base + lane;
        const bool valid = v >= 1 && v < g.nlayers;
        bool c = false;
        if (valid) {
            c = x >= qx && y >= qy;
            if (!c && nx0 >= 0) {  // an older point of the layer (rare: a layer off the chain of the alignment)
                const PA_GLOBAL pa_i32x4* cl = (const PA_GLOBAL pa_i32x4*)g.cell;
                int32_t nx = nx0;
                for (int32_t guard = g.nmatch; nx >= 0 && guard > 0; --guard) {
                    const pa_i32x4 e = cl[nx];
                    c = e.x >= qx && e.y >= qy;
                    nx = c ? -1 : e.z;
                }
            }
        }
        const uint64_t mt = __ballot(c);
        if (mt) {
            const int top = 63 - __builtin_clzll(mt);
            if (top < 63 || base + 64 >= g.nlayers) {  // the layer above the highest "yes" says no, or does not exist: the boundary
                *ans = base + top;
                return 1;
            }
        } else if (base <= 1) {  // layer 1 is in the window and says no (or does not exist): only layer 0 is left
            *ans = 0;
            return 1;
        }
        return 0;
    }
    __device__ __forceinline__ void window_fill(bool first, int32_t centre) {
        int32_t base = centre - 31;
        if (base > g.nlayers - 64) base = g.nlayers - 64;
        if (base < 1) base = 1;
        const int32_t v = base + lane;
        pa_i32x4 r = {0, 0, -1, 0};
        if (v < g.nlayers) r = ((const PA_GLOBAL pa_i32x4*)g.lrec)[v];
        n_round += 1;
        if (first) {
            sb0 = base;
            sx0 = r.x;
            sy0 = r.y;
            sn0 = r.z;
            sv0 = true;
        } else {
            sb1 = base;
            sx1 = r.x;
            sy1 = r.y;
            sn1 = r.z;
            sv1 = true;
        }
    }
    __device__ __forceinline__ int32_t score_search(int32_t qx, int32_t qy) {
        int32_t ans = 0;
        if (sv0 && window_probe(sb0, sx0, sy0, sn0, qx, qy, &ans)) {
            s_last0 = true;
            n_shit += 1;
            hint = ans;
            return ans;
        }
        if (sv1 && window_probe(sb1, sx1, sy1, sn1, qx, qy, &ans)) {
            s_last0 = false;
            n_shit += 1;
            hint = ans;
            return ans;
        }
        // neither: the least recently used window moves to the last answer's neighbourhood (one load round) and is asked; a boundary
        // that is not there either is searched in memory (64-ary), and the window is put around it
        const bool use0 = !sv0 || (sv1 && !s_last0);
        window_fill(use0, hint);
        if (window_probe(use0 ? sb0 : sb1, use0 ? sx0 : sx1, use0 ? sy0 : sy1, use0 ? sn0 : sn1, qx, qy, &ans)) {
            s_last0 = use0;
            hint = ans;
            return ans;
        }
        ans = score_mem(qx, qy);
        window_fill(use0, ans);
        s_last0 = use0;
        hint = ans;
        return ans;
    }
    __device__ __forceinline__ int32_t score(int32_t qx, int32_t qy) {
        n_probe += 1;
        if (!building && search_windows) return score_search(qx, qy);
        if (wvalid) {
            const int32_t v = wb + lane;
            const bool valid = v >= 1 && v < g.nlayers;
            bool c = false;
            if (valid) {
                c = wx >= qx && wy >= qy;
                if (!c && wn >= 0) {  // an older point of the layer (rare: a layer off the chain of the alignment)
                    const PA_GLOBAL pa_i32x4* cl = (const PA_GLOBAL pa_i32x4*)g.cell;
                    int32_t nx = wn;
                    for (int32_t guard = g.nmatch; nx >= 0 && guard > 0; --guard) {
                        const pa_i32x4 e = cl[nx];
                        c = e.x >= qx && e.y >= qy;
                        nx = c ? -1 : e.z;
                    }
                }
            }
            const uint64_t mt = __ballot(c);
            if (mt) {
                const int top = 63 - __builtin_clzll(mt);
                // the layer above the highest "yes" is in the window and says no, or does not exist: the boundary
                if (top < 63 || wb + 64 >= g.nlayers) {
                    hint = wb + top;
                    return hint;
                }
            } else if (wb <= 1) {  // layer 1 is in the window and says no (or does not exist): only layer 0 is left
                hint = 0;
                return 0;
            }
        }
        n_wmiss += 1;
        if (building) sync_mem();  // (the records this wavefront stored are what it loads)
        const int32_t ans = score_mem(qx, qy);
        hint = ans;
        return ans;
    }
    __device__ __forceinline__ int32_t h(int32_t i, int32_t j) {
        if (job.heur == kFullHeurGap) {
            const int32_t d = (job.n - i) - (job.m - j);
            return d < 0 ? -d : d;
        }
        if (job.heur == kFullHeurSH) return (int32_t)rfl((uint32_t)((const PA_GLOBAL int32_t*)job.sh_h)[i]);
        if (job.heur != kFullHeurGcsh) return 0;
        const uint64_t t0 = tick();
        const int32_t p = pot(i);
        const int32_t val = score(i - j - p, j - i - p);
        t_h += tick() - t0;
        if (val == 0) {  // csh.rs:178-187, seeds.rs:84-89
            const int32_t d = (g.n - i) - (g.m - j);
            const int32_t gap = d < 0 ? -d : d;
            const int32_t pd = p - pot(g.n);
            return gap > pd ? gap : pd;
        }
        return p - val;
    }

    // Contours from the active matches, last start first (hint_contours.rs:213-255; csh.rs:525-545 reaches the same state).
    // The layer window follows the top: a match of the alignment's chain opens a new layer, which is a register write and one store.
    __device__ __forceinline__ void build_contours() {
        g.nlayers = 1;
        hint = 0;
        dirty = false;
        sv0 = sv1 = false;  // (the layers are derived again: the search's windows hold nothing)
        const uint64_t t0 = tick();
        sync_mem();
        const int32_t ttx = g.n - g.m - pot(g.n), tty = g.m - g.n - pot(g.n);
        const PA_GLOBAL int32_t* mi = (const PA_GLOBAL int32_t*)g.mi;
        const PA_GLOBAL int32_t* mj = (const PA_GLOBAL int32_t*)g.mj;
        const PA_GLOBAL uint8_t* act = (const PA_GLOBAL uint8_t*)g.active;
        PA_GLOBAL pa_i32x4* lr = (PA_GLOBAL pa_i32x4*)g.lrec;
        PA_GLOBAL pa_i32x4* cl = (PA_GLOBAL pa_i32x4*)g.cell;
        wb = 0;
        wx = wy = 0;
        wn = -1;
        wvalid = true;
        building = true;
        for (int32_t top = g.nmatch - 1; top >= 0; top -= 64) {
            const int32_t t = top - lane;  // lane 0 holds the last match of this round
            int32_t sx = 0, sy = 0, ex = 0, ey = 0;
            bool ok = false;
            if (t >= 0) {
                const int32_t i = mi[t], j = mj[t];
                const int32_t ps = gd_potential(g, i), pe = gd_potential(g, i + g.k);
                sx = i - j - ps;
                sy = j - i - ps;
                ex = i - j - pe;  // T(i + k, j + k)
                ey = j - i - pe;
                ok = act[t] != 0 && ex <= ttx && ey <= tty;
            }
            uint64_t mask = __ballot(ok);
            while (mask) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                const int32_t qx = __builtin_amdgcn_readlane(ex, l), qy = __builtin_amdgcn_readlane(ey, l);
                const int32_t px = __builtin_amdgcn_readlane(sx, l), py = __builtin_amdgcn_readlane(sy, l);
                const int32_t v = score(qx, qy) + 1;
                pa_i32x4 rec = {px, py, -1, 0};
                if (v >= wb + 64) {  // (only a new layer at the top can be above the window) the window moves up by half
                    wx = __shfl_down(wx, 32, 64);
                    wy = __shfl_down(wy, 32, 64);
                    wn = __shfl_down(wn, 32, 64);
                    wb += 32;
                }
                if (v < g.nlayers) {  // the layer's previous newest point moves into this match's cell
                    if (v >= wb) {
                        const pa_i32x4 old = {__builtin_amdgcn_readlane(wx, v - wb), __builtin_amdgcn_readlane(wy, v - wb), __builtin_amdgcn_readlane(wn, v - wb), 0};
                        cl[top - l] = old;
                    } else {
                        sync_mem();
                        cl[top - l] = lr[v];
                    }
                    rec.z = top - l;
                } else {
                    g.nlayers = v + 1;
                }
                lr[v] = rec;
                if (v >= wb && lane == v - wb) {
                    wx = rec.x;
                    wy = rec.y;
                    wn = rec.z;
                }
            }
        }
        sync_mem();
        building = false;
        // The probes of the search sit at the band's edges, 50-150 layers below the layers of the diagonal, and alternate between
        // the edges: a 64-layer window would be re-centred by every other probe (measured: slower than no window).  Memory it is.
        wvalid = false;
        t_build += tick() - t0;
    }
    __device__ __forceinline__ void update_contours() {  // csh.rs:497-554, called at the start of a pass (domain.rs:365-371)
        if (job.heur == kFullHeurGcsh && dirty) build_contours();
    }
    // prune.rs:245-292: one lane per seed of the block
    __device__ __forceinline__ void prune_block(int32_t i0, int32_t i1, int32_t j0, int32_t j1) {
        if (job.heur != kFullHeurGcsh || !g.prune) return;
        int32_t s0 = div_k(i0 + g.k);
        int32_t s1 = div_k(i1) + 1;
        if (s0 < 0) s0 = 0;
        if (s1 > g.nseeds) s1 = g.nseeds;
        const uint64_t t0 = tick();
        for (int32_t base = s0; base < s1; base += 64) {
            const int32_t s = base + lane;
            int32_t cnt = 0;
            if (s < s1) cnt = gd_prune_seed(g, s, j0, j1);
            if (__ballot(cnt > 0)) dirty = true;
        }
        // (no fence: the seeds of two blocks are disjoint, and the flags are read by the next contour build, which starts with one)
        t_prune += tick() - t0;
    }
};

__device__ __forceinline__ void store_full_result(const FullJob& job, const FullResult& fr, uint64_t strip_instr, bool win_fail) {
    PairResult res;
    res.status = fr.status == kFullOk ? kOk : (fr.status == kFullErrPasses ? kErrTooManyPasses : (fr.status == kFullErrH0 ? kErrH0 : kErrRangeOrder));
    if (win_fail) res.status = kErrWindow;
    res.cost = fr.cost;
    res.f_max = fr.f_max;
    res.f_max_tries = fr.f_max_tries;
    res.sanity_violations = fr.sanity_violations;
    res.num_blocks = fr.num_blocks;
    res.num_incremental_blocks = fr.num_incremental_blocks;
    res.pad0 = 0;
    res.computed_lanes = fr.computed_lanes;
    res.unique_lanes = fr.unique_lanes;
    res.last_block_idx = fr.last_block_idx;
    res.blocks_len = fr.blocks_len;
    res.strip_instr = strip_instr;
    *job.result = res;  // (every lane stores the same 64 bytes)
}

// Pairs are claimed by ticket in the order of `order` (heaviest first); a block is four independent wavefronts.
// probe_stats (optional, diagnostics): [0] += h probes, [1] += load rounds they took, [2..9) += phase clocks (pa_batch_full_info).
#ifdef PA_UNIT_APA2_FULL  // (the kernel is compiled in a translation unit of its own: csrc/apa2_units.hpp)
#ifndef PA_APA2_FULL_WAVES
#define PA_APA2_FULL_WAVES 5  // wavefronts per SIMD the register allocator is asked for: 96 VGPRs with 2 spilled (4: 104, none).  Round 5: C4 10.57 ->
                              // 10.27 ms, 40 000 pairs 37.6 -> 34.4 ms, 4096 x 100 kbp unchanged (profiles/r05_runs/waves5.log)
#endif
__global__ __launch_bounds__(64 * kStripBlockWaves, PA_APA2_FULL_WAVES) void apa2_full_kernel(const FullJob* __restrict__ jobs, const int32_t* __restrict__ order, int npairs,
                                                                         FullParams sp, uint32_t* ticket, uint32_t* err, uint32_t* dbg,
                                                                         unsigned long long* probe_stats, RdvParams rp, unsigned long long* rdv_stats) {
    const int lane = (int)(threadIdx.x & 63);
    __shared__ RdvShared apa2_rdv;
    rdv_init(&apa2_rdv, kStripBlockWaves);
    const RdvLds rdv_lds{(lds_u32)&apa2_rdv, lane};
    const int wave_in_block = (int)rfl((uint32_t)(threadIdx.x >> 6));
    rdv::Counters rdv_total;
    for (;;) {
        uint32_t t = atomicAdd(ticket, lane == 0 ? 1u : 0u);  // (branch-free: see apa2_kernel.hpp)
        t = rfl(t);
        if (t >= (uint32_t)npairs) break;
        if (rp.prio) PA_SETPRIO_BY_RANK(t, npairs);
        const int pair = (int)rfl((uint32_t)order[t]);
        const FullJob job = jobs[pair];
        FullDevBackend be(job, err, dbg);
        be.timing = probe_stats != nullptr;
        be.rdv = rdv_lds;
        be.wave = wave_in_block;
        be.rp = rp;
        be.my_prio = rp.prio ? PA_PRIO_OF_RANK(t, npairs) : 0;
        be.search_windows = rp.search_windows != 0u;
        const uint64_t t_begin = be.tick();
        FullResult fr{};
        if (job.n > 0 && job.m > 0 && !(job.heur == kFullHeurGcsh && job.g.nmatch < 0)) {  // (nmatch < 0: the matches could not be built on the device)
            if (job.heur == kFullHeurGcsh) be.build_contours();
            PairProgFull<FullDevBackend> prog(be, sp, job.n, job.m);
            prog.run(&fr);
        } else {  // an empty sequence (or no matches): left to the host engine
            fr.status = kFullErrOrder;
        }
        if (rfl(*(const PA_GLOBAL uint32_t*)err) != PA_ERR_NONE && fr.status == kFullOk) fr.status = kFullErrOrder;
        store_full_result(be.job, fr, be.strip_instructions(), be.win_fail);
        if (probe_stats) {
            const unsigned long long vals[10] = {be.n_probe, be.n_round, be.t_build, be.t_dp, be.t_h, be.t_index, be.t_prune, be.t_init, be.tick() - t_begin, be.n_shit};
            for (int q = 0; q < 10; ++q) atomicAdd(probe_stats + q, lane == 0 ? vals[q] : 0ull);
            // per pair: the XCD it ran on and how long its band search took (100 MHz ticks)
            const unsigned long long xcc = (unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);  // HW_REG_XCC_ID[2:0]
            probe_stats[16 + pair] = (xcc << 32) | (unsigned long long)(uint32_t)(be.tick() - t_begin);
        }
        rdv_total.took += be.rdv_cnt.took;
        rdv_total.served += be.rdv_cnt.served;
        rdv_total.alone += be.rdv_cnt.alone;
        rdv_total.withdrawn += be.rdv_cnt.withdrawn;
    }
    rdv_lds.leave();  // (a block of this workgroup that waits for a partner now knows one candidate less)
    if (rdv_stats) {
        const unsigned long long vals[4] = {rdv_total.took, rdv_total.served, rdv_total.alone, rdv_total.withdrawn};
        for (int q = 0; q < 4; ++q) atomicAdd(rdv_stats + q, lane == 0 ? vals[q] : 0ull);
    }
}

// Diagnostics / tests: the device heuristic alone.  One wavefront derives the contours of job 0 and evaluates h at nq positions
// (q[2 t], q[2 t + 1]); out[t] = h, out[nq] = number of layers.  tests/test_gpu_apa2_full.py compares with csrc/gcsh.hpp on the host.
__global__ __launch_bounds__(64) void gcsh_probe_kernel(const FullJob* __restrict__ jobs, const int32_t* __restrict__ q, int nq, int32_t* __restrict__ out, uint32_t* err) {
    const FullJob job = jobs[0];
    FullDevBackend be(job, err, nullptr);
    be.build_contours();
    for (int t = 0; t < nq; ++t) {
        const int32_t i = (int32_t)rfl((uint32_t)q[2 * t]), j = (int32_t)rfl((uint32_t)q[2 * t + 1]);
        const int32_t v = be.h(i, j);
        out[t] = v;
    }
    out[nq] = be.g.nlayers;
}
#endif  // PA_UNIT_APA2_FULL

}  // namespace apa2
}  // namespace pa
