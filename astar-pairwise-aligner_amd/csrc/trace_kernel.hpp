// trace_kernel.hpp -- batched traceback on the GPU: one wavefront per pair walks from (n, m) back to (0, 0).
//
// What it replaces: Blocks::trace / fill_with_blocks / parent of the reference for sparse 256-column blocks without
// DT-trace (astarpa2/src/blocks/trace.rs:21-228, blocks.rs:572-662), i.e. the traceback of
// AstarPa2Params { domain: Full, doubling: None, block_width: 256, front: { sparse: true, dt_trace: false, .. } }.
// The forward pass (pair_kernel<K, CKPT=true>) leaves the right-edge column of every 256-column block in `ckpt`
// (the reference's sparse `Block`s, blocks.rs:322-339).  Per block the wavefront
//   1. re-fills the sub-rectangle (checkpoint column, to.i] x [to.j - height, to.j] rounded out to words, top row +1,
//      height = min(to.j, 5/4 width) doubling until the cost at `to` is reproduced (trace.rs:94-122), with the ordinary
//      strip step (run_strip<1, FILL>) into a per-pair scratch buffer,
//   2. walks `parent` steps (greedy matches, then insertion / deletion / substitution, trace.rs:145-228) on those
//      columns until it crosses the checkpoint column.
// All control flow is wavefront-uniform; the lanes are used for the fill, for prefix sums over a column (Block::index)
// and for 64-at-a-time match extension.  A re-fill taller than one strip (2048 rows: a huge indel inside one block) runs
// as several strips one after the other, like pair_kernel, up to kTraceScratchWords words.  A pair that needs more, or that
// hits a state the reference itself would panic on, is flagged and redone by the host engine.
#pragma once
#include "apa2_logic.hpp"
#include "strip_kernel.hpp"

namespace pa {

struct TraceJob {
    const uint8_t* a;          // ASCII of a (columns) and b (rows), device
    const uint8_t* b;
    const uint32_t* a_codes;   // packed 2-bit codes of a (strip kernel input)
    const uint32_t* b_prof;    // BitProfile words of b, u32 view
    const uint32_t* ckpt;      // checkpoint columns: u32 view of ckpt[c][w] (V each), c = column / 256 (c = 0 unused: V::one)
    const uint32_t* final_v;   // the column at i = n (the forward pass's v), u32 view: the last sparse block
    const int32_t* sum;        // bottom-row sum of the forward pass (cost = 64 w + sum, tail rows already removed)
    uint32_t* cigar;           // out: elements (count << 2) | op, from the END of the alignment to its start
    uint32_t* cigar_len;       // out: number of elements, or kTraceFailed
    int32_t* cost_out;         // out: the edit distance
    uint32_t* scratch_v;       // scratch_words words x 4 u32
    uint32_t* scratch_vals;    // 256 columns x scratch_words words x 4 u32
    int32_t scratch_words;     // tallest re-fill this pair's scratch can hold (min(w, kTraceScratchWords) words)
    uint64_t* scratch_gran;    // 2 rows x 8 granules, zero between uses (multi-strip re-fills hand their bottom row down)
    int32_t n, m, w;           // |a|, |b|, words of b
    uint32_t cigar_cap;
    int32_t dt_max_g, dt_fr_drop;  // DT-trace (trace.rs:231-416) before every re-fill: max_g (0 = off, <= kDtMaxG), fr_drop
    // Banded blocks (the A*PA2 batch of apa2_kernel.hpp): block k's right-edge column covers rows [rec[k].js, rec[k].je) only and
    // starts at rec[k].top_val; `ckpt` is then the pair's column store (slot k = block k, words at their absolute index, `w` words
    // per slot), `final_v` its last slot, and the cost comes from `res`.  nullptr: full-height checkpoints (pair_kernel<K, CKPT>).
    const sweep::BlockRec* rec;
    const apa2::PairResult* res;
    uint32_t* tstats;              // out (optional): TraceStats counters dt_trace_{tries, success, fallback}, fill_{tries, success, fallback}
    int32_t win;                   // banded: words per slot of the column store (sweep_logic.hpp SlotGeom; >= w: full columns)
    uint32_t slot_ratio;           // banded: SlotGeom::ratio
};
enum : uint32_t { kTraceFailed = 0xFFFFFFFFu };
// Re-fills of up to 128 words (8192 rows, four strips) stay on the GPU; a pair with a taller one (an indel of more than
// ~8000 rows inside one 256-column block) is flagged for the host engine instead of sizing every pair's scratch for it.
constexpr int kTraceScratchWords = 128;
enum : uint32_t { kOpMatch = 0, kOpSub = 1, kOpIns = 2, kOpDel = 3 };  // '=', 'X', 'I' (advances b), 'D' (advances a)

__device__ __forceinline__ int32_t wave_sum(int32_t x) { return wave_add(x); }

// Sum of the vertical deltas of the first `rows` rows of a column stored as V words (u32 view, word 0 first); nullptr = V::one.
__device__ __forceinline__ int32_t column_prefix(gcu32 col, int rows, int lane) {
    if (col == nullptr) return rows;
    const int full = rows >> 6, rem = rows & 63;
    int32_t acc = 0;
    for (int base = 0; base <= full; base += 64) {
        const int wi = base + lane;
        if (wi < full || (wi == full && rem != 0)) {
            const uint64_t p = (uint64_t)col[wi * 4 + 0] | ((uint64_t)col[wi * 4 + 1] << 32);
            const uint64_t m = (uint64_t)col[wi * 4 + 2] | ((uint64_t)col[wi * 4 + 3] << 32);
            const uint64_t mask = wi < full ? ~0ull : ((1ull << rem) - 1ull);
            acc += __builtin_popcountll(p & mask) - __builtin_popcountll(m & mask);
        }
    }
    return wave_sum(acc);
}

// The same for a column whose stored words end after `lim` rows: the rows beyond count +1 each (Block::index below the block's
// range, block.rs:75-77).
__device__ __forceinline__ int32_t column_prefix_lim(gcu32 col, int rows, int lim, int lane) {
    if (rows <= lim) return column_prefix(col, rows, lane);
    return column_prefix(col, lim, lane) + (rows - lim);
}

// Vertical delta of row `r` (relative to the column's first row); nullptr = V::one.
__device__ __forceinline__ int32_t column_diff(gcu32 col, int r) {
    if (col == nullptr) return 1;
    const uint32_t p = rfl(col[(r >> 6) * 4 + ((r >> 5) & 1)]);
    const uint32_t m = rfl(col[(r >> 6) * 4 + 2 + ((r >> 5) & 1)]);
    return (int32_t)((p >> (r & 31)) & 1u) - (int32_t)((m >> (r & 31)) & 1u);
}

// ---- DT-trace (blocks/trace.rs:231-416; engine.hpp dt_trace_block) -------------------------------------------------------
// Before a block is re-filled, a diagonal-transition search runs backwards from `to` through the block: level g holds, per
// diagonal d, the smallest column reachable with g edits (then extended left along matches); a diagonal that reaches the
// block's checkpoint column with the right value ends the block.  One lane per diagonal; the furthest-reaching columns of the
// current and the next level, the (extension, parent) table for the walk back, and the block's slices of a and b live in the
// wavefront's LDS slice.  Every decision follows the host code's order: candidates from d - 1, d, d + 1 with strict `<`,
// success at the lowest d, the midpoint and max_g early-outs, fr_drop pruning from both ends.
constexpr int kDtMaxG = 40;
struct DtLds {
    int32_t cur[2 * kDtMaxG + 8], nxt[2 * kDtMaxG + 8];  // column per diagonal, index d + kDtMaxG + 2
    uint16_t tbl[(kDtMaxG + 1) * (kDtMaxG + 1) + 3];       // ext | (parent_d + 1) << 12 at g*g + g + d
    int32_t chain[kDtMaxG + 2];
    uint32_t aw[(8 + 256 + 8) / 4];                        // 8 bytes of headroom, then a[i0 .. st_i)
    uint32_t bw[(8 + 256 + kDtMaxG + 24) / 4];             // 8 bytes of headroom, then b[b_lo .. st_j)
};
constexpr int32_t kDtInf = 0x7FFFFFFF;

// Wavefronts per SIMD asked of the register allocator, per instance (PA_TRACE_WAVES_DT_BANDED overrides the banded DT instance at build time).
#ifndef PA_TRACE_WAVES_DT_BANDED
#define PA_TRACE_WAVES_DT_BANDED 5
#endif
#define PA_TRACE_WAVES(DT_, BANDED_) ((DT_) ? ((BANDED_) ? PA_TRACE_WAVES_DT_BANDED : 5) : 6)
// BANDED: the blocks are the banded blocks of the batched A*PA2 (TraceJob::rec) and TraceStats are counted; the full-height
// checkpoints of the full-DP traced batch (pa_batch_create_trace) compile without either (round 4: with both in one instance the
// full-DP traceback of C4 had gone from 12.2 to 16.6 ms).
// The order the traceback's wavefronts start in: the most expensive pairs first.  A pair's traceback lasts about as long as its
// alignment has edits (DT levels; beyond ~38 edits per block the re-fill and the parent walk on top): in C4 a pair at 15 % takes
// 4.6 ms and a pair at 1 % 0.4 ms, the GPU holds half of the 10 000 wavefronts at a time, and in index order the kernel lasted as long
// as TWO of the slow pairs (9.4 ms for 4.2 ms' worth of work per wavefront slot).  One workgroup sorts the launch's pairs by the cost
// the forward pass found (a counting sort into 1024 buckets, descending); `out` is what trace_kernel walks, the text kernel keeps
// the index order.  The order inside a bucket is whatever the atomics make it: it decides when a pair runs, never what comes out.
__global__ __launch_bounds__(1024) void trace_order_kernel(const TraceJob* __restrict__ jobs, const int32_t* __restrict__ list, int cnt, int32_t* __restrict__ out,
                                                           uint32_t max_cost) {
    __shared__ uint32_t hist[1024], scan[1024];
    const int tid = (int)threadIdx.x;
    auto bucket = [&](int pair) -> uint32_t {
        const TraceJob& t = jobs[pair];
        int64_t c;
        if (t.res) c = t.res->cost;
        else c = t.n == 0 ? t.m : (t.m == 0 ? t.n : 64 * (int64_t)t.w + *t.sum);
        c = c < 0 ? 0 : c;
        const uint64_t b = (uint64_t)c * 1023u / (max_cost ? max_cost : 1u);
        return 1023u - (uint32_t)(b > 1023u ? 1023u : b);
    };
    hist[tid] = 0;
    __syncthreads();
    for (int k = tid; k < cnt; k += 1024) atomicAdd(&hist[bucket(list[k])], 1u);
    __syncthreads();
    uint32_t v = hist[tid];
    scan[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive scan over the buckets
        const uint32_t t = tid >= o ? scan[tid - o] : 0u;
        __syncthreads();
        scan[tid] += t;
        __syncthreads();
    }
    hist[tid] = scan[tid] - v;  // where the bucket starts
    __syncthreads();
    for (int k = tid; k < cnt; k += 1024) {
        const int pair = list[k];
        out[atomicAdd(&hist[bucket(pair)], 1u)] = pair;
    }
}

// -DPA_TRACE_CLOCKS (experiments; tools/trace_clocks.py): where the wavefronts' time goes, summed over a launch (100 MHz ticks).
#ifdef PA_TRACE_CLOCKS
__device__ unsigned long long g_trace_clk[10];  // ticks: DT that succeeded, DT that failed, re-fills, parent steps, whole wavefront; counts: levels, steps, blocks
#define PA_TCLK(...) __VA_ARGS__
#else
#define PA_TCLK(...)
#endif
template <bool DT, bool BANDED>
// (wavefronts per SIMD asked of the register allocator, measured on C4: the DT variant 13.1 ms without the bound, 9.1 / 9.9 / 13.8 ms
//  at 5 / 6 / 7; the re-fill variant 12.6 / 12.1 / 13.1 ms at 5 / 6 / 7)
__global__ __launch_bounds__(64 * kStripBlockWaves, PA_TRACE_WAVES(DT, BANDED)) void trace_kernel(const TraceJob* __restrict__ jobs, const int32_t* __restrict__ list, int npairs, uint32_t* err) {
    // `list`: the pairs of this launch (a chunk of the batch: pa_batch_align runs the chunks on streams of their own, so that the
    // traceback of one overlaps the forward pass of the next and the copy-out of the one before)
    const int slot_ = (int)rfl((uint32_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (slot_ >= npairs) return;
    const int pair = (int)rfl((uint32_t)list[slot_]);
    const int lane = (int)(threadIdx.x & 63);
    TraceJob tj = jobs[pair];
    // (every field gets a scalar live range of its own -- apa2_kernel.hpp own_sgpr: the descriptor arrives in one wide scalar load, and
    //  a spilled tuple is reloaded whole for every later use of one field)
#define PA_OWN(f) tj.f = own_sgpr(tj.f)
    PA_OWN(a); PA_OWN(b); PA_OWN(a_codes); PA_OWN(b_prof); PA_OWN(ckpt); PA_OWN(final_v); PA_OWN(sum); PA_OWN(cigar); PA_OWN(cigar_len); PA_OWN(cost_out);
    PA_OWN(scratch_v); PA_OWN(scratch_vals); PA_OWN(scratch_words); PA_OWN(scratch_gran); PA_OWN(n); PA_OWN(m); PA_OWN(w); PA_OWN(cigar_cap);
    PA_OWN(dt_max_g); PA_OWN(dt_fr_drop); PA_OWN(rec); PA_OWN(res); PA_OWN(tstats); PA_OWN(win); PA_OWN(slot_ratio);
#undef PA_OWN
    extern __shared__ unsigned char pa_trace_lds[];
    DtLds& L = *reinterpret_cast<DtLds*>(pa_trace_lds + (size_t)(threadIdx.x >> 6) * sizeof(DtLds));
    const gcu8 a = (gcu8)tj.a;
    const gcu8 b = (gcu8)tj.b;
    const gu32 cig = (gu32)tj.cigar;
    const int n = tj.n, m = tj.m, w = tj.w;

    constexpr bool banded = BANDED;
    int32_t g;
    bool failed = false;
    if (banded) {
        const PA_GLOBAL apa2::PairResult* rs = (const PA_GLOBAL apa2::PairResult*)tj.res;
        g = (int32_t)rfl((uint32_t)rs->cost);
        failed = rfl((uint32_t)rs->status) != 0u;  // the forward pass handed this pair back to the host engine
    } else {
        g = (n == 0) ? m : (m == 0 ? n : 64 * w + (int32_t)rfl((uint32_t)*(const PA_GLOBAL int32_t*)tj.sum));
    }
    if (lane == 0) *(gi32)tj.cost_out = g;

    PA_TCLK(uint64_t ck_dt_ok = 0; uint64_t ck_dt_fail = 0; uint64_t ck_fill = 0; uint64_t ck_walk = 0; uint32_t ck_levels = 0; uint32_t ck_steps = 0; const uint64_t ck_begin = wall_clock64();)
    uint32_t len = 0, cur_op = 0, cur_cnt = 0;
    uint32_t n_dt_try = 0, n_dt_ok = 0, n_dt_fb = 0, n_fill_try = 0, n_fill_ok = 0, n_fill_fb = 0;  // TraceStats (trace.rs:3-14)
    auto emit = [&](uint32_t op, uint32_t cnt) {
        if (cur_cnt != 0 && cur_op == op) {
            cur_cnt += cnt;
            return;
        }
        if (cur_cnt != 0) {
            if (len < tj.cigar_cap) {
                if (lane == 0) cig[len] = (cur_cnt << 2) | cur_op;
            } else {
                failed = true;
            }
            ++len;
        }
        cur_op = op;
        cur_cnt = cnt;
    };

    int to_i = n, to_j = m;
    // the re-filled rectangle: columns (f_i0, f_i1], rows [f_jlo, f_jhi), top-left value f_T0 (at (f_i0, f_jlo))
    int f_i0 = -1, f_i1 = -1, f_jlo = 0, f_jhi = 0, f_words = 0;
    int32_t f_T0 = 0;
    const gu32 vals = (gu32)tj.scratch_vals;
    const gu32 sv = (gu32)tj.scratch_v;
    gcu32 fcols = (gcu32)vals;  // where the filled columns live
    const sweep::SlotGeom geom{n, m, banded ? tj.win : w, banded ? tj.slot_ratio : 0u};
    auto slot_ptr = [&](int kb) -> gcu32 {  // slot kb of the column store, addressed by absolute word
        return (gcu32)tj.ckpt + ((int64_t)kb * (int64_t)geom.win - (int64_t)(banded ? sweep::slot_off(geom, kb) : 0)) * 4;
    };
    auto ckpt_col = [&](int i0) -> gcu32 {  // the stored column at i0 (a multiple of 256); column 0 is V::one
        return i0 == 0 ? (gcu32) nullptr : slot_ptr(i0 >> 8);
    };
    // rows [js, je) and top value of stored block kb (its right-edge column is at column min(256 kb, n))
    struct BlkMeta {
        int32_t js, je, top;
    };
    auto blk_meta = [&](int kb) -> BlkMeta {
        if (!banded) return BlkMeta{0, 64 * w, kb == 0 ? 0 : (kb * 256 < n ? kb * 256 : n)};
        const PA_GLOBAL int32_t* rp = (const PA_GLOBAL int32_t*)tj.rec + (size_t)kb * 8;
        BlkMeta bm;
        bm.js = (int32_t)rfl((uint32_t)rp[0]);
        bm.je = (int32_t)rfl((uint32_t)rp[1]);
        bm.top = kb == 0 ? 0 : (int32_t)rfl((uint32_t)rp[6]);
        return bm;
    };
    auto filled_col = [&](int i) -> gcu32 { return fcols + (size_t)(i - f_i0 - 1) * (size_t)f_words * 4; };

    while (!failed && (to_i > 0 || to_j > 0)) {
        // (the walk's state is wavefront-uniform; saying so once per step keeps it in scalar registers)
        to_i = (int)rfl((uint32_t)to_i);
        to_j = (int)rfl((uint32_t)to_j);
        g = (int32_t)rfl((uint32_t)g);
        len = rfl(len);
        cur_op = rfl(cur_op);
        cur_cnt = rfl(cur_cnt);
        if (to_i == 0) {  // first column: V::one all the way up (trace.rs parent on Block::first_col)
            emit(kOpIns, (uint32_t)to_j);
            g -= to_j;
            to_j = 0;
            break;
        }
        if (to_j == 0 && !banded) {  // top row: every column costs one deletion (banded: the general path, for the statistics)
            emit(kOpDel, (uint32_t)to_i);
            g -= to_i;
            to_i = 0;
            break;
        }
        // ---- DT-trace through the block, when it has more than one column left (trace.rs:51-75) ----
        if (DT && tj.dt_max_g > 0 && !(f_i0 < to_i && to_i <= f_i1)) {
            const int i0 = ((to_i - 1) >> 8) << 8;
            if (i0 < to_i - 1) {
                const int G = tj.dt_max_g, drop = tj.dt_fr_drop;
                const int st_i = to_i, st_j = to_j, cols = st_i - i0;
                PA_TCLK(const uint64_t ck_t0 = wall_clock64();)
                const gcu32 ck = ckpt_col(i0);
                const BlkMeta cm = blk_meta(i0 >> 8);
                n_dt_try += 1;
                const int b_lo = st_j - cols - G - 1 > 0 ? st_j - cols - G - 1 : 0;
                {
                    uint8_t* awb = reinterpret_cast<uint8_t*>(L.aw) + 8;
                    uint8_t* bwb = reinterpret_cast<uint8_t*>(L.bw) + 8;
                    for (int k = lane; k < cols; k += 64) awb[k] = a[i0 + k];
                    for (int k = lane; k < st_j - b_lo; k += 64) bwb[k] = b[b_lo + k];
                }
                __builtin_amdgcn_wave_barrier();
                // extension to the left along matches (trace.rs:443-500), all lanes at once, eight characters per round: the eight
                // bytes that END at the current character (three aligned LDS words and two byte funnel shifts), most significant byte
                // first.  (Four per round until round 4: a level lasts as long as its longest run, 6-8 rounds of ~0.1 us at 15 %.)
                auto extend = [&](int& i, int& j, bool on) -> int {
                    int cnt = 0;
                    bool go = on && i > i0 && j > 0;
                    for (;;) {
                        const uint64_t gm = __ballot(go);
                        if (gm == 0) break;
                        if ((gm & (gm - 1)) == 0) {
                            // one diagonal is still running (the alignment's own, as a rule): the whole wavefront extends it, 64 characters
                            // per round, one per lane
                            const int l = __builtin_ctzll(gm);
                            int si = __builtin_amdgcn_readlane(i, l), sj = __builtin_amdgcn_readlane(j, l), total = 0;
                            const uint8_t* ab = reinterpret_cast<const uint8_t*>(L.aw) + 8;
                            const uint8_t* bb = reinterpret_cast<const uint8_t*>(L.bw) + 8;
                            for (;;) {
                                const int room = si - i0 < sj ? si - i0 : sj;
                                if (room <= 0) break;
                                const bool eq = lane < room && ab[si - 1 - lane - i0] == bb[sj - 1 - lane - b_lo];
                                const uint64_t em = __ballot(eq);
                                const int run = em == ~0ull ? 64 : __builtin_ctzll(~em);
                                si -= run;
                                sj -= run;
                                total += run;
                                if (run < 64) break;
                            }
                            if (lane == l) {
                                i = si;
                                j = sj;
                                cnt += total;
                            }
                            break;
                        }
                        if (go) {
                            const int qa = i - 1 - i0 + 8 - 7, qb = j - 1 - b_lo + 8 - 7;  // byte offsets of the eight bytes (>= 1: headroom)
                            const uint32_t a0 = L.aw[qa >> 2], a1 = L.aw[(qa >> 2) + 1], a2 = L.aw[(qa >> 2) + 2];
                            const uint32_t b0 = L.bw[qb >> 2], b1 = L.bw[(qb >> 2) + 1], b2 = L.bw[(qb >> 2) + 2];
                            const uint32_t xh = __builtin_amdgcn_alignbyte(a2, a1, (uint32_t)(qa & 3)) ^ __builtin_amdgcn_alignbyte(b2, b1, (uint32_t)(qb & 3));
                            const uint32_t xl = __builtin_amdgcn_alignbyte(a1, a0, (uint32_t)(qa & 3)) ^ __builtin_amdgcn_alignbyte(b1, b0, (uint32_t)(qb & 3));
                            int run = xh ? (__builtin_clz(xh) >> 3) : (xl ? 4 + (__builtin_clz(xl) >> 3) : 8);
                            const int room = i - i0 < j ? i - i0 : j;
                            run = run < room ? run : room;
                            i -= run;
                            j -= run;
                            cnt += run;
                            go = run == 8 && i > i0 && j > 0;
                        }
                    }
                    return cnt;
                };
                // prev_block.index(j) = value at (i0, j): one prefix over the whole column per block (as the re-fill needs for its
                // top-left value), then only the words between
                int vj0 = (st_j - cols - G > 0 ? st_j - cols - G : 0) & ~63;
                vj0 = vj0 > cm.js ? vj0 : cm.js;
                int32_t vbase = 0;
                bool vbase_ok = false;
                auto value_at = [&](int j) -> int32_t {  // prev_block.get(j) (block.rs:126-131); kDtInf: outside the block's rows
                    if (j < cm.js || j > cm.je) return kDtInf;
                    if (ck == nullptr) return cm.top + (j - cm.js);
                    if (!vbase_ok) {
                        vbase = cm.top + column_prefix(ck + (size_t)(cm.js >> 6) * 4, vj0 - cm.js, lane);
                        vbase_ok = true;
                    }
                    return vbase + column_prefix(ck + (size_t)(vj0 >> 6) * 4, j - vj0, lane);
                };
                int found_g = -1, found_d = 0;
                // level 0
                {
                    int i = st_i, j = st_j;
                    const int cnt = extend(i, j, lane == 0);
                    if (lane == 0) {
                        L.cur[G + 2] = i;
                        L.tbl[0] = (uint16_t)(cnt | (1 << 12));
                    }
                    __builtin_amdgcn_wave_barrier();
                    const int i_f = (int)rfl((uint32_t)i), j_f = (int)rfl((uint32_t)j);
                    if (i_f == i0 && j_f >= 0 && value_at(j_f) == g) found_g = 0;
                }
                int lvl = 0, d_lo = 0, d_hi = 0;
                bool dt_fail = false;
                int32_t* cur = L.cur;  // furthest-reaching columns of level lvl / lvl + 1, swapped after every level
                int32_t* nxt = L.nxt;
                while (found_g < 0 && !dt_fail) {
                    // (all of these are wavefront-uniform; saying so keeps the level's loops on the scalar unit)
                    lvl = (int)rfl((uint32_t)lvl);
                    d_lo = (int)rfl((uint32_t)d_lo);
                    d_hi = (int)rfl((uint32_t)d_hi);
                    __builtin_amdgcn_wave_barrier();
                    const int ng = lvl + 1, nlo = d_lo - 1, nhi = d_hi + 1;
                    int32_t min_fr = kDtInf, min_i = kDtInf;
                    int32_t fr_lane = kDtInf;  // this lane's diagonal (nlo + lane) of the level just computed, when it fits one round
                    bool fr_mine = false;
                    for (int base = nlo; base <= nhi && found_g < 0; base += 64) {
                        const int e = base + lane;
                        const bool mine = e <= nhi;
                        // expand (trace.rs:351-364): candidates in the host's order, strict `<` (the three LDS words are asked for together;
                        // one outside the level's range is read and ignored)
                        int32_t bi = kDtInf, bpd = 0;
                        if (mine) {
                            const int32_t c0 = cur[e - 1 + G + 2], c1 = cur[e + G + 2], c2 = cur[e + 1 + G + 2];
                            const bool v0 = e - 1 >= d_lo && e - 1 <= d_hi, v1 = e >= d_lo && e <= d_hi, v2 = e + 1 >= d_lo && e + 1 <= d_hi;
                            if (v0 && c0 < bi) {
                                bi = c0;
                                bpd = -1;
                            }
                            if (v1 && c1 - 1 < bi) {
                                bi = c1 - 1;
                                bpd = 0;
                            }
                            if (v2 && c2 - 1 < bi) {
                                bi = c2 - 1;
                                bpd = 1;
                            }
                        }
                        // extend (trace.rs:370-385)
                        const bool reach = mine && bi < kDtInf - 4 * kDtMaxG;
                        int i = bi, j = reach ? st_j - (st_i - bi) - e : 0;
                        const int cnt = extend(i, j, reach);
                        fr_lane = reach ? i : bi;
                        fr_mine = mine;
                        if (mine) {
                            nxt[e + G + 2] = fr_lane;
                            L.tbl[ng * ng + ng + e] = (uint16_t)((reach ? cnt : 0) | ((bpd + 1) << 12));
                        }
                        // a diagonal at the checkpoint column with the right value ends the block: the lowest d first
                        uint64_t cand = __ballot(reach && i == i0 && j >= 0);
                        while (cand) {
                            const int l = __builtin_ctzll(cand);
                            cand &= cand - 1;
                            const int jl = __builtin_amdgcn_readlane(j, l);
                            if (value_at(jl) == g - ng) {
                                found_g = ng;
                                found_d = base + l;
                                break;
                            }
                        }
                        if (reach) {
                            min_fr = 2 * i - e < min_fr ? 2 * i - e : min_fr;
                            min_i = i < min_i ? i : min_i;
                        }
                    }
                    if (found_g >= 0) break;
                    lvl = ng;
                    d_lo = nlo;
                    d_hi = nhi;
                    {
                        int32_t* t = cur;
                        cur = nxt;
                        nxt = t;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lvl == G / 2) {  // trace.rs:388-391
                        min_i = wave_min(min_i);
                        if (min_i > (i0 + st_i) / 2) dt_fail = true;
                    }
                    if (lvl == G) dt_fail = true;
                    if (!dt_fail && drop > 0) min_fr = wave_min(min_fr);
                    if (!dt_fail && drop > 0) {  // trace.rs:396-413
                        if (d_hi - d_lo < 64) {
                            // the level fitted one round: every lane judges its own diagonal and the two loops of the reference become the
                            // first and the last set bit of one ballot (all bad: d_lo runs up to d_hi and the second loop never starts)
                            const int d = d_lo + lane;
                            const bool keep = fr_mine && !(fr_lane <= i0 || (int64_t)2 * fr_lane - d > (int64_t)min_fr + drop);
                            const uint64_t km = __ballot(keep);
                            if (km == 0) {
                                d_lo = d_hi;
                            } else {
                                d_hi = d_lo + 63 - __builtin_clzll(km);
                                d_lo = d_lo + __builtin_ctzll(km);
                            }
                        } else {
                            auto bad = [&](int d) -> bool {
                                const int32_t fi = (int32_t)rfl((uint32_t)cur[d + G + 2]);  // (uniform, and the compiler must know it: d_lo / d_hi steer every loop of the level)
                                return fi <= i0 || (int64_t)2 * fi - d > (int64_t)min_fr + drop;
                            };
                            while (d_lo < d_hi && bad(d_lo)) d_lo += 1;  // (uniform: every lane reads the same LDS words)
                            while (d_lo < d_hi && bad(d_hi)) d_hi -= 1;
                        }
                        if (d_lo > d_hi) dt_fail = true;
                    }
                }
                if (found_g >= 0) {
                    // walk back through the table (trace.rs:274-314); the elements go out from the END of the alignment
                    // (every word read back from LDS is wavefront-uniform; readfirstlane says so, and the run-length state of `emit` stays scalar)
                    int dk = found_d;
                    for (int k = found_g; k >= 0; --k) {
                        if (lane == 0) L.chain[k] = dk;
                        dk += (int)(rfl((uint32_t)L.tbl[k * k + k + dk]) >> 12) - 1;
                    }
                    __builtin_amdgcn_wave_barrier();
                    int d = (int)rfl((uint32_t)L.chain[0]);
                    for (int k = 0; k <= found_g; ++k) {
                        const uint32_t ext = rfl((uint32_t)L.tbl[k * k + k + d]) & 0xFFFu;
                        if (ext > 0) emit(kOpMatch, ext);
                        if (k < found_g) {
                            const int dn = (int)rfl((uint32_t)L.chain[k + 1]);
                            const int pd = (int)(rfl((uint32_t)L.tbl[(k + 1) * (k + 1) + (k + 1) + dn]) >> 12) - 1;
                            emit(pd == -1 ? kOpIns : (pd == 0 ? kOpSub : kOpDel), 1);
                            d = dn;
                        }
                    }
                    g -= found_g;
                    to_i = i0;
                    to_j = st_j - cols - found_d;
                    n_dt_ok += 1;
                    PA_TCLK(ck_dt_ok += wall_clock64() - ck_t0; ck_levels += (uint32_t)found_g;)
                    continue;
                }
                n_dt_fb += 1;
                PA_TCLK(ck_dt_fail += wall_clock64() - ck_t0; ck_levels += (uint32_t)lvl;)
            }
        }
        // ---- re-fill when the walk has left the filled columns (trace.rs:83-125) ----
        if (!(f_i0 < to_i && to_i <= f_i1) && to_i == n && ((n - 1) & 255) == 0) {
            // the last sparse block is a single column next to a checkpoint: the reference walks it as stored, without a
            // re-fill (trace.rs:86: neither `prev.e < to.i - 1` nor `block.e > to.i`)
            const BlkMeta lm = blk_meta((n + 255) >> 8);
            f_i0 = n - 1;
            f_i1 = n;
            f_jlo = lm.js;
            f_jhi = lm.je;
            f_words = (lm.je - lm.js) >> 6;
            f_T0 = lm.top - 1;
            fcols = (banded ? slot_ptr((n + 255) >> 8) : (gcu32)tj.final_v) + (size_t)(lm.js >> 6) * 4;
        }
        if (!(f_i0 < to_i && to_i <= f_i1)) {
            PA_TCLK(const uint64_t ck_t1 = wall_clock64();)
            fcols = (gcu32)vals;
            const int i0 = ((to_i - 1) >> 8) << 8;
            const int cols = to_i - i0;
            const gcu32 ck = ckpt_col(i0);
            const BlkMeta cm = blk_meta(i0 >> 8);          // the stored block left of the rectangle (prev_block)
            const int blk_js = blk_meta((i0 >> 8) + 1).js;  // first row of the block the walk is in (trace.rs:93)
            if (to_j < blk_js) {  // (jr would be empty: not a state the reference reaches)
                failed = true;
                break;
            }
            int height = to_j - blk_js < cols * 5 / 4 ? to_j - blk_js : cols * 5 / 4;
            for (;;) {
                const int jlo_raw = to_j - height > cm.js ? to_j - height : cm.js;
                const int jlo = jlo_raw & ~63, jhi = (to_j + 63) & ~63;
                const int words = (jhi - jlo) >> 6;
                if (words > tj.scratch_words) {  // taller than this pair's scratch: leave it to the host engine
                    failed = true;
                    break;
                }
                n_fill_try += 1;
                // left column = the checkpoint's words of these rows, V::one outside its range (init_v_with_overlap, blocks.rs:753-767)
                for (int wi = lane; wi < words; wi += 64) {
                    const int aw = (jlo >> 6) + wi;
                    const bool inr = ck != nullptr && aw >= (cm.js >> 6) && aw < (cm.je >> 6);
#pragma unroll
                    for (int c = 0; c < 4; ++c) sv[wi * 4 + c] = inr ? ck[aw * 4 + c] : (c < 2 ? 0xFFFFFFFFu : 0u);
                }
                const int32_t T0 = cm.top + (ck ? column_prefix_lim(ck + (size_t)(cm.js >> 6) * 4, jlo - cm.js, cm.je - cm.js, lane) : jlo - cm.js);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // same wavefront produces and consumes: ordering only, no L2 write-back
                const int S = (words + 31) >> 5;  // strips of 32 words, top to bottom
                for (int st = 0; st < S; ++st) {
                    StripJob j;
                    j.a_codes = tj.a_codes;
                    j.b_prof = tj.b_prof;
                    j.v = (uint32_t*)tj.scratch_v - (size_t)(jlo >> 6) * 4;
                    j.hin_gran = st > 0 ? tj.scratch_gran + (size_t)((st - 1) & 1) * 8 : nullptr;
                    j.hin_arr = nullptr;
                    j.hout_gran = st + 1 < S ? tj.scratch_gran + (size_t)(st & 1) * 8 : nullptr;
                    j.hout_arr = nullptr;
                    j.values = tj.scratch_vals;
                    j.sum_out = nullptr;
                    j.n = cols;
                    j.word0 = (jlo >> 6) + 32 * st;
                    j.nlanes = 2 * (words - 32 * st < 32 ? words - 32 * st : 32);
                    j.fill_stride = words;
                    j.fill_word0 = 32 * st;
                    j.exact_tail = 1;
                    j.flags = 0;
                    j.col0 = i0;
                    j.tail_rows = -1;
                    j.k = 1;
                    j.ckpt = nullptr;
                    j.ckpt_stride = 0;
                    j.hin_n = 0;
                    j.vsum_out = nullptr;
                    run_strip<1, true, false, false, true>(j, err);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
                f_i0 = i0;
                f_i1 = to_i;
                f_jlo = jlo;
                f_jhi = jhi;
                f_words = words;
                f_T0 = T0;
                const int32_t val = T0 + cols + column_prefix(filled_col(to_i), to_j - jlo, lane);
                if (val == g) {
                    n_fill_ok += 1;
                    break;
                }
                n_fill_fb += 1;
                if (jlo == 0 || height == 0 || to_j - height <= cm.js) {  // "No trace found through block" (the others: the reference would loop)
                    failed = true;
                    break;
                }
                height *= 2;
            }
            if (failed) break;
            PA_TCLK(ck_fill += wall_clock64() - ck_t1;)
        }
        // ---- parent (trace.rs:145-228) ----
        PA_TCLK(ck_steps += 1;)
        // Every address a step may need depends only on `to`, so all loads are issued together (one memory round trip per
        // step instead of up to four dependent ones); the decisions below then follow the reference's order.
        const int r = to_j - 1;
        const int lim = to_i < to_j ? (to_i < 64 ? to_i : 64) : (to_j < 64 ? to_j : 64);
        uint8_t ca = 0, cb = 1;
        if (lane < lim) {
            ca = a[to_i - 1 - lane];
            cb = b[to_j - 1 - lane];
        }
        const bool in_rows = r >= f_jlo && r < f_jhi;
        uint32_t cur_p = 0, cur_m = 0;  // the dwords of the current column that hold row r
        if (in_rows) {
            const gcu32 cc = filled_col(to_i);
            const int rr = r - f_jlo;
            cur_p = cc[(rr >> 6) * 4 + ((rr >> 5) & 1)];
            cur_m = cc[(rr >> 6) * 4 + 2 + ((rr >> 5) & 1)];
        }
        // previous column: the checkpoint itself when to_i - 1 == f_i0, else a filled column
        const bool prev_ck = (to_i - 1 == f_i0);
        BlkMeta pm{f_jlo, f_jhi, f_T0 + (to_i - 1 - f_i0)};
        if (prev_ck) pm = blk_meta(((f_i0 + 255) >> 8));
        const gcu32 pc_raw = prev_ck ? ckpt_col(f_i0) : filled_col(to_i - 1);
        const gcu32 pc = (prev_ck && pc_raw) ? pc_raw + (size_t)(pm.js >> 6) * 4 : pc_raw;  // first stored row = pm.js
        const int p_jlo = pm.js;
        const int32_t p_top = pm.top;
        const bool p_idx = to_j >= p_jlo;  // Block::index is defined there
        int32_t p_part = 0;                 // this lane's share of the prefix sum of the previous column up to row to_j
        uint32_t prev_p = 0, prev_m = 0;    // the dwords of the previous column that hold row r
        const bool p_wide = !prev_ck && f_words > 64;  // a multi-strip re-fill: use the general column sum below
        if (!prev_ck) {                     // a filled column of up to 64 words: one word per lane
            if (p_idx && !p_wide) {
                const int rows = to_j - p_jlo, full = rows >> 6, rem = rows & 63;
                if (lane < full || (lane == full && rem != 0)) {
                    const uint64_t p = (uint64_t)pc[lane * 4 + 0] | ((uint64_t)pc[lane * 4 + 1] << 32);
                    const uint64_t m = (uint64_t)pc[lane * 4 + 2] | ((uint64_t)pc[lane * 4 + 3] << 32);
                    const uint64_t mask = lane < full ? ~0ull : ((1ull << rem) - 1ull);
                    p_part = __builtin_popcountll(p & mask) - __builtin_popcountll(m & mask);
                }
            }
            if (r >= p_jlo) {
                const int rr = r - p_jlo;
                prev_p = pc[(rr >> 6) * 4 + ((rr >> 5) & 1)];
                prev_m = pc[(rr >> 6) * 4 + 2 + ((rr >> 5) & 1)];
            }
        }
        // ---- decisions ----
        {  // greedy matches, 64 characters at a time
            uint32_t cnt = 0;
            uint64_t mask = __ballot(lane < lim && ca == cb);
            for (;;) {
                const int run = mask == ~0ull ? 64 : __builtin_ctzll(~mask);
                cnt += (uint32_t)run;
                to_i -= run;
                to_j -= run;
                if (run < 64 || to_i == 0 || to_j == 0) break;
                const int lim2 = to_i < to_j ? (to_i < 64 ? to_i : 64) : (to_j < 64 ? to_j : 64);
                bool eq = false;
                if (lane < lim2) eq = a[to_i - 1 - lane] == b[to_j - 1 - lane];
                mask = __ballot(eq);
            }
            if (cnt > 0) {
                emit(kOpMatch, cnt);
                continue;
            }
        }
        // vertical delta of the current column at row to_j - 1 (Block::get_diff)
        if (in_rows) {
            const uint32_t p = rfl(cur_p), m = rfl(cur_m);
            if ((int32_t)((p >> (r & 31)) & 1u) - (int32_t)((m >> (r & 31)) & 1u) == 1) {
                g -= 1;
                to_j -= 1;
                emit(kOpIns, 1);
                continue;
            }
        }
        int32_t hd = 1;
        if (p_idx) hd = g - (p_top + (prev_ck ? column_prefix_lim(pc, to_j - p_jlo, pm.je - p_jlo, lane) : (p_wide ? column_prefix(pc, to_j - p_jlo, lane) : wave_sum(p_part))));
        if (hd == 1) {
            g -= 1;
            to_i -= 1;
            emit(kOpDel, 1);
            continue;
        }
        if (r < p_jlo) {  // the reference's get_diff(..).unwrap() would panic here
            failed = true;
            break;
        }
        int32_t pd;
        if (prev_ck && to_j > pm.je) {  // trace.rs:210-215: one row below the previous block's range
            if (to_j != pm.je + 1) {
                failed = true;
                break;
            }
            pd = 1 - hd;  // dd = 1
        } else if (prev_ck) {
            pd = column_diff(pc, r - p_jlo);
        } else {
            const uint32_t p = rfl(prev_p), m = rfl(prev_m);
            pd = (int32_t)((p >> (r & 31)) & 1u) - (int32_t)((m >> (r & 31)) & 1u);
        }
        const int32_t dd = pd + hd;
        if (dd == 1) {
            g -= 1;
            to_i -= 1;
            to_j -= 1;
            emit(kOpSub, 1);
            continue;
        }
        failed = true;  // "PARENT NOT FOUND IN TRACEBACK"
    }
    if (!failed) {
        if (cur_cnt != 0) {
            if (len < tj.cigar_cap) {
                if (lane == 0) cig[len] = (cur_cnt << 2) | cur_op;
            } else {
                failed = true;
            }
            ++len;
        }
        if (g != 0) failed = true;  // "trace ends at distance 0"
    }
    if (lane == 0) *(gu32)tj.cigar_len = failed ? kTraceFailed : len;
    PA_TCLK(if (lane == 0) {
        atomicAdd(&g_trace_clk[0], (unsigned long long)ck_dt_ok); atomicAdd(&g_trace_clk[1], (unsigned long long)ck_dt_fail);
        const uint64_t ck_all = wall_clock64() - ck_begin;
        ck_walk = ck_all - ck_dt_ok - ck_dt_fail - ck_fill;  // parent steps and everything else
        atomicAdd(&g_trace_clk[2], (unsigned long long)ck_fill); atomicAdd(&g_trace_clk[3], (unsigned long long)ck_walk);
        atomicAdd(&g_trace_clk[4], (unsigned long long)ck_all); atomicAdd(&g_trace_clk[5], (unsigned long long)ck_levels);
        atomicAdd(&g_trace_clk[6], (unsigned long long)ck_steps); atomicAdd(&g_trace_clk[7], (unsigned long long)(n_dt_try + n_fill_try));
    })
    if (BANDED && tj.tstats && lane == 0) {
        gu32 ts = (gu32)tj.tstats;
        ts[0] = n_dt_try;
        ts[1] = n_dt_ok;
        ts[2] = n_dt_fb;
        ts[3] = n_fill_try;
        ts[4] = n_fill_ok;
        ts[5] = n_fill_fb;
    }
}

}  // namespace pa
