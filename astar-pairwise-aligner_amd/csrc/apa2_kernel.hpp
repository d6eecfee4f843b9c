// apa2_kernel.hpp -- the gfx950 backend of apa2_logic.hpp: ONE WAVEFRONT runs A*PA2's whole band search for one pair.
//
// What it replaces: the loop `for (a, b) in pairs { aligner.align(a, b) }` of pa-bin (pa-bin/src/main.rs:24-35) over
// AstarPa2Params::simple() and its relatives (astarpa2/src/lib.rs:122-175, band.rs:100-182, domain.rs:356-541,
// blocks.rs:205-340), for many pairs at once.
//
// MI355X-first shape:
//  * A persistent grid: every wavefront pulls the next pair from an atomic ticket (heaviest pairs first), runs every pass of
//    its band search -- j_range, the DP of each 256-column block, fixed_j_range, the reuse test, the doubling of the bound --
//    and writes the pair's result.  No host round trip per block, per pass or per pair; no wavefront ever waits for another.
//  * The DP of a block is the strip step of strip_kernel.hpp (lane = 32 rows, anti-diagonal skew inside the wave); a band taller
//    than 2048 rows runs as several strips top to bottom, the bottom row handed down through two granule rows that stay in
//    the L2 (produced and consumed by the same wavefront).  Bands of at most 1024 rows use the half-wave strip.
//  * Block::index is a wave-parallel popcount prefix sum over the stored column; the two probing loops of fixed_j_range are
//    exact wave-parallel searches (see apa2_logic.hpp for why they end where the reference's jumping probes end).
//  * The right-edge column of every block stays in HBM at its absolute word position (slot k of the pair's column store), so
//    the traceback (trace_kernel.hpp) reads the blocks of the successful pass where the forward pass left them.
//  * Two pairs per strip (round 5): a block of at most 16 words -- half a wave -- meets a block of another wavefront of the same
//    workgroup (rdv_logic.hpp) and the two run as ONE strip, pair A in lanes 0..31 and pair B in lanes 32..63 (strip2_kernel.hpp):
//    25 instead of 24 + 24 VALU instructions per step for what is 80 % of this kernel's instructions on short pairs.
#pragma once
#include "apa2_jobs.hpp"
#include "apa2_logic.hpp"
#include "strip2_kernel.hpp"
#include "strip_kernel.hpp"

namespace pa {
namespace apa2 {


__device__ __forceinline__ int32_t wsum(int32_t x) { return wave_add(x); }

// A descriptor fetched with one wide scalar load lives in ONE register tuple: when the register allocator spills it (and it must: the
// strips need every scalar register), any later use of ONE field reloads all sixteen -- v_readlane after v_readlane on the vector
// unit.  Passing every field through an empty asm gives each its own live range (round 4; seen in the ISA of apa2_full_kernel).
using pa::own_sgpr;  // (strip_kernel.hpp)

struct DevBackend {
    const PairJob& job;
    HeurParams hp;
    uint32_t* err;
    uint32_t* dbg;  // diagnostics: host-mapped progress markers, or nullptr
    int lane;
    bool k1_only = false;  // experiments: K = 1 strips only (PA_APA2_K1)
    uint32_t lds_eq = 0;   // byte offset of this wavefront's 4 KB LDS slice for the eq words of K = 4 strips (strip_kernel.hpp LdsEq)
    mutable uint32_t strip_units = 0;  // modelled VALU instructions of the strips so far, in units of 32 (one per unrolled chunk step)
    // the rendezvous of half-wave blocks (strip2_kernel.hpp): this workgroup's shared word and mail, this wavefront's index in it
    RdvLds rdv{nullptr, 0};
    int wave = 0;
    RdvParams rp{0u, 0u, 0u, 0u};
    mutable rdv::Counters rdv_cnt;
    int32_t my_prio = 0;  // this wavefront's issue priority while it runs this pair (RdvParams::prio)
    __device__ __forceinline__ uint64_t strip_instructions() const { return (uint64_t)strip_units << 5; }

    // (own_sgpr on the descriptor's fields was tried here too, round 4: C4 forward 11.4 -> 11.7 ms -- this kernel's band logic is short
    //  enough that the tuple reloads do not matter, and the extra live ranges cost more than they save)
    __device__ __forceinline__ DevBackend(const PairJob& j, const HeurParams& h, uint32_t* e, uint32_t* d) : job(j), hp(h), err(e), dbg(d) { lane = (int)(threadIdx.x & 63); }
    __device__ __forceinline__ void mark(int slot_, uint32_t value) const {
        if (dbg && lane == 0) __hip_atomic_store(dbg + slot_, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }

    __device__ __forceinline__ int32_t uniform(int32_t x) const { return (int32_t)rfl((uint32_t)x); }  // back to a scalar register
    mutable bool win_fail = false;  // a block left the window of the column store
    // (a block that left the window ends the pass at once, like a strip error: the records, scans and prefix sums that would follow
    //  address words outside the slot -- other slots, other pairs, past the end of the store; apa2_full_kernel.hpp does the same)
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

    __device__ __forceinline__ BlockRec load_rec(int32_t k) const {
        const PA_GLOBAL int32_t* p = (const PA_GLOBAL int32_t*)job.rec + (size_t)k * 8;
        const int32_t x = lane < 8 ? p[lane] : 0;
        BlockRec r;
        r.js = __builtin_amdgcn_readlane(x, 0);
        r.je = __builtin_amdgcn_readlane(x, 1);
        r.ojs = __builtin_amdgcn_readlane(x, 2);
        r.oje = __builtin_amdgcn_readlane(x, 3);
        r.fs = __builtin_amdgcn_readlane(x, 4);
        r.fe = __builtin_amdgcn_readlane(x, 5);
        r.top_val = __builtin_amdgcn_readlane(x, 6);
        r.bot_val = __builtin_amdgcn_readlane(x, 7);
        return r;
    }
    __device__ __forceinline__ void store_rec(int32_t k, const BlockRec& r) const {
        // every lane stores the same 32 bytes (one write per instruction): no lane-dependent select chain -- that one made the compiler
        // keep the records in a stack array indexed by the lane, and whatever is loaded from there counts as divergent, which put
        // the whole band logic on the vector unit
        typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
        PA_GLOBAL i32x4* p = (PA_GLOBAL i32x4*)((PA_GLOBAL int32_t*)job.rec + (size_t)k * 8);
        const i32x4 lo = {r.js, r.je, r.ojs, r.oje}, hi = {r.fs, r.fe, r.top_val, r.bot_val};
        p[0] = lo;
        p[1] = hi;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the same wavefront reads it back (ordering only)
    }

    // Sum of the vertical deltas of rows [64 * w_from, j) of column slot k; words at or beyond w_end count +1 per row.
    __device__ __forceinline__ int32_t prefix(int32_t k, int32_t w_from, int32_t w_end, int32_t j) const {
        const gcu32 c = (gcu32)slot(k);
        const int32_t full = j >> 6, rem = j & 63;  // whole words [w_from, full), then `rem` rows of word `full`
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

    // Block::index (block.rs:69-122; from the top: the column is consistent, top_val + all deltas == bot_val)
    __device__ __forceinline__ int32_t index(int32_t k, const BlockRec& r, int32_t j) const {
        if (k == 0) return j;  // the first column: V::one from (0, 0)
        if (j > r.je) return r.bot_val + (j - r.je);
        return r.top_val + prefix(k, r.js >> 6, r.je >> 6, j);
    }

    // init_v_with_overlap (blocks.rs:753-767) + compute_block(HMode::None) (blocks.rs:686-748) for block k; the bottom-row sum.
    __device__ __forceinline__ int32_t compute(int32_t k, const BlockRec& prev, const BlockRec& cur, int32_t i0, int32_t i1) const {
        const int32_t w0 = cur.js >> 6, w1 = cur.je >> 6, words = w1 - w0;
        if (words <= 0) return i1 - i0;  // no rows: the bottom row is the top row (+1 per column)
        if (!sweep::slot_holds(geom(), k, w0, w1)) {
            win_fail = true;
            return 0;
        }
        const int32_t pw0 = prev.js >> 6, pw1 = prev.je >> 6;
        const gu32 dst = slot(k);
        const gcu32 src = (gcu32)slot(k - 1);
        for (int32_t wi = w0 + lane; wi < w1; wi += 64) {
            uint32_t x0 = 0xFFFFFFFFu, x1 = 0xFFFFFFFFu, x2 = 0u, x3 = 0u;
            if (k > 1 && wi >= pw0 && wi < pw1) {
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
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        // Strips top to bottom.  Height by what is left of the band: up to 32 words one K = 1 strip (half a wave for up to 16), up to
        // 64 words K = 2, up to 96 words K = 3, beyond that K = 4 strips of 128 words -- 23 / 35 / 47 / 59 VALU instructions per step,
        // i.e. 23 / 17.5 / 15.7 / 14.75 per 2048 cells of a FULL strip; a strip pays its 64 steps of skew only once (strip_kernel.hpp).
        // (K = 3 is there for the 65..96-word bands of 100 kbp pairs, which a K = 4 strip would run three quarters empty.)
        int32_t done = 0;
        for (int32_t st = 0; done < words; ++st) {
            mark(6, (uint32_t)st);
            const int32_t left = words - done;
            const int32_t kk = k1_only ? 1 : (left > 96 ? 4 : (left > 64 ? 3 : (left > 32 ? 2 : 1)));
            const int32_t take = left < 32 * kk ? left : 32 * kk;
            const bool last = done + take >= words;
            StripJob j;
            j.a_codes = job.a_codes;
            j.b_prof = job.b_prof;
            j.v = (uint32_t*)dst;
            j.hin_gran = st > 0 ? job.gran + (size_t)((st - 1) & 1) * 8 : nullptr;
            j.hin_arr = nullptr;
            j.hout_gran = last ? nullptr : job.gran + (size_t)(st & 1) * 8;
            j.hout_arr = nullptr;
            j.values = nullptr;
            j.sum_out = last ? job.sum : nullptr;
            j.n = i1 - i0;
            j.word0 = w0 + done;
            j.nlanes = 2 * take;
            j.fill_stride = 0;
            j.fill_word0 = 0;
            j.exact_tail = last ? 0 : 1;
            j.flags = 0;
            j.col0 = i0;
            j.tail_rows = -1;
            j.k = kk;
            j.ckpt = nullptr;
            j.ckpt_stride = 0;
            j.hin_n = 0;
            j.vsum_out = nullptr;
            // (a strip that is not the last one is full, the last one does not need an exact bottom row: NOPASS)
            if (kk == 1 && rp.enabled && dual_ok(j) && rdv_strip<false>(rdv, wave, rp, j, -1, err, &rdv_cnt, &strip_units, my_prio)) {
                // (this block and a block of another wavefront ran as one strip -- or a partner ran it: the column and the sum are in memory)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                break;
            }
            if (kk == 4) run_strip<4, false, false, false, true, true, false, true>(j, err, lds_eq);  // eq words from LDS: 50 instead of 59 per step
            else if (kk == 3) run_strip<3, false, false, false, true, false, false, true>(j, err);
            else if (kk == 2) run_strip<2, false, false, false, true, false, false, true>(j, err);
            else if (j.nlanes <= 32) run_strip<1, false, false, false, true, false, true, true>(j, err);
            else run_strip<1, false, false, false, true, false, false, true>(j, err);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            // (run_strip: ceil(n / 32) + 2 chunks of 32 steps, one chunk less for the half-wave strip; 11 + 12 K instructions per step)
            strip_units += (uint32_t)((((i1 - i0 + 31) >> 5) + ((kk == 1 && j.nlanes <= 32) ? 1 : 2)) * (kk == 4 ? 50 : 11 + 12 * kk));
            done += take;
        }
        return (int32_t)rfl((uint32_t)__hip_atomic_load((const PA_GLOBAL int32_t*)job.sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }

    __device__ __forceinline__ int32_t hval(int32_t i, int32_t j, int32_t sh_i) const {
        if (hp.kind == sweep::kHeurGap) {
            const int32_t d = (hp.n - i) - (hp.m - j);
            return d < 0 ? -d : d;
        }
        return hp.kind == sweep::kHeurSH ? sh_i : 0;
    }

    // First (LAST = false) / last (LAST = true) row j in [lo, hi] with index(j) + h(i, j) <= f_max.
    template <bool LAST>
    __device__ __forceinline__ bool scan(int32_t k, const BlockRec& r, int32_t i, int32_t f_max, int32_t lo, int32_t hi, int32_t* out) const {
        const gcu32 c = (gcu32)slot(k);
        const int32_t w_end = r.je >> 6;
        const int32_t wlo = lo >> 6, whi = hi >> 6;
        const int32_t sh_i = hp.kind == sweep::kHeurSH ? (int32_t)rfl((uint32_t)((const PA_GLOBAL int32_t*)hp.sh_h)[i]) : 0;
        int32_t base = index(k, r, wlo << 6);  // value at the first row of word wlo
        bool found = false;
        for (int32_t wc = wlo; wc <= whi; wc += 64) {
            const int32_t wi = wc + lane;
            const bool valid = wi <= whi;
            uint64_t p = ~0ull, mm = 0ull;  // rows at or beyond the block's end: +1 each (block.rs:75-77)
            if (valid && wi < w_end) {
                p = (uint64_t)c[(size_t)wi * 4 + 0] | ((uint64_t)c[(size_t)wi * 4 + 1] << 32);
                mm = (uint64_t)c[(size_t)wi * 4 + 2] | ((uint64_t)c[(size_t)wi * 4 + 3] << 32);
            }
            const int32_t val = valid ? __builtin_popcountll(p) - __builtin_popcountll(mm) : 0;
            const int32_t incl = wave_scan_add(val);
            const int32_t P = base + incl - val;  // value at the first row of word wi
            // f drops by at most 2 per row: a word whose first row is more than 126 above the bound holds no row within it
            const bool cand = valid && (P + hval(i, wi << 6, sh_i) - 126 <= f_max);
            uint64_t mask = __ballot(cand);
            while (mask) {
                const int l = LAST ? 63 - __builtin_clzll(mask) : __builtin_ctzll(mask);
                mask &= ~(1ull << l);
                const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)p, l), phi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(p >> 32), l);
                const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mm, l), mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mm >> 32), l);
                const int32_t Pw = __builtin_amdgcn_readlane(P, l);
                const uint64_t pw = (uint64_t)plo | ((uint64_t)phi << 32), mw = (uint64_t)mlo | ((uint64_t)mhi << 32);
                const uint64_t bits = (1ull << lane) - 1ull;
                const int32_t j = ((wc + l) << 6) + lane;
                const int32_t f = Pw + __builtin_popcountll(pw & bits) - __builtin_popcountll(mw & bits) + hval(i, j, sh_i);
                const uint64_t okm = __ballot(f <= f_max && j >= lo && j <= hi);
                if (okm) {
                    *out = ((wc + l) << 6) + (LAST ? 63 - __builtin_clzll(okm) : __builtin_ctzll(okm));
                    found = true;
                    break;
                }
            }
            if (found && !LAST) return true;
            base += __builtin_amdgcn_readlane(incl, 63);
        }
        return found;
    }
    __device__ __forceinline__ bool scan_first(int32_t k, const BlockRec& r, int32_t i, int32_t f_max, int32_t lo, int32_t hi, int32_t* out) const {
        return scan<false>(k, r, i, f_max, lo, hi, out);
    }
    __device__ __forceinline__ bool scan_last(int32_t k, const BlockRec& r, int32_t i, int32_t f_max, int32_t lo, int32_t hi, int32_t* out) const {
        return scan<true>(k, r, i, f_max, lo, hi, out);
    }
};

// Pairs are claimed by ticket in the order of `order` (heaviest first); a block is four independent wavefronts.
#ifdef PA_UNIT_APA2_SIMPLE  // (the kernel is compiled in a translation unit of its own: csrc/apa2_units.hpp)
__global__ __launch_bounds__(64 * kStripBlockWaves, 4) void apa2_kernel(const PairJob* __restrict__ jobs, const int32_t* __restrict__ order, int npairs,
                                                                    SearchParams sp, uint32_t* ticket, uint32_t* err, uint32_t* dbg, int k1_only,
                                                                    RdvParams rp, unsigned long long* rdv_stats) {
    const int lane = (int)(threadIdx.x & 63);
    __shared__ RdvShared apa2_rdv;
    rdv_init(&apa2_rdv, kStripBlockWaves);
    const RdvLds rdv_lds{(lds_u32)&apa2_rdv, lane};
    const int wave_in_block = (int)rfl((uint32_t)(threadIdx.x >> 6));
    rdv::Counters rdv_total;
    // one 4 KB slice per wavefront for the eq words of K = 4 strips, aligned to its size (the strip ORs offsets into the address)
    __shared__ __attribute__((aligned(4096))) uint32_t apa2_lds_eq[kStripBlockWaves][LdsEq<4>::kWaveBytes / 4];
    typedef __attribute__((address_space(3))) uint32_t* lds_u32p;
    const uint32_t lds_eq_off = (uint32_t)(uintptr_t)(lds_u32p)&apa2_lds_eq[rfl((uint32_t)(threadIdx.x >> 6))][0];
    for (;;) {
        // The ticket, WITHOUT a lane-dependent branch: with `if (lane == 0) t = atomicAdd(..)` here and `if (lane == 0) store` at the
        // end of the body, LLVM threads lanes 1..63 from the end of one iteration straight into the next with t = 0 known, so that
        // readfirstlane runs with lane 0 masked off and those lanes process pair 0 again, for ever (seen in the ISA of the first
        // version).  Every lane adds (lane 0: 1, the others 0; the compiler folds it into one atomic) and lane 0's old value counts.
        uint32_t t = atomicAdd(ticket, lane == 0 ? 1u : 0u);
        t = rfl(t);
        if (t >= (uint32_t)npairs) break;
        if (rp.prio) PA_SETPRIO_BY_RANK(t, npairs);
        const int pair = (int)rfl((uint32_t)order[t]);
        const PairJob job = jobs[pair];
        HeurParams hp;
        hp.kind = sp.heur;
        hp.n = job.n;
        hp.m = job.m;
        hp.sh_h = job.sh_h;
        DevBackend be(job, hp, err, dbg);
        be.k1_only = k1_only != 0;
        be.lds_eq = lds_eq_off;
        be.rdv = rdv_lds;
        be.wave = wave_in_block;
        be.rp = rp;
        be.my_prio = rp.prio ? PA_PRIO_OF_RANK(t, npairs) : 0;
        be.mark(7, (uint32_t)pair + 1u);
        const uint64_t t_pair0 = wall_clock64();
        PairProg<DevBackend> prog(be, hp, sp);
        PairResult res;
        if (job.n > 0 && job.m > 0) {
            prog.run(&res);
        } else {  // an empty sequence: left to the host engine
            res = PairResult{};
            res.status = kErrDegenerate;
        }
        res.pad0 = (uint32_t)(wall_clock64() - t_pair0);  // diagnostics: how long the pair's band search took its wavefront (100 MHz ticks; PA_ALIGN_PROFILE prints the spread)
        if (be.win_fail) res.status = kErrWindow;
        else if (rfl(*(const PA_GLOBAL uint32_t*)err) != PA_ERR_NONE && res.status == kOk) res.status = kErrDevice;
        *job.result = res;  // (every lane stores the same 64 bytes: no lane-dependent branch at the end of the loop body either)
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
#endif  // PA_UNIT_APA2_SIMPLE

}  // namespace apa2
}  // namespace pa
