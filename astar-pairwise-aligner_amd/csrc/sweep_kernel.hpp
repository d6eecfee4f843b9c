// sweep_kernel.hpp -- the gfx950 wave policy of the device-side A*PA2 sweep (sweep_wave.hpp) and its kernels.
//
// One wavefront per workgroup (every strip wants a SIMD of its own: a pass is bound by the dependent-issue latency of one
// wavefront per column, like one long pair in strip_kernel.hpp).  The Myers step is the K = 1 step of strip_kernel.hpp
// (v_bitop3 / v_alignbit / DPP wave_shr:1) plus, in the strip that holds the band's first row, one v_and_or that forces the
// incoming horizontal delta of that row to +1 (blocks.rs:730-734: the top row of a block is H::one()).
//
// Memory scopes (MI355X_MICROARCH.md, hand-off R1/R2): every word another wavefront polls is an 8-byte {tag | value} word
// stored with ONE relaxed agent-scope atomic (sc1, write-through) by lane 0 and read with relaxed agent-scope loads (L1
// bypass); column words are 8-byte agent-scope stores too, their flag (the strip's prefix word) is stored after
// s_waitcnt vmcnt(0).  No fences, no L2 write-backs, placement independent.
#pragma once
#include <hip/hip_runtime.h>

#include "strip_kernel.hpp"
#include "sweep_wave.hpp"

namespace pa {
namespace sweep {

#define PA_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct DeviceWave {
    using vec = uint32_t;
    using mask = bool;

    static __device__ __forceinline__ uint32_t rfl32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
    static __device__ __forceinline__ uint64_t rfl64(uint64_t x) {
        return ((uint64_t)rfl32((uint32_t)(x >> 32)) << 32) | rfl32((uint32_t)x);
    }
    static __device__ __forceinline__ vec splat(uint32_t x) { return x; }
    static __device__ __forceinline__ vec lane_ids() { return threadIdx.x & 63u; }
    static __device__ __forceinline__ vec select(mask m, vec x, vec y) { return m ? x : y; }
    static __device__ __forceinline__ mask eq_u(vec x, uint32_t y) { return x == y; }
    static __device__ __forceinline__ mask ne_u(vec x, uint32_t y) { return x != y; }
    static __device__ __forceinline__ mask le_u(vec x, uint32_t y) { return x <= y; }
    static __device__ __forceinline__ mask ge_i(vec x, int32_t y) { return (int32_t)x >= y; }
    static __device__ __forceinline__ mask lt_i(vec x, int32_t y) { return (int32_t)x < y; }
    static __device__ __forceinline__ mask gt_i(vec x, int32_t y) { return (int32_t)x > y; }
    static __device__ __forceinline__ mask le_i(vec x, int32_t y) { return (int32_t)x <= y; }
    static __device__ __forceinline__ mask and_m(mask x, mask y) { return x && y; }
    static __device__ __forceinline__ vec shr_v(vec x, vec s) { return x >> (s & 31u); }
    static __device__ __forceinline__ vec shl_v(vec x, vec s) { return x << (s & 31u); }
    static __device__ __forceinline__ uint32_t popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
    static __device__ __forceinline__ vec popc_v(vec x) { return (uint32_t)__builtin_popcount(x); }
    static __device__ __forceinline__ uint32_t readlane(vec x, int i) { return (uint32_t)__builtin_amdgcn_readlane((int)x, i); }
    static __device__ __forceinline__ int32_t readlane_i(vec x, int i) { return __builtin_amdgcn_readlane((int)x, i); }
    static __device__ __forceinline__ uint32_t reduce_add(vec x) {
        return (uint32_t)wave_add((int32_t)x);
    }
    static __device__ __forceinline__ vec prefix_excl(vec x) {
        return (uint32_t)wave_scan_add((int32_t)x) - x;
    }

    // ---- memory ----
    // NO divergent branch may appear in these primitives: `if (lane == 0) store` inside the wave program's loops makes LLVM
    // treat the loops as having divergent exits, and every uniform value carried by them (chunk counter, block index, ...)
    // moves to VGPRs with EXEC-mask control flow (measured: 10x per step).  Single-lane stores are inline asm with EXEC = 1,
    // guarded loads clamp their index instead of branching, the rare atomics run on all lanes.
    static __device__ __forceinline__ uint32_t load_u32(const uint32_t* p) {
        return rfl32(__hip_atomic_load((const PA_GLOBAL uint32_t*)p, PA_RLX_AGENT));
    }
    static __device__ __forceinline__ uint64_t load_u64(const uint64_t* p) {
        return rfl64(__hip_atomic_load((const PA_GLOBAL uint64_t*)p, PA_RLX_AGENT));
    }
    // ONE 8-byte write-through (sc1) store by lane 0, as a raw buffer store over an 8-byte buffer: the other lanes' offsets are
    // out of range and the hardware drops them -- no branch, and (unlike inline asm) the compiler still counts the store.
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void store_u64(uint64_t* p, uint64_t v) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, 0, 8, 0x00020000);
        // (readfirstlane: the word is uniform by construction; it also keeps the optimizer from building the pair as a widened
        //  load from the strip state, which would pin that state in scratch memory)
        const u32x2 d = {rfl32((uint32_t)v), rfl32((uint32_t)(v >> 32))};
        const uint32_t off = (threadIdx.x & 63u) == 0 ? 0u : 0x7FFFFFF0u;
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, off, 0, 16);  // aux 16 = sc1
    }
    static __device__ __forceinline__ bool cas_u32(uint32_t* p, uint32_t expect, uint32_t v) {  // rare (end of a pass): all lanes try
        uint32_t e = expect;
        const bool ok = __hip_atomic_compare_exchange_strong(p, &e, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __builtin_amdgcn_ballot_w64(ok) != 0;
    }
    static __device__ __forceinline__ void add_u64(uint64_t* p, uint64_t v) {  // rare (end of a strip): lane 0 adds v, the others 0
        __hip_atomic_fetch_add(p, (threadIdx.x & 63u) == 0 ? v : (uint64_t)0, PA_RLX_AGENT);
    }
    static __device__ __forceinline__ void nap(uint32_t spins) {
        const uint32_t naps = spins < 8u ? 1u : (spins < 64u ? 4u : 16u);
        for (uint32_t k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(8);
    }
    static __device__ __forceinline__ uint64_t clock() { return wall_clock64(); }  // 100 MHz
    static __device__ __forceinline__ void stretch_marker() { asm volatile("; plain stretch"); }
    static __device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    static __device__ __forceinline__ uint32_t ticket(uint32_t* p) {
        return rfl32(atomicAdd(p, (threadIdx.x & 63u) == 0 ? 1u : 0u));  // lane 0's view: the tickets taken before mine
    }
    // lane l (< n) reads the 8-byte word p[l] (the other lanes re-read the last one); the result stays in VGPRs (no
    // broadcast), so the load is in flight until a readlane consumes it
    static __device__ __forceinline__ void load_words(const uint64_t* p, int n, vec& lo, vec& hi) {
        const uint32_t l = threadIdx.x & 63u;
        const uint32_t i = l < (uint32_t)n ? l : (uint32_t)(n - 1);
        const uint64_t w = __hip_atomic_load((const PA_GLOBAL uint64_t*)p + i, PA_RLX_AGENT);
        lo = (uint32_t)w;
        hi = (uint32_t)(w >> 32);
    }
    // the merged records: written by the merge kernel of the previous pass, possibly while this launch is already running
    // (pipelined passes) -- agent-scope loads, never the CU's L1
    static __device__ __forceinline__ void load_i32s(const int32_t* p, int n, vec& v) {
        const uint32_t l = threadIdx.x & 63u;
        v = (uint32_t)__hip_atomic_load((const PA_GLOBAL int32_t*)p + (l < (uint32_t)n ? l : (uint32_t)(n - 1)), PA_RLX_AGENT);
    }
    static __device__ __forceinline__ void load_codes2(const uint32_t* codes, int32_t q, uint32_t& lo, uint32_t& hi) {
        typedef const __attribute__((address_space(4))) uint32_t* ccu32;  // scalar loads (s_load, lgkmcnt)
        const ccu32 cc = (ccu32)codes;
        lo = cc[2 * (int64_t)q];
        hi = cc[2 * (int64_t)q + 1];
    }
    static __device__ __forceinline__ void load_profile(const uint32_t* prof, uint32_t word0, int32_t wtot, vec lane, vec& nb0, vec& nb1) {
        const uint32_t w = word0 + (lane >> 1), half = lane & 1u;
        const bool in = (int32_t)w < wtot;
        const uint32_t wc = in ? w : (uint32_t)(wtot - 1);
        const gcu32 g = (gcu32)prof;
        const uint32_t a = g[(size_t)wc * 4 + half], b = g[(size_t)wc * 4 + 2 + half];
        nb0 = in ? a : 0u;
        nb1 = in ? b : 0u;
    }
    static __device__ __forceinline__ void load_v_words(const uint64_t* col, vec widx, mask inr, vec& plo, vec& phi, vec& mlo, vec& mhi) {
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)widx);  // lane 0's word is always in range
        const int64_t wi = inr ? widx : w0;
        const uint64_t p = __hip_atomic_load((const PA_GLOBAL uint64_t*)col + 2 * wi, PA_RLX_AGENT);
        const uint64_t m = __hip_atomic_load((const PA_GLOBAL uint64_t*)col + 2 * wi + 1, PA_RLX_AGENT);
        plo = inr ? (uint32_t)p : 0u;
        phi = inr ? (uint32_t)(p >> 32) : 0u;
        mlo = inr ? (uint32_t)m : 0u;
        mhi = inr ? (uint32_t)(m >> 32) : 0u;
    }
    // lane l holds half (l & 1) of word word0 + l / 2: the even lane stores the word's p, the odd lane its m (8 bytes each,
    // write-through) = byte 8 * l of the strip's 512-byte column segment; inactive lanes store out of range (dropped)
    static __device__ __forceinline__ void store_v_halves(uint64_t* col, uint32_t word0, vec lane, mask act, vec sp, vec sm) {
        // the other lane of the pair: quad_perm:[1,0,3,2]
        const uint32_t p_other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sp, 0xB1, 0xf, 0xf, true), m_other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sm, 0xB1, 0xf, 0xf, true);
        const bool odd = (lane & 1u) != 0;
        const u32x2 d = {odd ? m_other : sp, odd ? sm : p_other};
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(col + 2 * (int64_t)word0, 0, 512, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, act ? lane * 8u : 0x7FFFFFF0u, 0, 16);
    }

    // One Myers step (strip_kernel.hpp myers_step, K = 1).  FORCE: the lane holding the band's first row replaces the delta
    // coming from the lane above by +1 (andm = 3 keeps the base code, orm = bit 31); all other lanes have andm = ~0, orm = 0.
    template <bool FORCE>
    static __device__ __forceinline__ void myers_k(uint32_t s_x, vec& X, vec& vp, vec& vm, vec nb0, vec nb1, vec& acc, vec andm, vec orm,
                                                   uint32_t k40, uint32_t k80) {
        acc = __builtin_amdgcn_alignbit(acc, X, 30);  // (acc << 2) | (X >> 30)
        uint32_t Xin = dpp_wave_shr1(s_x, X);
        if (FORCE) Xin = (Xin & andm) | orm;
        const uint32_t a0 = (uint32_t)__builtin_amdgcn_sbfe((int)Xin, 0, 1);
        const uint32_t a1 = (uint32_t)__builtin_amdgcn_sbfe((int)Xin, 1, 1);
        const uint32_t hm0 = (Xin >> 30) & 1u;
        uint32_t eq = __builtin_amdgcn_bitop3_b32(a0, nb0, a1 ^ nb1, 0x28);  // (a0 ^ nb0) & (a1 ^ nb1)
        const uint32_t vx = eq | vm;
        eq |= hm0;
        const uint32_t sm = (eq & vp) + vp;
        const uint32_t hx = (sm ^ vp) | eq;
        const uint32_t hp = vm | ~(hx | vp);
        const uint32_t hm = vp & hx;
        const uint32_t xm = __builtin_amdgcn_bitop3_b32(k40, hm >> 1, Xin, 0xCA);  // k40 ? (hm >> 1) : Xin
        const uint32_t Xo = __builtin_amdgcn_bitop3_b32(k80, hp, xm, 0xCA);        // k80 ? hp : xm
        const uint32_t hp2 = __builtin_amdgcn_alignbit(hp, Xin, 31);               // (hp << 1) | carry-in
        const uint32_t hm2 = (hm << 1) | hm0;
        vp = __builtin_amdgcn_bitop3_b32(hm2, vx, hp2, 0xF1);  // hm2 | ~(vx | hp2)
        vm = hp2 & vx;
        X = Xo;
    }
    template <bool FORCE>
    static __device__ __forceinline__ void myers(uint32_t s_x, vec& X, vec& vp, vec& vm, vec nb0, vec nb1, vec& acc, vec andm, vec orm) {
        uint32_t k40 = 0x40000000u, k80 = 0x80000000u;
        asm volatile("" : "+v"(k40), "+v"(k80));
        myers_k<FORCE>(s_x, X, vp, vm, nb0, nb1, acc, andm, orm, k40, k80);
    }
    // The 32 steps of a CROSSING chunk are rolled loops (PA_SWEEP_UNROLL steps per trip) -- round 6: fully unrolled, the kernel was
    // 16 000 instructions (110 KB), the crossing chunks alone 8 KB each, executed twice per block between 25 us of other code: a lone
    // wavefront waits for every cold line of it (measured: 3.9 us per crossing chunk for 1 100 instructions, twice their issue time;
    // 3.1 us rolled: profiles/r06_runs/sweep_timers.log).
#ifndef PA_SWEEP_UNROLL
#define PA_SWEEP_UNROLL 4
#endif
#define PA_PRAGMA_(x) _Pragma(#x)
#define PA_UNROLL_N(n) PA_PRAGMA_(unroll n)
    template <bool FORCE>
    static __device__ __forceinline__ void chunk(vec XS, vec& X, vec& vp, vec& vm, vec nb0, vec nb1, vec& acc_lo, vec& acc_hi, vec andm, vec orm) {
        uint32_t k40 = 0x40000000u, k80 = 0x80000000u;  // kept in VGPRs and opaque (see strip_kernel.hpp)
        asm volatile("" : "+v"(k40), "+v"(k80));
        // (the plain chunk stays unrolled: six of them run back to back per block, the code is hot, and a loop costs it 20 %)
#pragma unroll
        for (int j = 0; j < 16; ++j) myers_k<FORCE>((uint32_t)__builtin_amdgcn_readlane((int)XS, j), X, vp, vm, nb0, nb1, acc_lo, andm, orm, k40, k80);
#pragma unroll
        for (int j = 16; j < 32; ++j) myers_k<FORCE>((uint32_t)__builtin_amdgcn_readlane((int)XS, j), X, vp, vm, nb0, nb1, acc_hi, andm, orm, k40, k80);
    }
    static __device__ __forceinline__ bool any(vec x) { return __builtin_amdgcn_ballot_w64(x != 0) != 0; }

    // A chunk in which lane cl0 + j leaves its block at step j: snapshot of its V (the block's column); EXTRA: a lane that may
    // hold the next block's first row starts forcing +1 (fpend), a lane that was below the band restarts from V::one()
    // (resetm; blocks.rs:753-767).  3 resp. 9 more VALU per step than `chunk`.
    template <bool FORCE, bool EXTRA>
    static __device__ __forceinline__ void cross_step(int j, vec XS, vec& X, vec& vp, vec& vm, vec nb0, vec nb1, vec& acc, vec& andm, vec& orm, uint32_t rel,
                                                       uint32_t relr, uint32_t relf, vec& snap_p, vec& snap_m, uint32_t k40, uint32_t k80) {
        // rel / relr / relf are ONE-HOT step masks (bit j set: the lane's event is due in step j): a predicate is one v_bfe_i32 (0 / ~0)
        // and the selects are v_bitop3 -- VALU to VALU, no scalar mask in between (same time as v_cmp + v_cndmask; fewer scalar registers).
        const uint32_t me = (uint32_t)__builtin_amdgcn_sbfe((int)rel, j, 1);
        snap_p = __builtin_amdgcn_bitop3_b32(me, vp, snap_p, 0xCA);  // me ? vp : snap_p
        snap_m = __builtin_amdgcn_bitop3_b32(me, vm, snap_m, 0xCA);
        if (EXTRA) {
            if (FORCE) {
                const uint32_t mf = (uint32_t)__builtin_amdgcn_sbfe((int)relf, j, 1);
                andm = __builtin_amdgcn_bitop3_b32(mf, 3u, andm, 0xCA);
                orm = __builtin_amdgcn_bitop3_b32(mf, k80, orm, 0xCA);
            }
            const uint32_t mr = (uint32_t)__builtin_amdgcn_sbfe((int)relr, j, 1);
            vp |= mr;
            vm &= ~mr;
        }
        myers_k<FORCE>((uint32_t)__builtin_amdgcn_readlane((int)XS, j), X, vp, vm, nb0, nb1, acc, andm, orm, k40, k80);
    }
    template <bool FORCE, bool EXTRA>
    static __device__ __forceinline__ void chunk_cross(vec XS, vec& X, vec& vp, vec& vm, vec nb0, vec nb1, vec& acc_lo, vec& acc_hi, vec& andm, vec& orm,
                                                        vec lane, int32_t cl0, vec& snap_p, vec& snap_m, vec resetm, vec fpend) {
        uint32_t k40 = 0x40000000u, k80 = 0x80000000u;
        asm volatile("" : "+v"(k40), "+v"(k80));
        const uint32_t d = lane - (uint32_t)cl0;                 // == j at the lane's crossing step
        const uint32_t rel = d < 32u ? 1u << d : 0u;             // one-hot over the chunk's steps
        const uint32_t relr = resetm != 0 ? rel : 0u;            // never set for lanes that do not reset
        const uint32_t relf = fpend != 0 ? rel : 0u;
        PA_UNROLL_N(PA_SWEEP_UNROLL)
        for (int j = 0; j < 16; ++j) cross_step<FORCE, EXTRA>(j, XS, X, vp, vm, nb0, nb1, acc_lo, andm, orm, rel, relr, relf, snap_p, snap_m, k40, k80);
        PA_UNROLL_N(PA_SWEEP_UNROLL)
        for (int j = 16; j < 32; ++j) cross_step<FORCE, EXTRA>(j, XS, X, vp, vm, nb0, nb1, acc_hi, andm, orm, rel, relr, relf, snap_p, snap_m, k40, k80);
    }
    // Steps [j0, j1) of such a chunk in the strip that runs the top-down scan (the scan interrupts the chunk where the scanned
    // row's lane crosses): the same step, one at a time.
    static __device__ __forceinline__ void chunk_cross_range(vec XS, vec& X, vec& vp, vec& vm, vec nb0, vec nb1, vec& acc_lo, vec& acc_hi, vec& andm,
                                                              vec& orm, vec lane, int32_t cl0, vec& snap_p, vec& snap_m, vec resetm, vec fpend, int32_t j0,
                                                              int32_t j1) {
        uint32_t k40 = 0x40000000u, k80 = 0x80000000u;
        asm volatile("" : "+v"(k40), "+v"(k80));
        const uint32_t d = lane - (uint32_t)cl0;
        const uint32_t rel = d < 32u ? 1u << d : 0u;
        const uint32_t relr = resetm != 0 ? rel : 0u;
        const uint32_t relf = fpend != 0 ? rel : 0u;
        // (four steps per trip instead of one changed nothing: profiles/r06_runs/sweep_timers.log)
        const int32_t mid = j1 < 16 ? j1 : 16;
#pragma unroll 1
        for (int j = j0; j < mid; ++j) cross_step<true, true>(j, XS, X, vp, vm, nb0, nb1, acc_lo, andm, orm, rel, relr, relf, snap_p, snap_m, k40, k80);
#pragma unroll 1
        for (int j = j0 > 16 ? j0 : 16; j < j1; ++j) cross_step<true, true>(j, XS, X, vp, vm, nb0, nb1, acc_hi, andm, orm, rel, relr, relf, snap_p, snap_m, k40, k80);
    }
};

// What the host decides before a pass (sweep_host.hpp PassInit) + where it goes.
struct InitArgs {
    BRec* brec;
    TRec* trec;
    uint64_t* bprog;
    uint64_t* strip_start;
    Status* status;
    uint32_t* ticket;
    uint32_t pass;
    int32_t js1, je1, ojs1, oje1, flags1, top1, fs0, last_strip, nstrips;
    uint64_t* gran;       // the pass's hand-off granules: cleared here (one launch less than a memset in front of this kernel)
    uint64_t gran_words;  // 8-byte words
};

constexpr int kInitThreads = 256;
__global__ __launch_bounds__(kInitThreads) void sweep_init_kernel(InitArgs a) {
    for (uint64_t i = (uint64_t)blockIdx.x * kInitThreads + threadIdx.x; i < a.gran_words; i += (uint64_t)gridDim.x * kInitThreads) a.gran[i] = 0;
    if (blockIdx.x != 0) return;
    const uint32_t l = threadIdx.x;
    const uint32_t t1 = blk_tag(a.pass, 1);
    if (l == 0) {
        BRec* b = a.brec + 1;
        b->js = tw_make(t1, a.js1);
        b->je = tw_make(t1, a.je1);
        b->ojs = tw_make(t1, a.ojs1);
        b->oje = tw_make(t1, a.oje1);
        b->flags = tw_make(t1, a.flags1);
        b->smax = tw_make(t1, a.last_strip);
        b->specmax = tw_make(t1, 0);
        TRec* t = a.trec + 1;
        t->js = tw_make(t1, a.js1);
        t->top_val = tw_make(t1, a.top1);
        t->fs_prev = tw_make(t1, a.fs0);
        t->lim = tw_make(t1, 0);
        t->found = tw_make(t1, 0);
        t->state = tw_make(t1, kTDesc);
        *a.bprog = tw_make(t1, a.oje1);
        *a.ticket = 0;
    }
    uint32_t* st = reinterpret_cast<uint32_t*>(a.status);
    for (uint32_t i = l; i < sizeof(Status) / 4; i += kInitThreads) st[i] = 0;
    for (int32_t r = (int32_t)l; r <= a.last_strip && r < a.nstrips; r += kInitThreads) a.strip_start[r] = tw_make(a.pass, 1);
}

__global__ __launch_bounds__(64) void sweep_kernel(Ctx c) { wave_main<DeviceWave>(c); }

// After a pass: merged_new = the older passes' records with this pass's on top (Blocks::blocks persists across
// align_for_bounded_dist calls, lib.rs:140-158; a block the pass did not reach, or did not fix, keeps its older record).  A pass
// that did not end by itself (aborted, cancelled) merges nothing.  The two merged arrays alternate from pass to pass.
// The workgroup that finishes last (a counter, left at zero) then publishes the word the next pass polls, after the merged
// records, and copies the pass's status into the host's pinned copy -- no separate launch, no copy command behind the pass.
__global__ void sweep_merge_kernel(const BRec* brec, const BlockRec* merged_old, BlockRec* merged_new, const Status* status, int32_t nblk,
                                   uint32_t* counter, uint64_t* done, uint32_t pass, Status* host_status) {
    const int32_t k = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k <= nblk + 1) {
        BlockRec d = merged_old[k];
        const bool ended = status->state == kStDone || status->state == kStNoPath;
        if (ended && k >= 1 && k <= nblk && k <= status->k_end) {
            const BRec& s = brec[k];
            d.js = tw_val(s.js);
            d.je = tw_val(s.je);
            d.ojs = tw_val(s.ojs);
            d.oje = tw_val(s.oje);
            if (k <= status->k_fixed) {
                d.fs = tw_val(s.fs);
                d.fe = tw_val(s.fe);
                d.top_val = tw_val(s.top_val);
                d.bot_val = tw_val(s.bot_val);
            }
        }
        merged_new[k] = d;
    }
    __threadfence();  // this thread's records are visible device-wide ...
    __syncthreads();  // ... for every thread of the workgroup, before it counts itself
    if (threadIdx.x != 0) return;
    const uint32_t c = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (c + 1 != gridDim.x) return;
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(status);
    uint32_t* dst = reinterpret_cast<uint32_t*>(host_status);
    for (uint32_t i = 0; i < sizeof(Status) / 4; ++i) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(done, (uint64_t)pass, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// Traceback: gather the blocks' columns (window-addressed, see col_base_word) into one packed buffer, and the fields of
// the block records the host needs.
struct BlockOut {
    int32_t js, je, ojs, oje, fs, fe, top_val, bot_val;
};
__global__ void sweep_records_kernel(const BRec* brec, BlockOut* out, int32_t nblk) {
    const int32_t k = 1 + (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k > nblk) return;
    const BRec& s = brec[k];
    BlockOut o;
    o.js = tw_val(s.js);
    o.je = tw_val(s.je);
    o.ojs = tw_val(s.ojs);
    o.oje = tw_val(s.oje);
    o.fs = tw_val(s.fs);
    o.fe = tw_val(s.fe);
    o.top_val = tw_val(s.top_val);
    o.bot_val = tw_val(s.bot_val);
    out[k] = o;
}
// The same records AND where every block's column goes in the packed buffer (an exclusive prefix over the blocks' heights), in one
// launch of one workgroup: the host does not have to see the records before the columns can be gathered (round 6: one stream
// synchronisation per traceback instead of two).  `out_host`: the host's pinned copy of the records.
__global__ __launch_bounds__(1024) void sweep_records_offsets_kernel(const BRec* brec, BlockOut* out, BlockOut* out_host, int64_t* offs, int32_t nblk) {
    __shared__ int64_t sh[1024];
    __shared__ int64_t carry_s;
    const int t = (int)threadIdx.x;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int32_t base = 1; base <= nblk; base += 1024) {
        const int32_t k = base + t;
        int64_t len = 0;
        if (k <= nblk) {
            const BRec& s = brec[k];
            BlockOut o;
            o.js = tw_val(s.js);
            o.je = tw_val(s.je);
            o.ojs = tw_val(s.ojs);
            o.oje = tw_val(s.oje);
            o.fs = tw_val(s.fs);
            o.fe = tw_val(s.fe);
            o.top_val = tw_val(s.top_val);
            o.bot_val = tw_val(s.bot_val);
            out[k] = o;
            out_host[k] = o;
            len = (o.je - o.js) / 64;
        }
        sh[t] = len;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan over the 1024 blocks of this round
            const int64_t v = t >= d ? sh[t - d] : 0;
            __syncthreads();
            sh[t] += v;
            __syncthreads();
        }
        const int64_t carry = carry_s;
        if (k <= nblk) offs[k] = carry + sh[t] - len;
        __syncthreads();
        if (t == 1023) carry_s = carry + sh[1023];
        __syncthreads();
    }
}
__global__ void sweep_gather_kernel(const uint64_t* col, int64_t col_stride, int32_t win, const BlockOut* recs, const int64_t* offs,
                                    uint64_t* out, int32_t nblk) {
    const int32_t k = 1 + (int32_t)blockIdx.x;
    if (k > nblk) return;
    const int64_t lo = (int64_t)(k - 1) * kBlockW - win;
    const int64_t base = lo <= 0 ? 0 : (lo >> 6);
    const uint64_t* src = col + ((int64_t)k * col_stride - base) * 2;
    const int64_t w0 = recs[k].js >> 6, w1 = recs[k].je >> 6;
    uint64_t* dst = out + offs[k] * 2;
    for (int64_t i = threadIdx.x; i < (w1 - w0) * 2; i += blockDim.x) dst[i] = src[w0 * 2 + i];
}

}  // namespace sweep
}  // namespace pa
