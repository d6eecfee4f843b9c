// slice_kernel.hpp -- the BIT-SLICED full-DP kernel for big cost-only batches (round 6).
//
// What it computes: the same thing as pair_kernel / strip_kernel for a cost-only rectangle that starts from fresh borders -- the unit-cost
// edit distance of every pair, by the column step of pa-bitpacking/src/myers.rs:27-55 -- in a TRANSPOSED layout: bit p of every 32-bit
// register belongs to pair p of a GROUP of 32 pairs, and one register holds ONE DP row (of 32 pairs) instead of 32 rows (of one pair).
// DP values do not depend on the schedule, so the distances are those of the reference bit for bit.
//
// Why: in this layout the step needs neither the add nor the shifts.  For one row i, with c_0 = hm_in and c_(i+1) = hm_i, the carry
// into bit i of ((eq & vp) + vp) IS the horizontal minus-delta of the row above, hx_i = eq_i | hm_(i-1), and `<< 1` is "take the row
// above's register":
//     eq_i  = (a0 ^ nb0_i) & (a1 ^ nb1_i)        profile.rs:141-144 on bit planes: a0 / a1 = the column's code bits over the 32 pairs,
//                                                nb0_i / nb1_i = the NEGATED code bits of row i (the reference's negated planes)
//     hm_i  = vp_i & (eq_i | hm_(i-1))           myers.rs:36-39  (hx, hm)
//     hp_i  = vm_i | ~(eq_i | hm_(i-1) | vp_i)   myers.rs:38
//     vp'_i = hm_(i-1) | ~(eq_i | vm_i | hp_(i-1))   myers.rs:33,50 with the shifted hp / hm of :44-47
//     vm'_i = hp_(i-1) & (eq_i | vm_i)           myers.rs:51
// 8 two- or three-input logic instructions per (row x 32 pairs x 64 lanes) = 2048 cells, all of the fast VALU class (4 v_bitop3 with three
// VGPR sources, 4 VOP2) -- no v_add_co / v_addc, no v_alignbit, no per-row DPP -- against 11.3 mixed instructions per 2048 cells in
// pair_kernel<8>.  tools/slice_probe.hip, profiles/r06_runs/slice_probe*.log: 196 TCUPS against 130.
//
// Shape: a lane owns R consecutive rows in registers (vp, vm, nb0, nb1: 4 R VGPRs; R = 56 at two wavefronts per SIMD).  The 64 lanes of a
// wavefront are skewed one column per lane (lane l works on column t - l at step t), so a STRIP is 64 R rows; the bottom row's (hp, hm)
// goes to the next lane through one DPP wave_shr:1 each per step.  Strips of a group hand their bottom row down through HBM, 8 bytes per
// column -- (hp, hm) of the 32 pairs -- and run CONCURRENTLY one behind the other: a consumer polls the value itself ("data is the flag",
// MI355X_MICROARCH.md R2: a boundary row is preset to hp = hm = ~0, which no real delta pair can be).  Jobs (group, strip) are claimed by an
// atomic ticket in producer-before-consumer order, so a consumer's producer has always started; every poll is bounded.
//
// Ragged pairs: rows beyond |b| of a pair are padding that nothing reads (rows only depend on rows above them).  Pairs of a group whose
// |a| differ are CAPTURED: when a lane has finished column |a_p| of pair p it stores bit p of its rows' (vp, vm) -- events sorted by
// column, one per distinct |a| of the group; the last event is the group's last column.  slice_score_kernel then sums, per pair, the
// vertical deltas of its rows of the captured column: cost = |a| + sum over rows < |b| of (vp - vm)   (Block::index, block.rs:100-121).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pa {
namespace slice {

#ifndef PA_SLICE_CLOCK_TURNS
#define PA_SLICE_CLOCK_TURNS 15
#endif
#ifndef PA_SLICE_YIELD
#define PA_SLICE_YIELD 2
#endif
constexpr int kPad = 64;          // entries in front of column 0 and behind the last column of the per-column arrays: no clamping in the loop
constexpr int kErrSpin = 7;       // a boundary value did not arrive in time (PA_ERR_SPIN_TIMEOUT of the strip kernels)
constexpr uint32_t kSpinLimit = 1u << 22;  // reloads of one boundary value (a reload is a round trip to the L2 or further: seconds in total)

struct SliceGroup {       // one group of up to 32 pairs (positions first_pos .. first_pos + npairs of the sorted order)
    uint64_t a_off;       // the group's column planes: A[a_off + kPad + c] = (a0, a1) of column c
    uint64_t b_off;       // the group's row planes: B[b_off + r] = (nb0, nb1) of row r, V[b_off + r] = captured (vp, vm); nstrips * 64 * R rows
    uint64_t h_off;       // the group's boundary rows: H[h_off + s * h_stride + c] = (hp, hm) below strip s at column c, s < nstrips - 1
    uint32_t h_stride;    // n + 2 * kPad
    int32_t n;            // columns: the longest a of the group
    int32_t nstrips;
    int32_t npairs;
    uint32_t first_pos;
    uint32_t ev_first, ev_count;  // capture events: events[ev_first .. ev_first + ev_count), increasing columns, the last one at n
    uint32_t pad_;
};
struct SliceEvent {
    int32_t col;    // capture AFTER this many columns (= |a| of the pairs in mask)
    uint32_t mask;  // the pairs (bits) whose a ends there
};
struct SliceJob {
    uint32_t group, strip;
};
struct SlicePair {  // per position of the sorted order
    uint64_t code_off, prof_off;  // into the batch's packed codes (u32, 16 columns each) / profile (two u64 per 64 rows)
    int32_t n, m;
    uint32_t pair;  // index of the pair in the batch (where its cost goes)
    uint32_t pad_;
};

typedef uint32_t pa_slice_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t old_, uint32_t src) {
    // v_mov_b32_dpp wave_shr:1 ; lane 0 has no source lane and keeps `old_`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old_, (int)src, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_wave_rol1(uint32_t x) {
    // v_mov_b32_dpp wave_rol:1 ; lane i takes lane i + 1's value, lane 63 lane 0's
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x134, 0xf, 0xf, false);
}
__device__ __forceinline__ uint2 ld_boundary(const uint2* p) {  // L1-bypassing 8-byte load, one access (agent scope: the producer may sit on another XCD)
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ void st_boundary(uint2* p, uint32_t hp, uint32_t hm) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)hp | ((unsigned long long)hm << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Two rows (A = i, B = i + 1) of one column step, in an order in which no instruction reads the result of the one before it and the chain
// value hm_B is ready eight instructions before the block ends.  Inline asm because the compiler's own order hoists the whole hm chain of
// the lane in front of everything else (2 R live temporaries: spills at R >= 40) -- tools/slice_probe.hip, SLICE_ASM=0.  Every asm statement
// costs one s_nop (the hazard recognizer assumes the worst of a register that one asm statement writes and the next reads); blocks of FOUR
// rows halve those and run 16-20 % SLOWER all the same, whichever order the 32 instructions have (profiles/r06_runs/slice_variants.log).
#define PA_SLICE_ROW_PAIR(vpA, vmA, vpB, vmB, nb0A, nb1A, nb0B, nb1B, a0, a1, hpp, hmp, hpo, hmo)                                      \
    do {                                                                                                                                \
        uint32_t eA_, eB_, x_, vx_, hmA_, hpA_;                                                                                          \
        asm volatile(                                                                                                                   \
            "v_xor_b32 %[eA], %[a1_], %[nb1A_]\n\t"                                                                                     \
            "v_xor_b32 %[eB], %[a1_], %[nb1B_]\n\t"                                                                                     \
            "v_bitop3_b32 %[eA], %[a0_], %[nb0A_], %[eA] bitop3:0x28\n\t"                                                               \
            "v_bitop3_b32 %[eB], %[a0_], %[nb0B_], %[eB] bitop3:0x28\n\t"                                                               \
            "v_bitop3_b32 %[hmA], %[vpA_], %[eA], %[hmp_] bitop3:0xe0\n\t"                                                              \
            "v_or_b32 %[x], %[eA], %[hmp_]\n\t"                                                                                         \
            "v_or_b32 %[vx], %[eA], %[vmA_]\n\t"                                                                                        \
            "v_bitop3_b32 %[hmB], %[vpB_], %[eB], %[hmA] bitop3:0xe0\n\t"                                                               \
            "v_bitop3_b32 %[hpA], %[vmA_], %[x], %[vpA_] bitop3:0xf1\n\t"                                                               \
            "v_bitop3_b32 %[vpA_], %[hmp_], %[vx], %[hpp_] bitop3:0xf1\n\t"                                                             \
            "v_and_b32 %[vmA_], %[hpp_], %[vx]\n\t"                                                                                     \
            "v_or_b32 %[x], %[eB], %[hmA]\n\t"                                                                                          \
            "v_or_b32 %[vx], %[eB], %[vmB_]\n\t"                                                                                        \
            "v_bitop3_b32 %[hpB], %[vmB_], %[x], %[vpB_] bitop3:0xf1\n\t"                                                               \
            "v_bitop3_b32 %[vpB_], %[hmA], %[vx], %[hpA] bitop3:0xf1\n\t"                                                               \
            "v_and_b32 %[vmB_], %[hpA], %[vx]"                                                                                          \
            : [eA] "=&v"(eA_), [eB] "=&v"(eB_), [x] "=&v"(x_), [vx] "=&v"(vx_), [hmA] "=&v"(hmA_), [hpA] "=&v"(hpA_), [hmB] "=&v"(hmo),  \
              [hpB] "=&v"(hpo), [vpA_] "+v"(vpA), [vmA_] "+v"(vmA), [vpB_] "+v"(vpB), [vmB_] "+v"(vmB)                                  \
            : [a0_] "v"(a0), [a1_] "v"(a1), [nb0A_] "v"(nb0A), [nb1A_] "v"(nb1A), [nb0B_] "v"(nb0B), [nb1B_] "v"(nb1B), [hpp_] "v"(hpp), \
              [hmp_] "v"(hmp));                                                                                                         \
    } while (0)

// ticket_err[0] = ticket, [1] = error code (first one wins)
template <int R>
__global__ __launch_bounds__(64, 2) void slice_kernel(const SliceJob* __restrict__ jobs, int njobs, const SliceGroup* __restrict__ groups,
                                                      const SliceEvent* __restrict__ events, const uint2* __restrict__ A, const uint2* __restrict__ B,
                                                      uint2* H, uint2* V, uint32_t* ticket_err, unsigned long long* dbg) {
    static_assert(R % 2 == 0, "rows are stepped in pairs");
    const int lane = (int)threadIdx.x;
    const uint32_t wave_slot = (uint32_t)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 4) & 15u;  // HW_REG_HW_ID[3:0]: this wavefront's slot on its SIMD
    for (;;) {
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(ticket_err, 1u);
        tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
        if (tk >= (uint32_t)njobs) break;
        const SliceJob job = jobs[tk];
        const SliceGroup grp = groups[job.group];
        // diagnostics (PA_SLICE_JOBTIMES; dbg is null otherwise): ticks of 10 ns of this job, and of them asleep waiting for the strip above
        const unsigned long long t_job0 = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
        unsigned long long t_parked = 0, n_parks = 0;
        [[maybe_unused]] bool waited_before = false;
        [[maybe_unused]] int slack = 0;
        const int s = (int)job.strip, n = grp.n;
        const bool has_in = s > 0, has_out = s + 1 < grp.nstrips;
        const uint2* Ag = A + grp.a_off + kPad;
        const size_t row0 = grp.b_off + ((size_t)s * 64 + (size_t)lane) * R;
        const uint2* Hin = H + grp.h_off + (size_t)(has_in ? s - 1 : 0) * grp.h_stride + kPad;
        uint2* Hout = H + grp.h_off + (size_t)(has_out ? s : 0) * grp.h_stride + kPad;
        const SliceEvent* ev = events + grp.ev_first;
        uint32_t nb0[R], nb1[R], vp[R], vm[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const uint2 b = B[row0 + i];
            nb0[i] = b.x;
            nb1[i] = b.y;
            vp[i] = ~0u;  // the left column of the matrix: +1 everywhere (V::one, blocks.rs:163)
            vm[i] = 0u;
        }
        // Columns come in CHUNKS of 64: lane j loads column 64 q + j of the group's planes and of the boundary row above (one coalesced load
        // each per chunk, issued a whole chunk ahead); the chunk registers ROTATE one lane per step (DPP wave_rol:1), so that at step j lane 0 finds column 64 q + j in its own lane,
        // and everything -- the column's two code planes AND the row's (hp, hm) -- then moves down the lanes one lane per step through DPP
        // wave_shr:1.  No vector memory load sits in the step loop, so no wait does either.
        //
        // The wait for the prefetched chunk (round 6).  Loads and stores of a wavefront retire in order through ONE counter (vmcnt), and the
        // compiler's wait for the next chunk's registers -- placed at their first use, the top of the next chunk -- was vmcnt(0): behind the
        // write-through boundary store of the step just before, i.e. one round trip to memory per chunk with the wavefront parked
        // (tools/slice_wait_probe.py).  So the prefetch loads are issued from inline asm (the compiler keeps no score for them) and waited
        // for by hand kWaitStep steps later with vmcnt(kWaitStep): by then exactly kWaitStep younger stores have been issued -- ONE PER STEP,
        // by EVERY strip, by ALL lanes: a raw buffer store whose offset is out of range except in lane 63 of a strip that has a strip below
        // (an out-of-range store is dropped by the memory pipeline but issued and counted like any other), so the instruction is issued
        // whenever any lane is at a column, which is every step of the loop -- and "at most kWaitStep operations outstanding" then means
        // "everything older than those stores has retired": the loads, and the stores of the chunk before, issued microseconds ago.
        // Rules that keep this sound (tests/test_slice_isa.py checks the first two in the compiled code of every instantiation):
        //  * between the load and the wait nothing may READ the loads' destination registers -- the compiler believes they are valid from
        //    the asm statement on; one place issues them and one place consumes them, on every path (no prefetch in a strip's last chunk:
        //    every other chunk has all 64 steps), so there is no join that would need a copy;
        //  * no compiler-visible wait may sit in the step loop (the row loads of the job are waited for before the loops);
        //  * more operations in between (the capture's atomics) only make the wait stronger.
        constexpr int kWaitStep = 8;
        uint32_t o_hp = 0, o_hm = 0, o_a0 = 0, o_a1 = 0;
        uint32_t ev_i = 0;
        int ev_col = ev[0].col;
        const int nchunks = (n + 63 + 63) / 64;  // steps 0 .. n + 62
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)Hout, 0, (int)((uint32_t)n * 8u), 0x00020000);
        unsigned long long nA = 0, nH = ~0ull;
        uint2 kA = Ag[lane], kH = make_uint2(~0u, 0u);  // chunk 0: ordinary loads
        if (has_in) kH = ld_boundary(Hin + lane);
        __builtin_amdgcn_s_waitcnt(0);  // nothing in flight when the loops start (a wait the compiler derives from these loads would sit inside them)
        for (int q = 0; q < nchunks; ++q) {
            // Fair shares of the SIMD.  With equal priority the OLDER of a SIMD's two wavefronts wins every arbitration (MI355X_MICROARCH.md:
            // priority, then age): equal jobs took 83 ms on one and 96 ms on the other wavefront of a SIMD (PA_SLICE_JOBTIMES), a chain of
            // strips runs at the pace of its slowest member and the fast members sleep at the chunk tops while their SIMD runs half empty.
            // So the two take TURNS at priority 1, the wavefront in the odd slot of the SIMD (HW_REG_HW_ID) the other way round -- by a bit of the
            // shared 100 MHz clock (2^15 ticks = 0.33 ms, five chunks), looked at once per chunk, so that the two are complementary whatever
            // their phases (by the wavefront's own chunk count: equal jobs 87 .. 92 ms, the bench batch 408 -> 392 ms; by the clock 1.5 % more.
            // Turns of 8 / 16 / 32 steps, other clock bits, a third wavefront per SIMD with three-way turns: profiles/r06_runs/slice_variants.log).
#if PA_SLICE_CLOCK_TURNS
            // (whose turn it is by the shared 100 MHz clock instead of the wavefront's own chunk count: complementary whatever the phases)
            if ((((uint32_t)__builtin_amdgcn_s_memrealtime() >> PA_SLICE_CLOCK_TURNS) ^ wave_slot) & 1u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#else
            if (((uint32_t)q ^ wave_slot) & 1u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#endif
            uint2 cA = kA, cH = kH;
            const int col = q * 64 + lane;  // the column this lane holds for lane 0
            if (has_in) {
                // The strip above has to be past this chunk.  It normally is (it started first and runs at the same pace); when this strip has
                // caught up, waiting here until all 64 columns are there lets it run them at full speed.  (Letting the strip above get three
                // chunks ahead once a chunk was found missing changed nothing: profiles/r06_runs/slice_variants.log.)
                uint32_t spins = 0;
                const unsigned long long t_p0 = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
                while (col < n && (cH.x & cH.y) != 0u) {
                    __builtin_amdgcn_s_sleep(32);
                    cH = ld_boundary(Hin + col);
                    if (++spins > kSpinLimit) {
                        atomicCAS(ticket_err + 1, 0u, (uint32_t)kErrSpin);
                        cH = make_uint2(~0u, 0u);
                    }
                }
                const bool waited = __builtin_amdgcn_readfirstlane((int)(__ballot(spins != 0u) != 0ull)) != 0;
#if PA_SLICE_YIELD == 1
                if (waited) __builtin_amdgcn_s_setprio(0);  // slack: whatever the turn, the neighbour goes first in this chunk
#elif PA_SLICE_YIELD == 2
                if (waited || waited_before) __builtin_amdgcn_s_setprio(0);
                waited_before = waited;
#elif PA_SLICE_YIELD >= 3
                // yield in the chunk of a wait and the PA_SLICE_YIELD - 1 chunks after it
                slack = waited ? PA_SLICE_YIELD : (slack > 0 ? slack - 1 : 0);
                if (slack > 0) __builtin_amdgcn_s_setprio(0);
#endif
                if (dbg && waited) {
                    t_parked += __builtin_amdgcn_s_memrealtime() - t_p0;
                    n_parks += 1;
                }
            }
            // the next chunk's values, a whole chunk ahead of their use (issued after the test above: a wait for THIS chunk's values must not
            // cover loads that have only just been issued)
            const int pcol = min(col + 64, n + kPad - 1);  // (behind the last column: the pad, never used)
            const bool more = q + 1 < nchunks;              // (uniform) a chunk follows: this one has all 64 steps
            if (more) {
                asm volatile("global_load_dwordx2 %0, %1, off ; pa_prefetch_load" : "=v"(nA) : "v"(Ag + pcol) : "memory");
                if (has_in) asm volatile("global_load_dwordx2 %0, %1, off sc1 ; pa_prefetch_load" : "=v"(nH) : "v"(Hin + pcol) : "memory");
            }
            const int jend = min(64, n + 63 - q * 64);
            const int wait_at = more ? kWaitStep : -1;  // (one scalar compare per step)
            for (int j = 0; j < jend; ++j) {
                if (j == wait_at) {  // (uniform)
                    asm volatile("s_waitcnt vmcnt(8) ; pa_prefetch_wait" : "+v"(nA), "+v"(nH)::"memory");
                    kA = make_uint2((uint32_t)nA, (uint32_t)(nA >> 32));
                    if (has_in) kH = make_uint2((uint32_t)nH, (uint32_t)(nH >> 32));
                }
                const int c = q * 64 + j - lane;
                // lane 0 takes the chunk registers' value of ITS lane -- they rotate one lane per step, so that is column 64 q + j --, every other
                // lane the value the lane above it had a step ago
                const uint32_t a0 = dpp_wave_shr1(cA.x, o_a0), a1 = dpp_wave_shr1(cA.y, o_a1);
                uint32_t hpp = dpp_wave_shr1(cH.x, o_hp), hmp = dpp_wave_shr1(cH.y, o_hm);
                cA.x = dpp_wave_rol1(cA.x);
                cA.y = dpp_wave_rol1(cA.y);
                cH.x = dpp_wave_rol1(cH.x);
                cH.y = dpp_wave_rol1(cH.y);
                o_a0 = a0;
                o_a1 = a1;
                if ((unsigned)c < (unsigned)n) {
#pragma unroll
                    for (int i = 0; i < R; i += 2) {
                        uint32_t hpo, hmo;
                        PA_SLICE_ROW_PAIR(vp[i], vm[i], vp[i + 1], vm[i + 1], nb0[i], nb1[i], nb0[i + 1], nb1[i + 1], a0, a1, hpp, hmp, hpo, hmo);
                        hpp = hpo;
                        hmp = hmo;
                    }
                    o_hp = hpp;
                    o_hm = hmp;
                    {  // lane 63's (hp, hm) of column c: 8 bytes, write-through; every other offset is out of range (see the rules above)
                        const pa_slice_u32x2 d = {hpp, hmp};
                        __builtin_amdgcn_raw_buffer_store_b64(d, hrs, (lane == 63 && has_out) ? (uint32_t)c * 8u : 0x7FFFFFF0u, 0, 16);  // aux 16 = sc1
                    }
                    if (c + 1 == ev_col) {  // some pairs' a ends here: keep their bits of this lane's rows (the last event is the last column)
                        const uint32_t mask = ev[ev_i].mask;
                        // V is zeroed before every pass; a pair is captured once, so OR-ing its bit in is exact.  One row at a time (the
                        // compiler would otherwise hold 2 R masked values at once: this path is rare, registers are not)
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            atomicOr(&V[row0 + i].x, vp[i] & mask);
                            atomicOr(&V[row0 + i].y, vm[i] & mask);
                            asm volatile("" ::: "memory");
                        }
                        ++ev_i;
                        ev_col = ev_i < grp.ev_count ? ev[ev_i].col : -1;
                        __builtin_amdgcn_s_waitcnt(0);  // (otherwise the compiler waits for this load at the test above, in EVERY step -- behind the store)
                    }
                }
            }
        }
        if (dbg && lane == 0) {
            const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t_job0;
            atomicMin(dbg + 0, dt);
            atomicMax(dbg + 1, dt);
            atomicAdd(dbg + 2, dt);
            atomicAdd(dbg + 3, 1ull);
            atomicAdd(dbg + 4, t_parked);
            atomicAdd(dbg + 5, n_parks);
            if (s == 0) {
                atomicMin(dbg + 6, dt);
                atomicMax(dbg + 7, dt);
            }
            const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7u;  // HW_REG_XCC_ID[3:0]: which of the eight XCDs ran the job
            atomicAdd(dbg + 8 + 2 * xcc, dt - t_parked);  // (awake time)
            atomicAdd(dbg + 9 + 2 * xcc, 1ull);
        }
    }
}

// ---- transposes: the batch's packed 2-bit codes of a (16 columns per u32) and the profile of b (the reference's negated bit planes, two u64
//      per 64 rows: profile.rs:127-132) -> bit planes over the 32 pairs of a group ----

// one wavefront: 32 columns (two code words per pair; lanes 0..31 = the group's pairs for the first word, lanes 32..63 for the second)
__global__ __launch_bounds__(256) void slice_pack_a_kernel(const SliceGroup* __restrict__ groups, const SlicePair* __restrict__ spairs,
                                                           const uint32_t* __restrict__ codes, uint2* __restrict__ A) {
    const SliceGroup grp = groups[blockIdx.x];
    const int lane = (int)(threadIdx.x & 63), q = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));  // columns [32 q, 32 q + 32)
    if (q * 32 >= grp.n) return;
    const int p = lane & 31, half = lane >> 5, word = 2 * q + half;
    uint32_t w = 0;
    if (p < grp.npairs) {
        const SlicePair sp = spairs[grp.first_pos + p];
        if (word < (sp.n + 15) / 16) w = codes[sp.code_off + word];
    }
    uint32_t a0 = 0, a1 = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const unsigned long long b0 = __ballot((w >> (2 * k)) & 1u), b1 = __ballot((w >> (2 * k + 1)) & 1u);
        if ((lane & 15) == k) {  // lane L < 32 keeps column 32 q + L: code position L & 15 of word L >> 4
            a0 = (lane & 16) ? (uint32_t)(b0 >> 32) : (uint32_t)b0;
            a1 = (lane & 16) ? (uint32_t)(b1 >> 32) : (uint32_t)b1;
        }
    }
    const int c = q * 32 + lane;
    if (lane < 32 && c < grp.n) A[grp.a_off + kPad + c] = make_uint2(a0, a1);
}

// one wavefront: 64 rows (one profile word per pair; lanes 0..31 take the low halves, lanes 32..63 the high halves)
__global__ __launch_bounds__(256) void slice_pack_b_kernel(const SliceGroup* __restrict__ groups, const SlicePair* __restrict__ spairs,
                                                           const uint64_t* __restrict__ prof, uint2* __restrict__ B, int rows_per_strip) {
    const SliceGroup grp = groups[blockIdx.x];
    const int lane = (int)(threadIdx.x & 63), j = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));  // rows [64 j, 64 j + 64)
    if ((long long)j * 64 >= (long long)grp.nstrips * rows_per_strip) return;
    const int p = lane & 31, half = lane >> 5;
    uint32_t x0 = 0, x1 = 0;
    if (p < grp.npairs) {
        const SlicePair sp = spairs[grp.first_pos + p];
        if (j < (sp.m + 63) / 64) {
            const uint64_t p0 = prof[2 * (sp.prof_off + j)], p1 = prof[2 * (sp.prof_off + j) + 1];
            x0 = (uint32_t)(p0 >> (32 * half));
            x1 = (uint32_t)(p1 >> (32 * half));
        }
    }
    uint32_t nb0 = 0, nb1 = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const unsigned long long b0 = __ballot((x0 >> k) & 1u), b1 = __ballot((x1 >> k) & 1u);
        if ((lane & 31) == k) {  // lane L keeps row 64 j + L: bit L & 31 of half L >> 5
            nb0 = half ? (uint32_t)(b0 >> 32) : (uint32_t)b0;
            nb1 = half ? (uint32_t)(b1 >> 32) : (uint32_t)b1;
        }
    }
    B[grp.b_off + (size_t)j * 64 + lane] = make_uint2(nb0, nb1);
}

// cost of pair p = |a_p| + sum over its rows r < |b_p| of (vp - vm) of the captured column (Block::index from the top: block.rs:100-121).
// One wavefront per (group, span of kScoreSpan rows); cost_out is zeroed before the pass and every wavefront adds its share.
constexpr int kScoreSpan = 4096;
__global__ __launch_bounds__(64) void slice_score_kernel(const SliceGroup* __restrict__ groups, const SlicePair* __restrict__ spairs,
                                                         const uint2* __restrict__ V, int32_t* __restrict__ cost_out) {
    const SliceGroup grp = groups[blockIdx.x];
    const int lane = (int)threadIdx.x;
    int my_m = 0, my_n = 0;
    uint32_t my_pair = 0;
    if (lane < grp.npairs) {
        const SlicePair sp = spairs[grp.first_pos + lane];
        my_m = sp.m;
        my_n = sp.n;
        my_pair = sp.pair;
    }
    int max_m = my_m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_m = max(max_m, __shfl_xor(max_m, o));
    const int lo = (int)blockIdx.y * kScoreSpan, hi = min(max_m, lo + kScoreSpan);
    if (lo >= max_m) return;
    int acc = blockIdx.y == 0 ? my_n : 0;
    for (int r0 = lo; r0 < hi; r0 += 64) {
        const int r = r0 + lane;
        uint2 v = make_uint2(0u, 0u);
        if (r < hi) v = V[grp.b_off + r];
        for (int p = 0; p < grp.npairs; ++p) {
            const int mp = __shfl(my_m, p);
            const bool in = r < mp;
            const int pos = __popcll(__ballot(in && ((v.x >> p) & 1u))), neg = __popcll(__ballot(in && ((v.y >> p) & 1u)));
            if (lane == p) acc += pos - neg;
        }
    }
    if (lane < grp.npairs && acc != 0) atomicAdd(&cost_out[my_pair], acc);
}

}  // namespace slice
}  // namespace pa
