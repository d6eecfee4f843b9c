// strip_kernel.hpp -- the bit-parallel (Myers/Hyyro) block-DP kernel for gfx950 (MI355X).
//
// What it replaces: pa_bitpacking::simd::{compute,fill} and the column loop of
// compute_block_of_rows (reference pa-bitpacking/src/simd.rs:98-315,326-547; myers.rs:27-91).
//
// MI355X-first design (not the CPU's 8-lane AVX2 strip):
//  * A *strip* is one 64-lane wavefront.  Lane l owns K subwords of 32 DP rows (K = 1, 2, 4, 8; subword s = l*K + k is
//    half s&1 of reference word word0 + s/2), so a strip covers 2048*K rows = 32*K reference words.  K = 1 makes every
//    Myers op one VALU instruction (lowest latency per column: one long pair); larger K amortises the 11 per-lane
//    "plumbing" instructions over 12 per subword (23 / 35 / 59 / 107 instructions per step) -- the kernel is bound by
//    VALU issue, so instructions per DP cell are the cost.
//  * Anti-diagonal skew inside the wave: at step t lane l processes column t-l.  The horizontal
//    delta (2 bits) and the column's 2-bit base code travel lane->lane+1 in ONE packed register
//    through a DPP `wave_shr:1` move -- no LDS, no barrier.
//  * The bottom row of strip s (2 bits/column) reaches strip s+1 through 8-byte granules of 32 columns each in global
//    memory.  A granule needs no tag: every 2-bit delta field is stored +1 (1..3), so a written granule is never
//    zero; the consumer hands every granule back zeroed, so the buffer is cleared only once.
//      - strip_kernel: strips of a rectangle run concurrently, one wavefront each, chained through agent-scope relaxed
//        atomic stores / polled loads ("the data is the flag"; MI355X_MICROARCH.md, handoff R2; no fences, no L2
//        write-back).  Jobs are claimed through an atomic ticket in producer-before-consumer order => forward
//        progress without assuming dispatch order.  Every spin is bounded.
//      - pair_kernel: ONE wavefront runs all strips of a pair top to bottom (two granule rows, ping-pong,
//        workgroup-scope accesses that stay in the L2).  Nothing polls; this is the shape of big batches.
//      - rect_kernel: one rectangle of the A*PA2 engine per launch, described by kernel arguments, results and the
//        completion word in host-mapped memory.
//  * Variants (templates): FILL stores every column's V (traceback re-fills), SCATTER reads the four-mask profile of the
//    semi-global search, CKPT stores the V column after every 256th column (sparse blocks of the batched traceback);
//    banded pairs give every strip its own column window (StripJob::col0 / n / hin_n / vsum_out).
//  * HBM traffic is tiny by construction (0.25 B/column of `a`, 16 B/word of profile, 32 B/word of v,
//    0.5 B/column of h per strip boundary): the kernel is integer-VALU-issue bound, not HBM bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef PA_STRIP_DEBUG
// Probe builds only (tools/strip_probe.hip): lane 0 publishes progress markers to host-visible memory.
extern __device__ unsigned int* g_pa_dbg;
#define PA_DBG(slot, value)                                                                               \
    do {                                                                                                    \
        if (g_pa_dbg && (threadIdx.x & 63) == 0)                                                            \
            __hip_atomic_store(g_pa_dbg + (slot), (unsigned int)(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
    } while (0)
#else
#define PA_DBG(slot, value) do { } while (0)
#endif

// Scheduling fence between the phases of a tall (K >= 8) Myers step.  Runs of "simple" VALU ops (VOP2 logic/add,
// 3-VGPR v_bitop3) issue at about twice the rate of ops with carries, SGPR operands, DPP or v_alignbit/v_bfe, but only
// when they are not interleaved with those (profiles/r01_runs/issue_probe3.log); keeping the phases apart is worth ~5 %
// at K = 8 and nothing or less below.
#define PA_PHASE()                                        \
    do {                                                  \
        if (K >= 8) __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

namespace pa {

// One strip job = one wavefront.  All pointers are device pointers.
struct StripJob {
    const uint32_t* a_codes;  // 2-bit base codes (A0 C1 G2 T3) of the WHOLE sequence a, 16 per u32: column i at bits
                              // 2*(i%16)+{0,1} of word i/16; the rectangle starts at absolute column `col0`
    const uint32_t* b_prof;   // BitProfile of b: u32 view of (nb0:u64, nb1:u64) per 64-row word (profile.rs:112-133);
                              // scatter kernels: u32 view of [u64; 4] match masks per word (profile.rs:25-75)
    uint32_t* v;              // V(p:u64, m:u64) per 64-row word, u32 view, updated in place (encoding.rs:5-6)
    const uint64_t* hin_gran; // granules from the strip above (32 columns each, see kGranuleBias), or nullptr
    const uint8_t* hin_arr;   // top-row deltas, one byte per ABSOLUTE column (bit0 = +1, bit1 = -1), or nullptr => all +1
    uint64_t* hout_gran;      // granules for the strip below, or nullptr
    uint8_t* hout_arr;        // bottom-row deltas out, one byte per ABSOLUTE column, or nullptr
    uint32_t* values;         // fill mode: V of every column, u32 view of values[col][fill_stride] (V each); or nullptr
    int32_t* sum_out;         // *sum_out = sum of bottom-row deltas over the n columns (if non-null)
    int32_t n;                // columns
    int32_t word0;            // first 64-row word of this strip (index into b_prof / v)
    int32_t nlanes;           // real 32-row subwords: 2 * (words in this strip), 2..128K, even (K subwords per lane)
    int32_t fill_stride;      // words per column in `values`
    int32_t fill_word0;       // word index of this strip inside a `values` column
    int32_t exact_tail;       // nlanes<64 only.  1: lanes >= nlanes forward h unchanged, so lane 63 carries the true
                              //    bottom row (needed when the bottom deltas themselves are an output).
                              // 0: lanes >= nlanes run as zero pad rows (b = Bits(0,0), v = V(0,0)) and the sum is
                              //    corrected with their right edge -- the identity the reference's padded tail uses
                              //    (simd.rs:184-225).
    int32_t flags;            // kJobVInitOne: start from V::one() instead of loading v (first column, blocks.rs:163)
    int32_t col0;             // absolute index of the rectangle's first column (into a_codes / hin_arr / hout_arr)
    int32_t tail_rows;        // >= 0: |b|; the reported sum additionally subtracts the right-edge deltas of rows >= |b|
                              //       (Block::index from the bottom, block.rs:110-120), so the host adds 64*words only
                              // < 0: plain sum of the bottom-row deltas
    int32_t k;                // pair_kernel only: subwords per lane of THIS strip (1 = short tail strip, else the kernel's K)
    uint32_t* ckpt;           // CKPT kernels: V column after every 256th column, u32 view of ckpt[c][ckpt_stride] (V each),
                              // c = (column + 1) / 256; the sparse blocks of the traceback (blocks.rs:322-339).  Or nullptr
    int32_t ckpt_stride;      // words per checkpoint column
    int32_t hin_n;            // banded strips: only the first hin_n columns (a multiple of 32, or >= n) have granules from
                              // the strip above; the rest of the top row is +1 (outside the band).  0 = all of them
    int32_t* vsum_out;        // banded strips: atomically add the sum of this strip's right-edge vertical deltas (rows
                              // below tail_rows excluded); cost = n + sum over the strips of a pair.  Or nullptr
};
enum : int32_t {
    kJobVInitOne = 1,
    kJobLog = 16,      // diagnostics (cost-only strips): `values` points at 8 words that receive HW_ID, XCC_ID, start and end
                       // time and the number of chunks that had to poll (PA_STRIP_WAVELOG)
    kJobPace = 32,     // top strip of a pair in a chained batch (cost-only): `ckpt` points at a u64 counter that every such strip
                       // increments once per chunk, `ckpt_stride` = number of such strips; a strip more than kPaceLead chunks
                       // ahead of the average naps until the others caught up (bounded), one that finished adds kPaceDone so
                       // that nobody waits for it
    kJobRotatePrio = 2,  // chained strips sharing SIMDs: rotate the issue priority chunk by chunk (see the chunk loop)
};
static_assert(sizeof(StripJob) == 136, "StripJob layout");

enum : uint32_t {
    PA_ERR_NONE = 0,
    PA_ERR_SPIN_TIMEOUT = 1,  // a producer strip never delivered its granule
};

constexpr int kPaceLead = 6;
constexpr unsigned long long kPaceDone = 1ull << 40;
constexpr uint64_t kSpinTimeoutTicks = 30ull * 100000000ull;  // 30 s of the 100 MHz wall clock: a lost producer ends the
                                                                 // wave with PA_ERR_SPIN_TIMEOUT instead of hanging the GPU

// Every pointer of a StripJob is device global memory; say so, or the compiler emits flat_* accesses whose
// out-of-order return forces s_waitcnt vmcnt(0) everywhere.
#define PA_GLOBAL __attribute__((address_space(1)))
typedef const PA_GLOBAL uint32_t* gcu32;
typedef const PA_GLOBAL uint8_t* gcu8;
typedef const PA_GLOBAL uint64_t* gcu64;
typedef PA_GLOBAL uint32_t* gu32;
typedef PA_GLOBAL uint8_t* gu8;
typedef PA_GLOBAL uint64_t* gu64;
typedef PA_GLOBAL int32_t* gi32;

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t old_, uint32_t src) {
    // v_mov_b32_dpp wave_shr:1 ; lane 0 has no source lane and keeps `old_`.
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old_, (int)src, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// A scalar value gets a live range of its own (see apa2_kernel.hpp, where the note on descriptors fetched by one wide scalar load is).
template <class T>
__device__ __forceinline__ T own_sgpr(T x) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "scalar or pointer");
    asm volatile("" : "+s"(x));
    return x;
}

// Wavefront sums, minima and prefix sums on the DPP data path: six dependent VALU instructions whose second operand comes from another
// lane of the row (row_shr:1/2/4/8: a scan inside each row of 16), then from lane 15 of the row before (row_bcast:15 into rows 1 and 3)
// and from lane 31 (row_bcast:31 into rows 2 and 3).  __shfl_xor / __shfl_up compile to ds_bpermute_b32 -- six dependent round trips
// through the LDS crossbar (~0.7 us for one sum of a lone wavefront, measured in the traceback's DT levels).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int32_t dpp_or_zero(int32_t x) {  // lanes without a source lane (and rows outside ROW_MASK) read 0
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int32_t dpp_or_self(int32_t x) {  // ... read their own value
    return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int32_t wave_scan_add(int32_t x) {  // inclusive prefix sum over the 64 lanes
    x += dpp_or_zero<0x111, 0xf>(x);
    x += dpp_or_zero<0x112, 0xf>(x);
    x += dpp_or_zero<0x114, 0xf>(x);
    x += dpp_or_zero<0x118, 0xf>(x);
    x += dpp_or_zero<0x142, 0xa>(x);
    x += dpp_or_zero<0x143, 0xc>(x);
    return x;
}
__device__ __forceinline__ int32_t wave_add(int32_t x) {  // sum over the 64 lanes, wavefront-uniform
    return __builtin_amdgcn_readlane(wave_scan_add(x), 63);
}
__device__ __forceinline__ int32_t wave_min(int32_t x) {  // minimum over the 64 lanes, wavefront-uniform
    int32_t y;
    y = dpp_or_self<0x111, 0xf>(x); x = y < x ? y : x;
    y = dpp_or_self<0x112, 0xf>(x); x = y < x ? y : x;
    y = dpp_or_self<0x114, 0xf>(x); x = y < x ? y : x;
    y = dpp_or_self<0x118, 0xf>(x); x = y < x ? y : x;
    y = dpp_or_self<0x142, 0xa>(x); x = y < x ? y : x;
    y = dpp_or_self<0x143, 0xc>(x); x = y < x ? y : x;
    return __builtin_amdgcn_readlane(x, 63);
}

// Packed pipeline register X:  bit31 = h.p (delta +1), bit30 = h.m (delta -1), bits[1:0] = base code, rest 0.
// One Myers step on a lane of K 32-row subwords (myers.rs:27-55 on a 32K-bit word; eq from profile.rs:141-144).
// `acc` collects the lane's outgoing deltas delayed by one step: newest column in bits [1:0] = (p,m),
// i.e. after 16 steps column k of the chunk sits at bit 31-2k (p) / 30-2k (m).
// Instruction budget: 10 "plumbing" ops per lane-step (accumulate, cross-lane shift, field extracts, repack) + 12 per
// subword (+1 for the carry-in of subword 0).  The chip is VALU-issue bound at ~4 cycles per wave instruction per SIMD
// (profiles/r01_runs/issue_probe*.log), so instructions per DP cell is THE cost: K = 1, 2, 4 cost 23, 17.5, 14.75
// instructions per 2048 cells.  Larger K trades per-step latency (single-pair speed) for throughput.
// SCATTER: eq comes from a ScatterProfile (profile.rs:25-75): four match masks per word, selected by the text code
// (A0 C1 T2 G3), so pattern wildcards (N, *, Y, R) work.
//
// LDSEQ (tall strips of big batches): the four possible `eq` words of every subword sit in the wavefront's LDS slice
// ([code][K/4][lane] x 16 bytes, filled once per strip), and the packed register carries, next to the delta of column c,
// the LDS offset of the code of column c + 1: each step issues the K/4 ds_read_b128 of the NEXT step's eq (`eqn`) and
// consumes the ones the previous step fetched.  That moves 2 of the 12 VALU instructions per subword (and the two code
// extracts) to the LDS port, which issues beside the VALU: 107 -> 89 VALU instructions per K = 8 step.
template <int K>
struct LdsEq {
    static constexpr int kCodeShift = K > 8 ? 12 : (K > 4 ? 11 : 10);  // one code's eq words of all 64 lanes: 64 * 4K bytes, rounded up to a power of two
    static constexpr uint32_t kCodeMask = 3u << kCodeShift;
    static constexpr uint32_t kWaveBytes = 4u << kCodeShift;
};
typedef uint32_t pa_u32x4 __attribute__((ext_vector_type(4)));
typedef pa_u32x4 __attribute__((address_space(3))) pa_lds_u32x4;
typedef pa_lds_u32x4* pa_lds4;

template <int K>
__device__ __forceinline__ void lds_eq_fetch(uint32_t addr, uint32_t (&e)[K]) {
#pragma unroll
    for (int i = 0; i < K / 4; ++i) {
        const pa_u32x4 t = *(pa_lds4)(uintptr_t)(addr + 1024u * (uint32_t)i);
        e[4 * i] = t.x;
        e[4 * i + 1] = t.y;
        e[4 * i + 2] = t.z;
        e[4 * i + 3] = t.w;
    }
}

// HALF (K = 1, strips of <= 32 lanes): the strip lives in lanes 32..63 and the column input enters at lane 32 (written into
// lane 31's X, which the DPP shift then delivers), so the skew is 32 steps instead of 64.
template <int K, bool PRED, bool PASS, bool SCATTER, bool LDSEQ = false, bool HALF = false>
__device__ __forceinline__ void myers_step(uint32_t s_x, uint32_t& X, uint32_t (&vp)[K], uint32_t (&vm)[K],
                                           const uint32_t (&nb0)[K], const uint32_t (&nb1)[K], const uint32_t (&nb2)[K],
                                           const uint32_t (&nb3)[K], uint32_t& acc, bool active, bool pass_lane, uint32_t k40,
                                           uint32_t k80, uint32_t (&eqn)[K], uint32_t lds_lane, uint32_t kcm) {
    acc = __builtin_amdgcn_alignbit(acc, X, 30);  // (acc << 2) | (X >> 30)
    uint32_t Xin;
    if (HALF) {
        uint32_t Xw = X;
        asm("v_writelane_b32 %0, %1, 31" : "+v"(Xw) : "s"(s_x));
        Xin = dpp_wave_shr1(Xw, Xw);
    } else {
        Xin = dpp_wave_shr1(s_x, X);
    }
    uint32_t a0 = 0, a1 = 0;
    uint32_t eq[K], vx[K], sm[K], hp[K], hm[K];
    if (LDSEQ) {
#pragma unroll
        for (int k = 0; k < K; ++k) eq[k] = eqn[k];
        lds_eq_fetch<K>((Xin & kcm) | lds_lane, eqn);
    } else {
        a0 = (uint32_t)__builtin_amdgcn_sbfe((int)Xin, 0, 1);
        a1 = (uint32_t)__builtin_amdgcn_sbfe((int)Xin, 1, 1);
    }
    const uint32_t hm0 = (Xin >> 30) & 1u;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (LDSEQ) {
        } else if (SCATTER) {
            const uint32_t e01 = __builtin_amdgcn_bitop3_b32(a0, nb1[k], nb0[k], 0xCA);  // a0 ? mask[1] : mask[0]
            const uint32_t e23 = __builtin_amdgcn_bitop3_b32(a0, nb3[k], nb2[k], 0xCA);
            eq[k] = __builtin_amdgcn_bitop3_b32(a1, e23, e01, 0xCA);
        } else {
            eq[k] = __builtin_amdgcn_bitop3_b32(a0, nb0[k], a1 ^ nb1[k], 0x28);  // (a0 ^ nb0) & (a1 ^ nb1) in two ops
        }
        vx[k] = eq[k] | vm[k];
    }
    eq[0] |= hm0;
    PA_PHASE();
    // (eq & vp) + vp over the whole 32K-bit word
    if (K == 1) {
        sm[0] = (eq[0] & vp[0]) + vp[0];
    } else {
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uint32_t co;
            sm[k] = __builtin_addc(eq[k] & vp[k], vp[k], carry, &co);
            carry = co;
        }
    }
    PA_PHASE();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t hx = (sm[k] ^ vp[k]) | eq[k];
        hp[k] = vm[k] | ~(hx | vp[k]);
        hm[k] = vp[k] & hx;
    }
    PA_PHASE();
    // two bit-field inserts (v_bitop3 each); bits 29:2 of X are always 0, so keeping Xin's other bits is exact.
    // k40 / k80 are opaque to the optimizer on purpose, otherwise it re-expands this into 5 ops.
    const uint32_t xm = __builtin_amdgcn_bitop3_b32(k40, hm[K - 1] >> 1, Xin, 0xCA);  // k40 ? (hm >> 1) : Xin
    uint32_t Xo = __builtin_amdgcn_bitop3_b32(k80, hp[K - 1], xm, 0xCA);             // k80 ? hp : xm
    if (PASS) Xo = pass_lane ? Xin : Xo;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t hp2 = __builtin_amdgcn_alignbit(hp[k], k == 0 ? Xin : hp[k - 1], 31);  // (hp << 1) | carry-in
        const uint32_t hm2 = k == 0 ? ((hm[0] << 1) | hm0) : __builtin_amdgcn_alignbit(hm[k], hm[k - 1], 31);
        const uint32_t nvp = __builtin_amdgcn_bitop3_b32(hm2, vx[k], hp2, 0xF1);  // hm2 | ~(vx | hp2)
        const uint32_t nvm = hp2 & vx[k];
        if (PRED) {
            vp[k] = active ? nvp : vp[k];
            vm[k] = active ? nvm : vm[k];
        } else {
            vp[k] = nvp;
            vm[k] = nvm;
        }
    }
    X = Xo;
}

// One chunk = 32 columns = 32 unrolled steps.  Lane j (< 32) of XS carries the packed pipeline input of column 32q+j.
// The lagged accumulator of steps 0..15 is acc_lo, of steps 16..31 acc_hi (static, so no register moves).
// `lane` is the LOGICAL lane (HALF: physical lane - 32, negative for the idle half).
template <int K, bool PRED, bool PASS, bool FILL, bool SCATTER, bool CKPT, bool LDSEQ = false, bool HALF = false>
__device__ __forceinline__ void run_chunk(const StripJob& job, int q, uint32_t XS, uint32_t& X, uint32_t (&vp)[K],
                                          uint32_t (&vm)[K], const uint32_t (&nb0)[K], const uint32_t (&nb1)[K],
                                          const uint32_t (&nb2)[K], const uint32_t (&nb3)[K], uint32_t& acc_lo,
                                          uint32_t& acc_hi, int lane, bool pass_lane, gu32 vout, uint32_t k40, uint32_t k80,
                                          uint32_t (&eqn)[K], uint32_t lds_lane, uint32_t kcm) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const uint32_t s_x = (uint32_t)__builtin_amdgcn_readlane((int)XS, j);
        const int col = q * 32 + j - lane;
        const bool active = PRED ? ((unsigned)col < (unsigned)job.n) : true;
        myers_step<K, PRED, PASS, SCATTER, LDSEQ, HALF>(s_x, X, vp, vm, nb0, nb1, nb2, nb3, j < 16 ? acc_lo : acc_hi, active, pass_lane,
                                                        k40, k80, eqn, lds_lane, kcm);
        if (FILL) {
            if (active) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int sub = lane * K + k;
                    if ((unsigned)sub < (unsigned)job.nlanes) {
                        gu32 dst = vout + (size_t)col * (size_t)job.fill_stride * 4 + (size_t)(sub >> 1) * 4 + (sub & 1);
                        dst[0] = vp[k];
                        dst[2] = vm[k];
                    }
                }
            }
        }
        if (CKPT) {
            // the one lane (if any) whose column just completed a 256-column block stores its V words
            const int t = (q * 32 + j + 1) & 255;
            if (lane == t && active) {
                gu32 base = (gu32)job.ckpt + ((size_t)((col + 1) >> 8) * (size_t)job.ckpt_stride + (size_t)job.word0) * 4;
                if (K == 1) {
                    if (lane < job.nlanes) {
                        base[(size_t)(lane >> 1) * 4 + (lane & 1)] = vp[0];
                        base[(size_t)(lane >> 1) * 4 + 2 + (lane & 1)] = vm[0];
                    }
                } else {
                    // a lane owns K/2 whole V words: one 16-byte store each
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    typedef PA_GLOBAL u32x4* gu32x4;
#pragma unroll
                    for (int i = 0; i < K / 2; ++i) {
                        if (lane * K + 2 * i < job.nlanes) {
                            const u32x4 word = {vp[2 * i], vp[2 * i + 1], vm[2 * i], vm[2 * i + 1]};
                            *(gu32x4)(base + ((size_t)lane * (K / 2) + i) * 4) = word;
                        }
                    }
                }
            }
        }
    }
}

// Hand-off granule: 32 columns x 2 bits, low word = columns 0..15, high word = 16..31; inside a word column k sits at
// bits 31-2k (p) / 30-2k (m).  (p,m) is never (1,1), so adding 1 to every field never carries and makes each field
// non-zero: a written granule is distinguishable from the zeroed buffer without a tag.
constexpr uint64_t kGranuleBias = 0x5555555555555555ull;

template <bool LOCAL>
__device__ __forceinline__ uint64_t load_granule(gcu64 g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, LOCAL ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT);
}

// Resolve the (possibly prefetched) granule of chunk q; polls when the producer is not there yet.
// Returns false (after a bounded wall-clock time) if the producer never delivered.
template <bool LOCAL>
__device__ __forceinline__ bool resolve_granule(gcu64 g, uint64_t pre, int q, uint32_t& lo, uint32_t& hi) {
    uint32_t l = rfl((uint32_t)pre);
    if (l == 0u) {  // slow path: the producer is not there yet
        const uint64_t t0 = wall_clock64();
        uint32_t spins = 0;
        do {
            // back off while the wait is long (a banded strip may wait milliseconds for the diagonal to reach it): a
            // polling wavefront shares its SIMD's issue slots with working ones.  Capped at ~3 us so that a chain of
            // strips does not accumulate start-up delay.
            const uint32_t naps = spins < 8u ? 1u : (spins < 64u ? 4u : 16u);
            for (uint32_t k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(8);
            pre = load_granule<LOCAL>(g + q);
            l = rfl((uint32_t)pre);
            if ((++spins & 255u) == 0 && wall_clock64() - t0 > kSpinTimeoutTicks) break;
        } while (l == 0u);
    }
    const uint32_t h = rfl((uint32_t)(pre >> 32));
    lo = l - 0x55555555u;
    hi = h - 0x55555555u;
    return l != 0u;
}

// Rarely taken paths of chained batches, kept OUT of line: inlined, their loads share registers with the chunk loop's and the
// compiler protects those with an s_waitcnt vmcnt(0) in front of every granule store -- one memory round trip per chunk on the
// critical path of every chain (measured: one pair 5.45 -> 6.39 ms).
__device__ __attribute__((noinline)) void pace_top_strip(const uint32_t* counter, int32_t strips, int q) {
    typedef PA_GLOBAL unsigned long long* gull;
    const gull ctr = (gull)counter;
    const int lane = (int)(threadIdx.x & 63);
    unsigned long long g = 0;
    if (lane == 0) g = __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t glo = rfl((uint32_t)g), ghi = rfl((uint32_t)(g >> 32));
    const unsigned long long mine = (unsigned long long)(q > kPaceLead ? q - kPaceLead : 0) * (unsigned long long)strips;
    // ahead of the average by more than kPaceLead chunks: nap (a chunk is ~5 us of work), at most ~8 chunks' worth
    for (int nap = 0; nap < 16 && (((unsigned long long)ghi << 32) | glo) < mine; ++nap) {
        __builtin_amdgcn_s_sleep(127);
        if (lane == 0) g = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        glo = rfl((uint32_t)g);
        ghi = rfl((uint32_t)(g >> 32));
    }
}
// Process one strip.  On a spin timeout the error word is set and the strip stops early.
// K = 32-row subwords per lane: the strip covers 64*K subwords = 32*K reference words.
// LOCAL: the granules are produced and consumed by the SAME wavefront (pair_kernel, trace_kernel): workgroup-scope
// accesses, so the rows stay in the L2 instead of being written through / fetched around it 8 bytes at a time.
// LDSEQ: eq words come from the wavefront's LDS slice at byte offset `lds_wave` (LdsEq<K>::kWaveBytes, aligned to its size).
// HALF (K = 1 and nlanes <= 32 only): the strip occupies lanes 32..63, the pipeline is 32 steps deep instead of 64, so the
// strip takes C + 1 chunks instead of C + 2 -- 10 % of a 256-column block of the engine, of a traceback re-fill.
// NOPASS: the caller never asks for an exact bottom row of a partial strip (job.exact_tail is 0 whenever nlanes < 64 K): the
// pass-through chunk variants are not compiled (half the unrolled code of the strip; apa2_kernel.hpp holds four strip heights).
// TAP (apa2_full_kernel.hpp; K <= 2): job.hout_arr receives the horizontal deltas that LEAVE logical lane `tap_lane` (>= 0) -- a row
// INSIDE the strip -- instead of the bottom row: incremental doubling stores the deltas of row j_h (blocks.rs:342-469), and with the
// tap the rows above and below j_h run as ONE strip (HMode::Output + HMode::Input, or Update + Input, of blocks.rs:406-468).  Lane l's
// accumulators hold columns 32q - 1 - l .. 32q + 30 - l after chunk q (see run_chunk), so the bytes go out unaligned, per column.
template <int K, bool FILL, bool SCATTER, bool CKPT = false, bool LOCAL = false, bool LDSEQ = false, bool HALF = false, bool NOPASS = false, bool TAP = false>
__device__ __forceinline__ void run_strip(const StripJob& job, uint32_t* err, uint32_t lds_wave = 0, int tap_lane = -1) {
    static_assert(!LDSEQ || (K >= 4 && !SCATTER && !FILL), "LDSEQ: tall cost-only strips");
    static_assert(!HALF || (K == 1 && !CKPT && !LDSEQ), "HALF: short K = 1 strips without checkpoints");
    constexpr int kGranScope = LOCAL ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT;
    constexpr int kDrain = HALF ? 1 : 2;               // chunks between a column entering the strip and leaving lane 63
    const int plane = (int)(threadIdx.x & 63);         // physical lane: builds the chunk inputs, publishes
    const int lane = HALF ? plane - 32 : plane;        // logical lane: rows 32K*lane .. of the strip (negative: idle)
    const int n = job.n;
    const int C = (n + 31) >> 5;  // 32-column chunks == granules
    const bool pass_lane = lane * K >= job.nlanes;  // the whole lane is below the rectangle

    uint32_t vp[K], vm[K], nb0[K], nb1[K], nb2[K], nb3[K];
    const gcu32 g_prof = (gcu32)job.b_prof;
    const gu32 g_v = (gu32)job.v;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int sub = lane * K + k;  // subword of this strip: word = word0 + sub/2, half = sub&1
        const int word = job.word0 + (sub >> 1), half = sub & 1;
        vp[k] = vm[k] = nb0[k] = nb1[k] = nb2[k] = nb3[k] = 0;
        if ((unsigned)sub < (unsigned)job.nlanes) {
            if (SCATTER) {  // [B; 4] per word
                nb0[k] = g_prof[word * 8 + half];
                nb1[k] = g_prof[word * 8 + 2 + half];
                nb2[k] = g_prof[word * 8 + 4 + half];
                nb3[k] = g_prof[word * 8 + 6 + half];
            } else {
                nb0[k] = g_prof[word * 4 + half];
                nb1[k] = g_prof[word * 4 + 2 + half];
            }
            if (job.flags & kJobVInitOne) {
                vp[k] = 0xFFFFFFFFu;
                vm[k] = 0u;
            } else if (TAP && job.values != nullptr) {
                // init_v_with_overlap (blocks.rs:753-767) folded into the strip: the left edge is the PREVIOUS block's column
                // (`values`, same word indexing) where that block has rows (words [fill_word0, fill_stride)), V::one elsewhere
                const gcu32 g_vs = (gcu32)job.values;
                const bool in_src = word >= job.fill_word0 && word < job.fill_stride;
                vp[k] = 0xFFFFFFFFu;
                vm[k] = 0u;
                if (in_src) {
                    vp[k] = g_vs[word * 4 + half];
                    vm[k] = g_vs[word * 4 + 2 + half];
                }
            } else {
                vp[k] = g_v[word * 4 + half];
                vm[k] = g_v[word * 4 + 2 + half];
            }
        }
    }
    gu32 vout = nullptr;
    if (FILL) vout = (gu32)job.values + (size_t)job.fill_word0 * 4;
    uint32_t eqn[K];
    uint32_t kcm = LdsEq<K>::kCodeMask;
    const uint32_t lds_lane = lds_wave + 16u * (uint32_t)plane;
#pragma unroll
    for (int k = 0; k < K; ++k) eqn[k] = 0;
    if (LDSEQ) {
        asm volatile("" : "+v"(kcm));  // a VGPR operand: gfx9 VOP3 takes no literal
        // eq of code c (A0 C1 G2 T3) against the negated bit planes: (c0 ^ nb0) & (c1 ^ nb1), profile.rs:141-144
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t m0 = (c & 1) ? 0xFFFFFFFFu : 0u, m1 = (c & 2) ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int i = 0; i < K / 4; ++i) {
                pa_u32x4 t;
                t.x = (m0 ^ nb0[4 * i]) & (m1 ^ nb1[4 * i]);
                t.y = (m0 ^ nb0[4 * i + 1]) & (m1 ^ nb1[4 * i + 1]);
                t.z = (m0 ^ nb0[4 * i + 2]) & (m1 ^ nb1[4 * i + 2]);
                t.w = (m0 ^ nb0[4 * i + 3]) & (m1 ^ nb1[4 * i + 3]);
                *(pa_lds4)(uintptr_t)(lds_lane + ((uint32_t)c << LdsEq<K>::kCodeShift) + 1024u * (uint32_t)i) = t;
            }
        }
    }

    uint32_t X = 0, acc_lo = 0, acc_hi = 0;
    int32_t sum = 0;
    uint32_t k40 = 0x40000000u, k80 = 0x80000000u;  // see myers_step; kept in VGPRs (SGPR operands halve the issue rate)
    asm volatile("" : "+v"(k40), "+v"(k80));
    const int cj = plane & 15;
    const bool upper = (plane & 16) != 0;          // lanes 16..31 build columns 16..31 of the chunk
    const uint32_t sh = 2u * (uint32_t)cj;
    const bool exact_tail = !NOPASS && job.exact_tail != 0 && job.nlanes < (HALF ? 32 : 64) * K;

    // Per-chunk inputs.  The packed sequence is read with SCALAR loads (constant address space -> s_load, tracked by
    // lgkmcnt, so it never waits behind the granule stores); the granule and the optional top-row bytes are vector
    // loads issued one chunk ahead.  Everything is branch-free so the compiler can use counted s_waitcnt.
    typedef const __attribute__((address_space(4))) uint32_t* ccu32;
    const ccu32 c_codes = (ccu32)job.a_codes;
    const gcu8 g_hin = (gcu8)job.hin_arr;
    const gcu64 g_gran = (gcu64)job.hin_gran;
    const bool has_hin = job.hin_arr != nullptr;
    const bool has_gran = job.hin_gran != nullptr;
    const gcu8 hin_src = has_hin ? g_hin : (gcu8)job.a_codes;
    const gcu64 gran_src = has_gran ? g_gran : (gcu64)job.a_codes;
    const int Cm1 = C > 0 ? C - 1 : 0;
    const int last_word = (job.col0 + n - 1) >> 4;  // last valid dword of a_codes for this rectangle
    // LDSEQ: the pipeline register carries the LDS offset of the NEXT column's code, and a lane fetches the eq words of its first
    // column one step before it gets there, from whatever its idle predecessor passes down -- so every lane starts out passing
    // column 0's code (an idle lane keeps bits 29:0 of what it receives).  With X = 0 lanes >= 1 took eq('A') for column 0.
    if (LDSEQ && n > 0) X = ((c_codes[job.col0 >> 4] >> (2u * ((unsigned)job.col0 & 15u))) & 3u) << LdsEq<K>::kCodeShift;
    // raw code words of columns col0+32q .. +31 (three dwords cover any alignment of col0); shifted only at decode time
    struct RawCodes {
        uint32_t w0, w1, w2;
    };
    auto load_codes = [&](int q) -> RawCodes {
        const int c0 = job.col0 + 32 * (q < Cm1 ? q : Cm1);
        const int i0 = c0 >> 4;
        const int i1 = i0 + 1 < last_word ? i0 + 1 : last_word;
        const int i2 = i0 + 2 < last_word ? i0 + 2 : last_word;
        return RawCodes{c_codes[i0], c_codes[i1], c_codes[i2]};
    };
    auto decode_codes = [&](const RawCodes& r, int q) -> uint64_t {  // column k of the chunk at bits 2k+1:2k
        const unsigned s2 = 2u * (unsigned)((job.col0 + 32 * (q < Cm1 ? q : Cm1)) & 15);
        const uint64_t lo64 = (uint64_t)r.w0 | ((uint64_t)r.w1 << 32);
        return s2 == 0 ? lo64 : ((lo64 >> s2) | ((uint64_t)r.w2 << (64 - s2)));
    };
    auto load_hin_byte = [&](int q) -> uint32_t {  // lanes 0..31: top delta byte of column 32q + lane
        int c = 32 * q + (plane & 31);
        c = c < n ? c : n - 1;
        return (uint32_t)hin_src[has_hin ? job.col0 + c : 0];
    };
    auto load_gran = [&](int q) -> uint64_t {
        const int qq = q < Cm1 ? q : Cm1;
        return __hip_atomic_load(gran_src + (has_gran ? qq : 0), __ATOMIC_RELAXED, kGranScope);
    };
    // Publish granule g (columns 32g..32g+31 of the bottom row) from lane 63's lagged accumulators.
    auto publish = [&](int g) {
        const int cols = n - 32 * g;  // >= 1
        uint32_t vlo = (uint32_t)__builtin_amdgcn_readlane((int)acc_lo, 63);
        uint32_t vhi = (uint32_t)__builtin_amdgcn_readlane((int)acc_hi, 63);
        if (cols < 32) {  // ragged last granule only: clear the columns past n (column k at bits 31-2k, 30-2k)
            const int cl = cols >= 16 ? 16 : cols, ch = cols > 16 ? cols - 16 : 0;
            vlo &= cl >= 16 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (2 * cl));
            vhi &= ch == 0 ? 0u : ~(0xFFFFFFFFu >> (2 * ch));
        }
        if (job.hout_gran) {
            if (plane == 0)
                __hip_atomic_store((gu64)job.hout_gran + g, (((uint64_t)vhi << 32) | (uint64_t)vlo) + kGranuleBias,
                                   __ATOMIC_RELAXED, kGranScope);
        }
        if (!TAP && job.hout_arr) {
            if (plane < 32 && plane < cols) {
                const uint32_t tb = ((upper ? vhi : vlo) >> (30 - 2 * cj)) & 3u;  // bit1 = p, bit0 = m
                ((gu8)job.hout_arr)[job.col0 + 32 * g + plane] = (uint8_t)((tb >> 1) | ((tb & 1u) << 1));
            }
        }
        sum += __builtin_popcount(vlo & 0xAAAAAAAAu) + __builtin_popcount(vhi & 0xAAAAAAAAu) -
               __builtin_popcount(vlo & 0x55555555u) - __builtin_popcount(vhi & 0x55555555u);
    };

    const bool extras = (job.flags & (kJobRotatePrio | kJobPace | kJobLog)) != 0;
    const uint32_t prio_slot = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 4);  // HW_ID.wave_id: the slot on this SIMD
    const uint64_t log_t0 = (!FILL && (job.flags & kJobLog)) ? wall_clock64() : 0;
    uint32_t log_polled = 0;
    RawCodes codes_next = load_codes(0);
    uint64_t gran_next = load_gran(0);
    uint32_t hinb_next = load_hin_byte(0);

    // Steps t = 0 .. 32*(C+2)-1; lane l handles column t-l.  The accumulators lag one step, so after chunk q lane 63's
    // (acc_lo, acc_hi) hold the bottom-row deltas of columns 32(q-2) .. 32(q-2)+31 == granule q-2 (HALF: C+1 chunks, q-1).
    // Order inside an iteration: decode inputs (the only waits) -> publish the previous chunk's granule -> prefetch the
    // next chunk -> 32 steps.  Nothing conditional is ever younger than a prefetch, so its wait stays cheap.
    bool alive = true;
    PA_DBG(1, 1);
    // A strip whose only product is `values` (+ v) can stop as soon as its last REAL lane has finished column n - 1; the
    // two drain chunks exist for lane 63's bottom row, which nobody reads then.
    const bool fill_only = FILL && !job.hout_gran && !job.hout_arr && !job.sum_out && !job.vsum_out;
    const int last_lane = (job.nlanes + K - 1) / K;  // real lanes
    const int Q = fill_only ? ((n + last_lane + 31) >> 5 < C + kDrain ? (n + last_lane + 31) >> 5 : C + kDrain) : C + kDrain;
    for (int q = 0; q < Q && alive; ++q) {
        PA_DBG(2, q + 1);
        // ---- decode this chunk's inputs (both prefetched values are consumed here so the only vector-memory wait of
        //      the iteration sits before the publish store, never after it) ----
        {
            uint32_t g_lo = (uint32_t)gran_next, g_hi = (uint32_t)(gran_next >> 32);
            asm volatile("" : "+v"(g_lo), "+v"(g_hi), "+v"(hinb_next));
            gran_next = ((uint64_t)g_hi << 32) | g_lo;
        }
        const uint64_t codes64 = decode_codes(codes_next, q);
        const uint32_t cw = upper ? (uint32_t)(codes64 >> 32) : (uint32_t)codes64;
        uint32_t code = (32 * q + (plane & 31) < n) ? ((cw >> sh) & 3u) : 0u;
        if (LDSEQ) {
            // the pipeline carries the code of the NEXT column (its eq is fetched one step ahead), as an LDS offset
            const unsigned s2 = 2u * (unsigned)((job.col0 + 32 * (q < Cm1 ? q : Cm1)) & 15);
            const uint64_t ahead = (codes64 >> 2) | ((uint64_t)((codes_next.w2 >> s2) & 3u) << 62);
            const uint32_t cwa = upper ? (uint32_t)(ahead >> 32) : (uint32_t)ahead;
            code = (32 * q + (plane & 31) + 1 < n) ? (((cwa >> sh) & 3u) << LdsEq<K>::kCodeShift) : 0u;
            if (q == 0) lds_eq_fetch<K>((((uint32_t)codes64 & 3u) << LdsEq<K>::kCodeShift) | lds_lane, eqn);  // column 0, for lane 0's step 0
        }
        // top delta of this lane's column as (p << 31) | (m << 30); H::one() when there is no top row (blocks.rs:732)
        uint32_t hin2 = has_hin ? (((hinb_next & 1u) << 31) | ((hinb_next & 2u) << 29)) : 0x80000000u;
        if (q < C && has_gran && (job.hin_n == 0 || q * 32 < job.hin_n)) {
            uint32_t glo, ghi;
            if (!LOCAL && !FILL && extras && rfl((uint32_t)gran_next) == 0u) ++log_polled;  // (diagnostics)
            alive = resolve_granule<LOCAL>(g_gran, gran_next, q, glo, ghi);
            hin2 = ((upper ? ghi : glo) << sh) & 0xC0000000u;
            // every granule has exactly one consumer: hand it back zeroed, so the buffer needs clearing only once
            if (plane == 0) __hip_atomic_store((gu64)job.hin_gran + q, (uint64_t)0, __ATOMIC_RELAXED, kGranScope);
        }
        if (!LOCAL && !FILL && extras && (job.flags & kJobRotatePrio)) {
            // rotate the issue priority chunk by chunk, with a phase per wave slot: the SIMD serves the highest priority
            // first and the OLDEST wavefront among equals, which starves the younger wavefronts of a shared SIMD -- and a
            // chain advances at the pace of its most starved strip
            const uint32_t ph = ((uint32_t)q + prio_slot) & 3u;
            if (ph == 0) __builtin_amdgcn_s_setprio(0);
            else if (ph == 1) __builtin_amdgcn_s_setprio(1);
            else if (ph == 2) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
        }
        if (!LOCAL && !CKPT && !FILL && extras && (job.flags & kJobPace) && q < C) pace_top_strip(job.ckpt, job.ckpt_stride, q);
        const uint32_t XS = code | hin2;
        // ---- publish the granule completed by the previous chunk (q-1) ----
        if (q >= kDrain + 1) publish(q - kDrain - 1);
        // ---- prefetch the next chunk ----
        codes_next = load_codes(q + 1);
        gran_next = load_gran(q + 1);
        hinb_next = load_hin_byte(q + 1);

        const bool interior = (q >= kDrain) && (q * 32 + 31 < n);  // every lane is inside [0, n): no predication needed
        // a 256-column block can only complete in a chunk whose first column is 0, 32 or 224 (mod 256): lane l reaches column
        // 255 (mod 256) at step 32q + j = 255 + l.  Only those chunks pay for the checkpoint test.
        const bool ck_chunk = CKPT && job.ckpt != nullptr && ((q & 7) <= 1 || (q & 7) == 7);
#define PA_RUN_CHUNK(PRED_, PASS_, FILL_, CK_) \
    run_chunk<K, PRED_, PASS_, FILL_, SCATTER, CK_, LDSEQ, HALF>(job, q, XS, X, vp, vm, nb0, nb1, nb2, nb3, acc_lo, acc_hi, lane, pass_lane, vout, k40, k80, eqn, lds_lane, kcm)
        if (CKPT && ck_chunk) {
            if (interior) {
                if (exact_tail) PA_RUN_CHUNK(false, true, FILL, CKPT);
                else PA_RUN_CHUNK(false, false, FILL, CKPT);
            } else {
                if (exact_tail) PA_RUN_CHUNK(true, true, FILL, CKPT);
                else PA_RUN_CHUNK(true, false, FILL, CKPT);
            }
        } else {
            if (interior) {
                if (exact_tail) PA_RUN_CHUNK(false, true, FILL, false);
                else PA_RUN_CHUNK(false, false, FILL, false);
            } else {
                if (exact_tail) PA_RUN_CHUNK(true, true, FILL, false);
                else PA_RUN_CHUNK(true, false, FILL, false);
            }
        }
#undef PA_RUN_CHUNK
        if (TAP && tap_lane >= 0) {
            const int tp = HALF ? tap_lane + 32 : tap_lane;
            const uint32_t tlo = (uint32_t)__builtin_amdgcn_readlane((int)acc_lo, tp);
            const uint32_t thi = (uint32_t)__builtin_amdgcn_readlane((int)acc_hi, tp);
            const int c = 32 * q - 1 - tap_lane + (plane & 31);
            if (plane < 32 && c >= 0 && c < n) {
                const uint32_t tb = ((upper ? thi : tlo) >> (30 - 2 * cj)) & 3u;  // bit1 = p, bit0 = m
                ((gu8)job.hout_arr)[job.col0 + c] = (uint8_t)((tb >> 1) | ((tb & 1u) << 1));
            }
        }
    }
    if (alive) publish(C - 1);  // the last granule (completed by chunk Q-1 = C+1)
    if (!LOCAL && !CKPT && !FILL && (job.flags & kJobPace) && plane == 0)
        (void)__hip_atomic_fetch_add((PA_GLOBAL unsigned long long*)job.ckpt, kPaceDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!FILL && (job.flags & kJobLog) && plane == 0) {
        const uint64_t t1 = wall_clock64();
        gu32 lg = (gu32)job.values;
        lg[0] = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
        lg[1] = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        lg[2] = (uint32_t)log_t0;
        lg[3] = (uint32_t)(log_t0 >> 32);
        lg[4] = (uint32_t)t1;
        lg[5] = (uint32_t)(t1 >> 32);
        lg[6] = log_polled;
    }
    PA_DBG(1, 2);
    if (!alive) {
        if (plane == 0) __hip_atomic_store((gu32)err, (uint32_t)PA_ERR_SPIN_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int sub = lane * K + k;
        if ((unsigned)sub < (unsigned)job.nlanes) {
            const int word = job.word0 + (sub >> 1), half = sub & 1;
            g_v[word * 4 + half] = vp[k];
            g_v[word * 4 + 2 + half] = vm[k];
        }
    }
    if (job.vsum_out) {  // banded pairs: value at the strip's bottom-right = value at its top-right + this
        int32_t c = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int sub = lane * K + k;
            if ((unsigned)sub < (unsigned)job.nlanes) {
                uint32_t keep = 0xFFFFFFFFu;
                if (job.tail_rows >= 0) {
                    const int row0 = 64 * job.word0 + 32 * sub;
                    int live = job.tail_rows - row0;  // rows of this subword above |b|
                    live = live < 0 ? 0 : (live > 32 ? 32 : live);
                    keep = live == 32 ? 0xFFFFFFFFu : ((1u << live) - 1u);
                }
                c += __builtin_popcount(vp[k] & keep) - __builtin_popcount(vm[k] & keep);
            }
        }
        c = wave_add(c);
        if (plane == 0) atomicAdd((int32_t*)job.vsum_out, c);
    }
    if (job.sum_out) {
        int32_t c = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int sub = lane * K + k;
            const bool real = (unsigned)sub < (unsigned)job.nlanes;
            // zero pad rows: subtract their right-edge value (simd.rs:202-224)
            if (!exact_tail && !real && lane >= 0) c += __builtin_popcount(vp[k]) - __builtin_popcount(vm[k]);
            if (job.tail_rows >= 0 && real) {
                const int row0 = 64 * job.word0 + 32 * sub;  // first DP row of this subword
                int over = row0 + 32 - job.tail_rows;         // its rows at or beyond |b|
                over = over < 0 ? 0 : (over > 32 ? 32 : over);
                const uint32_t tm = over == 0 ? 0u : (over == 32 ? 0xFFFFFFFFu : ~((1u << (32 - over)) - 1u));
                c += __builtin_popcount(vp[k] & tm) - __builtin_popcount(vm[k] & tm);
            }
        }
        c = wave_add(c);
        sum -= c;
        if (plane == 0) *(gi32)job.sum_out = sum;
    }
    PA_DBG(1, 3);
}

// One wavefront = one strip job, claimed by ticket (jobs are listed producer before consumer, so a consumer's producer
// has always started; no assumption about dispatch order).  Blocks are kStripBlockWaves wavefronts so that the
// dispatcher spreads them over the four SIMDs of a CU evenly: every strip of a pair advances at the pace of the most
// crowded SIMD, so balance is worth more than anything else at 1-4 wavefronts per SIMD.
// `ticket` and `err` must be zeroed (and every hin/hout granule buffer cleared) before the launch.
constexpr int kStripBlockWaves = 4;
constexpr int kStripMaxBlockWaves = 16;  // chained batches of tall strips (k >= 4) beyond one wavefront per SIMD: one workgroup
                                         // per CU (<= 128 VGPRs); k = 1, 2 kernels keep 4-wavefront workgroups
// LDSEQ (K >= 4, cost-only): eq words from LDS, see LdsEq; the launch provides one slice per wavefront of the block.
template <int K, bool FILL, bool SCATTER = false, bool CKPT = false, bool LDSEQ = false>
__global__ __launch_bounds__(64 * (K >= 4 ? kStripMaxBlockWaves : kStripBlockWaves)) void strip_kernel(const StripJob* __restrict__ jobs, int njobs,
                                                   uint32_t* ticket, uint32_t* err) {
    uint32_t t = 0;
    if ((threadIdx.x & 63) == 0) t = atomicAdd(ticket, 1u);
    t = rfl(t);
    PA_DBG(0, t + 1);
    if (t < (uint32_t)njobs) {
        const StripJob job = jobs[t];
        if constexpr (K == 1 && !CKPT) {
            if (job.nlanes <= 32) run_strip<1, FILL, SCATTER, false, false, false, true>(job, err);  // half-wave: one chunk less
            else run_strip<1, FILL, SCATTER, false, false, false>(job, err);
        } else {
            run_strip<K, FILL, SCATTER, CKPT, false, LDSEQ>(job, err, rfl((uint32_t)(threadIdx.x >> 6)) * LdsEq<K>::kWaveBytes);
        }
    }
    PA_DBG(0, 0x1000 + t);
}

// One rectangle of the A*PA2 block engine per launch, described entirely by kernel arguments (no descriptor upload):
// strips are handed out by a ticket (counter[1]) in dependency order, so a strip never waits on one that has not
// started, whatever order the dispatcher picks workgroups in and however many are resident.  `v`, the
// sum, the error word and the completion word live in host-mapped memory: the engine's host thread spins on `done`
// instead of paying a stream synchronisation per 256-column block.
struct RectArgs {
    const uint32_t* a_codes;
    const uint32_t* b_prof;
    uint32_t* v;            // biased so that it is indexed by absolute word (host-mapped)
    const uint8_t* hin_arr;
    uint8_t* hout_arr;
    uint64_t* gran;         // (S-1) rows of gran_stride granules, all zero between launches
    uint64_t gran_stride;
    int32_t* sum_out;       // host-mapped
    uint32_t* err;          // host-mapped
    uint32_t* done;         // host-mapped: receives `seq` when every strip has finished
    uint32_t* counter;      // device: [0] strips finished so far, [1] the strip ticket; both left at zero
    int32_t n, col0, w0, w1, exact_end;
    uint32_t seq;
    uint32_t* values;       // FILL: V of every column, values[col][fill_stride] (host-mapped: the traceback's re-fill reads it on the host)
    int32_t fill_stride;
};

template <int K, bool FILL = false>
__global__ __launch_bounds__(64) void rect_kernel(RectArgs r) {
    const int S = (int)gridDim.x;
    const int lane = (int)(threadIdx.x & 63);
    uint32_t tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add(r.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int s = (int)rfl(tk);
    constexpr int wps = 32 * K;
    StripJob j;
    j.a_codes = r.a_codes;
    j.b_prof = r.b_prof;
    j.v = r.v;
    j.hin_gran = s > 0 ? r.gran + (size_t)(s - 1) * r.gran_stride : nullptr;
    j.hin_arr = s == 0 ? r.hin_arr : nullptr;
    j.hout_gran = s + 1 < S ? r.gran + (size_t)s * r.gran_stride : nullptr;
    j.hout_arr = s + 1 < S ? nullptr : r.hout_arr;
    j.values = FILL ? r.values : nullptr;
    j.sum_out = s + 1 < S ? nullptr : r.sum_out;
    j.n = r.n;
    j.word0 = r.w0 + s * wps;
    const int words = (r.w1 - j.word0) < wps ? (r.w1 - j.word0) : wps;
    j.nlanes = 2 * words;
    j.fill_stride = FILL ? r.fill_stride : 0;
    j.fill_word0 = FILL ? j.word0 - r.w0 : 0;
    j.exact_tail = s + 1 < S ? 1 : ((r.exact_end || r.hout_arr) ? 1 : 0);
    j.flags = 0;
    j.col0 = r.col0;
    j.tail_rows = -1;
    j.k = K;
    j.ckpt = nullptr;
    j.ckpt_stride = 0;
    j.hin_n = 0;
    j.vsum_out = nullptr;
    if (K == 1 && j.nlanes <= 32) run_strip<1, FILL, false, false, false, false, true>(j, r.err);  // half-wave: one chunk less
    else run_strip<K, FILL, false>(j, r.err);
    // completion: results first (system scope: v and the sum are in host memory), then the count, then the flag
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    uint32_t c = 0;
    if (lane == 0) c = __hip_atomic_fetch_add(r.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    c = rfl(c);
    if (c == (uint32_t)(S - 1) && lane == 0) {
        __hip_atomic_store(r.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(r.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(r.done, r.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Up to three vertically stacked rectangles of one 256-column block in ONE launch: the ranges of the engine's incremental
// doubling (blocks.rs:370-469: None / Update / Input, or Output / Input).  Segments are independent except that a
// segment with top == kTopChain takes the bottom row of the segment before it, which is the row that segment also stores
// into the persistent h row -- exactly what a separate launch would have read back from there.
enum : int32_t { kTopOne = 0, kTopStored = 1, kTopChain = 2 };
struct ChainArgs {
    const uint32_t* a_codes;
    const uint32_t* b_prof;
    uint32_t* v;            // host-mapped, biased so that it is indexed by absolute word
    uint8_t* h_arr;         // the persistent h row (device), one byte per absolute column
    uint64_t* gran;         // one row of gran_stride granules per strip of the launch, all zero between launches
    uint64_t gran_stride;
    int32_t* sum_out;       // host-mapped: bottom-row sum of the LAST segment
    uint32_t* err;
    uint32_t* done;
    uint32_t* counter;
    int32_t n, col0, nseg;
    uint32_t seq;
    int32_t w0[3], w1[3], top[3], store[3];
};

template <int K>
__global__ __launch_bounds__(64) void rect_chain_kernel(ChainArgs r) {
    const int total = (int)gridDim.x;
    const int lane = (int)(threadIdx.x & 63);
    uint32_t tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add(r.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int b = (int)rfl(tk);  // ticket order == dependency order (see rect_kernel)
    constexpr int wps = 32 * K;
    int g = 0, s = b;
    for (; g < r.nseg; ++g) {
        const int S = (r.w1[g] - r.w0[g] + wps - 1) / wps;
        if (s < S) break;
        s -= S;
    }
    const int S = (r.w1[g] - r.w0[g] + wps - 1) / wps;
    const bool last = s + 1 == S;
    const bool feeds_next = last && g + 1 < r.nseg && r.top[g + 1] == kTopChain;
    StripJob j;
    j.a_codes = r.a_codes;
    j.b_prof = r.b_prof;
    j.v = r.v;
    j.hin_gran = (s > 0 || r.top[g] == kTopChain) ? r.gran + (size_t)(b - 1) * r.gran_stride : nullptr;
    j.hin_arr = (s == 0 && r.top[g] == kTopStored) ? r.h_arr : nullptr;
    j.hout_gran = (!last || feeds_next) ? r.gran + (size_t)b * r.gran_stride : nullptr;
    j.hout_arr = (last && r.store[g]) ? r.h_arr : nullptr;
    j.values = nullptr;
    j.sum_out = (last && g + 1 == r.nseg) ? r.sum_out : nullptr;
    j.n = r.n;
    j.word0 = r.w0[g] + s * wps;
    const int words = (r.w1[g] - j.word0) < wps ? (r.w1[g] - j.word0) : wps;
    j.nlanes = 2 * words;
    j.fill_stride = 0;
    j.fill_word0 = 0;
    j.exact_tail = (!last || feeds_next || r.store[g]) ? 1 : 0;
    j.flags = 0;
    j.col0 = r.col0;
    j.tail_rows = -1;
    j.k = K;
    j.ckpt = nullptr;
    j.ckpt_stride = 0;
    j.hin_n = 0;
    j.vsum_out = nullptr;
    if (K == 1 && j.nlanes <= 32) run_strip<1, false, false, false, false, false, true>(j, r.err);  // half-wave: one chunk less
    else run_strip<K, false, false>(j, r.err);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    uint32_t c = 0;
    if (lane == 0) c = __hip_atomic_fetch_add(r.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    c = rfl(c);
    if (c == (uint32_t)(total - 1) && lane == 0) {
        __hip_atomic_store(r.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(r.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(r.done, r.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Sequential-pairs variant: one wavefront runs ALL strips of one rectangle top to bottom (jobs[first[p]] ..
// jobs[first[p+1]-1]); the bottom row of strip s goes through the same granule rows (two per rectangle, ping-pong) but
// is produced and consumed by the same wavefront, so nothing ever polls.  With >= one rectangle per SIMD this removes the
// strip-to-strip coupling that costs chained strips 40-70 % at 2-7 wavefronts per SIMD (profiles/r01_runs/chain_probe2.log).
// The second launch bound asks for four wavefronts per SIMD (<= 128 VGPRs; the K = 8 step fits without spilling in its loops):
// a 4096-pair batch is exactly four per SIMD, three would leave a quarter of it for a second, mostly empty round.
// LDSEQ (K >= 4): the launch provides kStripBlockWaves * LdsEq<K>::kWaveBytes of dynamic LDS, the kernel's only LDS, so the
// slices start at offset 0 and are aligned to their size.
// (K = 16, an experiment of round 3: 256 VGPRs, two wavefronts per SIMD, 16 KB of LDS per wavefront)
template <int K, bool CKPT = false, bool LDSEQ = false>
__global__ __launch_bounds__(64 * kStripBlockWaves, K >= 16 ? 2 : 4) void pair_kernel(const StripJob* __restrict__ jobs,
                                                                      const int32_t* __restrict__ first, int npairs,
                                                                      uint32_t* err) {
    const int p = (int)rfl((uint32_t)(blockIdx.x * kStripBlockWaves + (threadIdx.x >> 6)));
    if (p >= npairs) return;
    const uint32_t lds_wave = rfl((uint32_t)(threadIdx.x >> 6)) * LdsEq<K>::kWaveBytes;
    const int j0 = first[p], j1 = first[p + 1];
    for (int j = j0; j < j1; ++j) {
        const StripJob job = jobs[j];
        // the ragged bottom of a pair runs as short 32-row-per-lane strips instead of one mostly empty tall one
        if (K > 1 && job.k == 1) run_strip<1, false, false, CKPT, true>(job, err);
        else run_strip<K, false, false, CKPT, true, LDSEQ>(job, err, lds_wave);
        // the next strip reads what this one stored (granules): drain and order the stores first (same wavefront, same CU)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

}  // namespace pa
