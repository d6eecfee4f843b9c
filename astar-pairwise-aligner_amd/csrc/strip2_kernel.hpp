// strip2_kernel.hpp -- TWO half-wave strips in one wavefront (gfx950), and the LDS side of their rendezvous (rdv_logic.hpp).
//
// What it is for: the batched A*PA2 kernels (apa2_kernel.hpp, apa2_full_kernel.hpp).  A 256-column block of a short or similar pair is
// about ten 64-row words tall: its strip (strip_kernel.hpp run_strip<1, .., HALF>) keeps 20 of the 64 lanes busy and is 60-85 % of
// all the instructions those kernels issue.  run_strip_dual runs the block of pair A in lanes 0..31 and the block of pair B in lanes
// 32..63 with ONE instruction stream: 25 VALU instructions per step instead of 24 + 24.
//   * Same Myers step (myers.rs:27-55 on 32-row subwords), same anti-diagonal skew: at step t logical lane l of either half handles
//     column t - l of ITS rectangle.  The packed pipeline register still moves lane -> lane + 1 through one DPP wave_shr:1; pair A's
//     column input enters lane 0 through the DPP's `old` operand, pair B's enters lane 32 by being written over lane 31's copy of the
//     register (v_writelane) just before the shift -- lane 31's own outgoing deltas were accumulated one instruction earlier.
//   * Everything that differs between the two rectangles -- pointers, first word, number of columns, first column, the tap lane of
//     incremental doubling -- is a per-lane (vector) value selected by the lane's half; control flow stays wavefront-uniform, the
//     chunk count is the larger of the two.
//   * Bottom rows: pair A's leaves lane 31, pair B's lane 63; lanes beyond a rectangle's rows run as zero pad rows and the sums are
//     corrected by their right edge (the identity of the reference's padded tail, simd.rs:184-225) -- per half.
// A strip qualifies when it is a whole block on its own: K = 1, at most 32 lanes, no granules in or out, no exact bottom row wanted.
#pragma once
#include "rdv_logic.hpp"
#include "rdv_params.hpp"
#include "strip_kernel.hpp"

namespace pa {

// What travels through the mailbox: the part of a StripJob a half-wave block needs (22 dwords).
struct DualJob {
    const uint32_t* a_codes;  // as StripJob
    const uint32_t* b_prof;
    uint32_t* v;              // the block's column, updated in place (or written from `values`)
    const uint8_t* hin_arr;   // top-row deltas by absolute column, or nullptr => all +1
    uint8_t* hout_arr;        // TAP: receives the deltas leaving logical lane `tap` (when tap >= 0)
    const uint32_t* values;   // TAP: source of the left edge (the previous block's column; words [fill_word0, fill_stride)), or nullptr => v
    int32_t* sum_out;         // bottom-row sum
    int32_t n, word0, nlanes, fill_stride, fill_word0, col0, tap, prio;  // prio: the issue priority of the wavefront that owns the strip (0..3)
};
static_assert(sizeof(DualJob) == 88, "DualJob layout");
constexpr int kDualWords = 22;

template <bool TAP>
__device__ __forceinline__ DualJob dual_from(const StripJob& j, int tap) {
    DualJob d;
    d.a_codes = j.a_codes;
    d.b_prof = j.b_prof;
    d.v = j.v;
    d.hin_arr = j.hin_arr;
    d.hout_arr = TAP ? j.hout_arr : nullptr;
    d.values = TAP ? j.values : nullptr;
    d.sum_out = j.sum_out;
    d.n = j.n;
    d.word0 = j.word0;
    d.nlanes = j.nlanes;
    d.fill_stride = j.fill_stride;
    d.fill_word0 = j.fill_word0;
    d.col0 = j.col0;
    d.tap = TAP ? tap : -1;
    d.prio = 0;
    return d;
}
// Does a strip of the band-search kernels qualify (see the header)?
__device__ __forceinline__ bool dual_ok(const StripJob& j) {
    return j.nlanes <= 32 && j.hin_gran == nullptr && j.hout_gran == nullptr && j.exact_tail == 0 && j.flags == 0 && j.sum_out != nullptr && j.n > 0;
}

template <bool PRED>
__device__ __forceinline__ void myers_step_dual(uint32_t s0, uint32_t s1, uint32_t& X, uint32_t& vp, uint32_t& vm, uint32_t nb0, uint32_t nb1, uint32_t& acc,
                                                bool active, uint32_t k40, uint32_t k80) {
    acc = __builtin_amdgcn_alignbit(acc, X, 30);  // (acc << 2) | (X >> 30): the lane's own outgoing deltas, one step late
    uint32_t Xw = X;
    asm("v_writelane_b32 %0, %1, 31" : "+v"(Xw) : "s"(s1));  // pair B's column input, delivered to lane 32 by the shift
    const uint32_t Xin = dpp_wave_shr1(s0, Xw);               // pair A's enters lane 0 (no source lane: keeps `old`)
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
    const uint32_t xm = __builtin_amdgcn_bitop3_b32(k40, hm >> 1, Xin, 0xCA);
    const uint32_t Xo = __builtin_amdgcn_bitop3_b32(k80, hp, xm, 0xCA);
    const uint32_t hp2 = __builtin_amdgcn_alignbit(hp, Xin, 31);
    const uint32_t hm2 = (hm << 1) | hm0;
    const uint32_t nvp = __builtin_amdgcn_bitop3_b32(hm2, vx, hp2, 0xF1);
    const uint32_t nvm = hp2 & vx;
    if (PRED) {
        vp = active ? nvp : vp;
        vm = active ? nvm : vm;
    } else {
        vp = nvp;
        vm = nvm;
    }
    X = Xo;
}

template <bool PRED>
__device__ __forceinline__ void run_chunk_dual(int q, uint32_t XS, uint32_t& X, uint32_t& vp, uint32_t& vm, uint32_t nb0, uint32_t nb1, uint32_t& acc_lo,
                                               uint32_t& acc_hi, int lane, int n_l, uint32_t k40, uint32_t k80) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)XS, j);
        const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)XS, 32 + j);
        const int col = q * 32 + j - lane;
        const bool active = PRED ? ((unsigned)col < (unsigned)n_l) : true;
        myers_step_dual<PRED>(s0, s1, X, vp, vm, nb0, nb1, j < 16 ? acc_lo : acc_hi, active, k40, k80);
    }
}

// J0 runs in lanes 0..31, J1 in lanes 32..63.  TAP: see run_strip (apa2_full_kernel.hpp: the stored row of incremental doubling and the
// left edge read straight from the previous block's column).
template <bool TAP>
__device__ __forceinline__ void run_strip_dual(const DualJob& J0, const DualJob& J1) {
    const int plane = (int)(threadIdx.x & 63);
    const bool hi = plane >= 32;
    const int lane = plane & 31;  // logical lane of either half: rows 32 * lane .. of its strip
    const int n0 = J0.n, n1 = J1.n;
    const int n_l = hi ? n1 : n0;
    const int C0 = (n0 + 31) >> 5, C1 = (n1 + 31) >> 5;
    const int C = C0 > C1 ? C0 : C1;
    const int nlanes_l = hi ? J1.nlanes : J0.nlanes;
    const int col0_l = hi ? J1.col0 : J0.col0;
    const bool real = lane < nlanes_l;
    const gcu32 g_prof = (gcu32)(hi ? J1.b_prof : J0.b_prof);
    const gu32 g_v = (gu32)(hi ? J1.v : J0.v);
    const int word = (hi ? J1.word0 : J0.word0) + (lane >> 1), half = lane & 1;

    uint32_t vp = 0, vm = 0, nb0 = 0, nb1 = 0;
    if (real) {
        nb0 = g_prof[word * 4 + half];
        nb1 = g_prof[word * 4 + 2 + half];
        const gcu32 g_vs = (gcu32)(hi ? J1.values : J0.values);
        if (TAP && g_vs != nullptr) {
            // init_v_with_overlap (blocks.rs:753-767) folded into the strip, as in run_strip<.., TAP>
            const int fw0 = hi ? J1.fill_word0 : J0.fill_word0, fw1 = hi ? J1.fill_stride : J0.fill_stride;
            vp = 0xFFFFFFFFu;
            vm = 0u;
            if (word >= fw0 && word < fw1) {
                vp = g_vs[word * 4 + half];
                vm = g_vs[word * 4 + 2 + half];
            }
        } else {
            vp = g_v[word * 4 + half];
            vm = g_v[word * 4 + 2 + half];
        }
    }

    uint32_t X = 0, acc_lo = 0, acc_hi = 0;
    int32_t sum0 = 0, sum1 = 0;
    uint32_t k40 = 0x40000000u, k80 = 0x80000000u;
    asm volatile("" : "+v"(k40), "+v"(k80));
    const int cj = plane & 15;
    const bool upper = (plane & 16) != 0;
    const uint32_t sh = 2u * (uint32_t)cj;

    typedef const __attribute__((address_space(4))) uint32_t* ccu32;
    const ccu32 c_codes0 = (ccu32)J0.a_codes, c_codes1 = (ccu32)J1.a_codes;
    const gcu8 g_hin = (gcu8)(hi ? J1.hin_arr : J0.hin_arr);
    const bool has_hin = g_hin != nullptr;
    const gcu8 hin_src = has_hin ? g_hin : (gcu8)(hi ? J1.a_codes : J0.a_codes);  // (any readable address)
    const int Cm1_0 = C0 - 1, Cm1_1 = C1 - 1;  // (n > 0: see dual_ok)
    const int lw0 = (J0.col0 + n0 - 1) >> 4, lw1 = (J1.col0 + n1 - 1) >> 4;
    struct RawCodes {
        uint32_t w0, w1, w2;
    };
    auto load_codes = [&](ccu32 c, int col0, int Cm1, int last_word, int q) -> RawCodes {
        const int c0 = col0 + 32 * (q < Cm1 ? q : Cm1);
        const int i0 = c0 >> 4;
        const int i1 = i0 + 1 < last_word ? i0 + 1 : last_word;
        const int i2 = i0 + 2 < last_word ? i0 + 2 : last_word;
        return RawCodes{c[i0], c[i1], c[i2]};
    };
    auto decode_codes = [&](const RawCodes& r, int col0, int Cm1, int q) -> uint64_t {
        const unsigned s2 = 2u * (unsigned)((col0 + 32 * (q < Cm1 ? q : Cm1)) & 15);
        const uint64_t lo64 = (uint64_t)r.w0 | ((uint64_t)r.w1 << 32);
        return s2 == 0 ? lo64 : ((lo64 >> s2) | ((uint64_t)r.w2 << (64 - s2)));
    };
    auto load_hin_byte = [&](int q) -> uint32_t {  // top delta byte of column 32q + lane of the lane's own rectangle
        int c = 32 * q + lane;
        c = c < n_l ? c : n_l - 1;
        return (uint32_t)hin_src[has_hin ? col0_l + c : 0];
    };
    // granule g (columns 32g .. 32g + 31 of a bottom row) from the lagged accumulators of the half's last lane
    auto publish = [&](int g) {
        if (g < C0) {
            const int cols = n0 - 32 * g;
            uint32_t vlo = (uint32_t)__builtin_amdgcn_readlane((int)acc_lo, 31), vhi = (uint32_t)__builtin_amdgcn_readlane((int)acc_hi, 31);
            if (cols < 32) {
                const int cl = cols >= 16 ? 16 : cols, ch = cols > 16 ? cols - 16 : 0;
                vlo &= cl >= 16 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (2 * cl));
                vhi &= ch == 0 ? 0u : ~(0xFFFFFFFFu >> (2 * ch));
            }
            sum0 += __builtin_popcount(vlo & 0xAAAAAAAAu) + __builtin_popcount(vhi & 0xAAAAAAAAu) - __builtin_popcount(vlo & 0x55555555u) -
                    __builtin_popcount(vhi & 0x55555555u);
        }
        if (g < C1) {
            const int cols = n1 - 32 * g;
            uint32_t vlo = (uint32_t)__builtin_amdgcn_readlane((int)acc_lo, 63), vhi = (uint32_t)__builtin_amdgcn_readlane((int)acc_hi, 63);
            if (cols < 32) {
                const int cl = cols >= 16 ? 16 : cols, ch = cols > 16 ? cols - 16 : 0;
                vlo &= cl >= 16 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (2 * cl));
                vhi &= ch == 0 ? 0u : ~(0xFFFFFFFFu >> (2 * ch));
            }
            sum1 += __builtin_popcount(vlo & 0xAAAAAAAAu) + __builtin_popcount(vhi & 0xAAAAAAAAu) - __builtin_popcount(vlo & 0x55555555u) -
                    __builtin_popcount(vhi & 0x55555555u);
        }
    };

    RawCodes codes0 = load_codes(c_codes0, J0.col0, Cm1_0, lw0, 0), codes1 = load_codes(c_codes1, J1.col0, Cm1_1, lw1, 0);
    uint32_t hinb_next = load_hin_byte(0);
    const int tap_l = TAP ? (hi ? J1.tap : J0.tap) : -1;
    const gu8 g_hout = (gu8)(hi ? J1.hout_arr : J0.hout_arr);
    const int tp0 = TAP && J0.tap >= 0 ? J0.tap : 0, tp1 = 32 + (TAP && J1.tap >= 0 ? J1.tap : 0);
    const bool any_tap = TAP && (J0.tap >= 0 || J1.tap >= 0);
    const int nmin = n0 < n1 ? n0 : n1;

    // Steps t = 0 .. 32 (C + 1) - 1; the accumulators lag one step, so after chunk q the last lane of a half holds granule q - 1.
    for (int q = 0; q < C + 1; ++q) {
        asm volatile("" : "+v"(hinb_next));
        const uint64_t c64_0 = decode_codes(codes0, J0.col0, Cm1_0, q), c64_1 = decode_codes(codes1, J1.col0, Cm1_1, q);
        const uint32_t cw = hi ? (upper ? (uint32_t)(c64_1 >> 32) : (uint32_t)c64_1) : (upper ? (uint32_t)(c64_0 >> 32) : (uint32_t)c64_0);
        const uint32_t code = (32 * q + lane < n_l) ? ((cw >> sh) & 3u) : 0u;
        const uint32_t hin2 = has_hin ? (((hinb_next & 1u) << 31) | ((hinb_next & 2u) << 29)) : 0x80000000u;
        const uint32_t XS = code | hin2;
        if (q >= 2) publish(q - 2);
        codes0 = load_codes(c_codes0, J0.col0, Cm1_0, lw0, q + 1);
        codes1 = load_codes(c_codes1, J1.col0, Cm1_1, lw1, q + 1);
        hinb_next = load_hin_byte(q + 1);
        const bool interior = q >= 1 && q * 32 + 31 < nmin;  // every lane of both halves is inside its rectangle
        if (interior) run_chunk_dual<false>(q, XS, X, vp, vm, nb0, nb1, acc_lo, acc_hi, lane, n_l, k40, k80);
        else run_chunk_dual<true>(q, XS, X, vp, vm, nb0, nb1, acc_lo, acc_hi, lane, n_l, k40, k80);
        if (any_tap) {
            // logical lane tap's accumulators hold columns 32q - 1 - tap .. 32q + 30 - tap after chunk q (run_strip's TAP)
            const uint32_t tlo = hi ? (uint32_t)__builtin_amdgcn_readlane((int)acc_lo, tp1) : (uint32_t)__builtin_amdgcn_readlane((int)acc_lo, tp0);
            const uint32_t thi = hi ? (uint32_t)__builtin_amdgcn_readlane((int)acc_hi, tp1) : (uint32_t)__builtin_amdgcn_readlane((int)acc_hi, tp0);
            const int c = 32 * q - 1 - tap_l + lane;
            if (tap_l >= 0 && c >= 0 && c < n_l) {
                const uint32_t tb = ((upper ? thi : tlo) >> (30 - 2 * cj)) & 3u;  // bit1 = p, bit0 = m
                g_hout[col0_l + c] = (uint8_t)((tb >> 1) | ((tb & 1u) << 1));
            }
        }
    }
    publish(C - 1);
    if (real) {
        g_v[word * 4 + half] = vp;
        g_v[word * 4 + 2 + half] = vm;
    }
    // zero pad rows: subtract their right-edge value (simd.rs:202-224), per half
    const int32_t c = real ? 0 : __builtin_popcount(vp) - __builtin_popcount(vm);
    const int32_t incl = wave_scan_add(c);
    const int32_t pad0 = __builtin_amdgcn_readlane(incl, 31), pad1 = __builtin_amdgcn_readlane(incl, 63) - pad0;
    if (plane == 0) *(gi32)J0.sum_out = sum0 - pad0;
    if (plane == 32) *(gi32)J1.sum_out = sum1 - pad1;
}

// ---- the LDS side of the rendezvous (policy of rdv_logic.hpp) -----------------------------------------------------------------------
struct RdvShared {
    uint32_t st;        // one byte per wavefront (rdv_logic.hpp)
    uint32_t live;      // wavefronts still inside their pair loop
    uint32_t pad[62];
    uint32_t dummy[64]; // the lanes that take no part in an atomic aim here (no lane-dependent branch around the atomic)
    uint32_t mail[rdv::kMaxWaves][64];
};
typedef __attribute__((address_space(3))) uint32_t* lds_u32;

struct RdvLds {
    lds_u32 base;  // &RdvShared in LDS
    int lane;
    __device__ __forceinline__ lds_u32 st_addr() const { return base; }
    __device__ __forceinline__ lds_u32 live_addr() const { return base + 1; }
    __device__ __forceinline__ lds_u32 mine(lds_u32 target) const { return lane == 0 ? target : base + 64 + lane; }
    __device__ __forceinline__ uint32_t load() const { return rfl(__hip_atomic_load(st_addr(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
    __device__ __forceinline__ uint32_t live() const { return rfl(__hip_atomic_load(live_addr(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
    __device__ __forceinline__ bool cas(uint32_t expect, uint32_t desired) const {
        uint32_t e = lane == 0 ? expect : 0u;
        const uint32_t d = lane == 0 ? desired : 0u;
        __hip_atomic_compare_exchange_strong(mine(st_addr()), &e, d, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return rfl(e) == expect;  // (lane 0's old value)
    }
    __device__ __forceinline__ void add(uint32_t delta) const {
        (void)__hip_atomic_fetch_add(mine(st_addr()), lane == 0 ? delta : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void leave() const {  // this wavefront has left its pair loop
        (void)__hip_atomic_fetch_add(mine(live_addr()), lane == 0 ? 0xFFFFFFFFu : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ uint64_t now() const { return wall_clock64(); }
    __device__ __forceinline__ void nap() const { __builtin_amdgcn_s_sleep(16); }  // ~0.4 us, issues nothing

    // lane i holds dword i of the job; one ds_write / ds_read per wavefront
    __device__ __forceinline__ void write_mail(int w, const DualJob& j) const {
        uint32_t x = 0;
#define PA_PUT(i, v) asm("v_writelane_b32 %0, %1, " #i : "+v"(x) : "s"(rfl((uint32_t)(v))))  // (rfl: a value the compiler keeps in a vector register is uniform all the same)
#define PA_PUT_PTR(i, i1, p)                                   \
    do {                                                       \
        const uint64_t u_ = (uint64_t)(uintptr_t)(p);          \
        PA_PUT(i, (uint32_t)u_);                               \
        PA_PUT(i1, (uint32_t)(u_ >> 32));                      \
    } while (0)
        PA_PUT_PTR(0, 1, j.a_codes);
        PA_PUT_PTR(2, 3, j.b_prof);
        PA_PUT_PTR(4, 5, j.v);
        PA_PUT_PTR(6, 7, j.hin_arr);
        PA_PUT_PTR(8, 9, j.hout_arr);
        PA_PUT_PTR(10, 11, j.values);
        PA_PUT_PTR(12, 13, j.sum_out);
        PA_PUT(14, j.n);
        PA_PUT(15, j.word0);
        PA_PUT(16, j.nlanes);
        PA_PUT(17, j.fill_stride);
        PA_PUT(18, j.fill_word0);
        PA_PUT(19, j.col0);
        PA_PUT(20, j.tap);
        PA_PUT(21, j.prio);
#undef PA_PUT_PTR
#undef PA_PUT
        *(base + 128 + 64 * w + lane) = x;
    }
    __device__ __forceinline__ DualJob read_mail(int v) const {
        const uint32_t x = *(base + 128 + 64 * v + lane);
        auto get = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)x, i); };
        auto get_ptr = [&](int i) { return (uintptr_t)((uint64_t)get(i) | ((uint64_t)get(i + 1) << 32)); };
        DualJob j;
        j.a_codes = (const uint32_t*)get_ptr(0);
        j.b_prof = (const uint32_t*)get_ptr(2);
        j.v = (uint32_t*)get_ptr(4);
        j.hin_arr = (const uint8_t*)get_ptr(6);
        j.hout_arr = (uint8_t*)get_ptr(8);
        j.values = (const uint32_t*)get_ptr(10);
        j.sum_out = (int32_t*)get_ptr(12);
        j.n = (int32_t)get(14);
        j.word0 = (int32_t)get(15);
        j.nlanes = (int32_t)get(16);
        j.fill_stride = (int32_t)get(17);
        j.fill_word0 = (int32_t)get(18);
        j.col0 = (int32_t)get(19);
        j.tap = (int32_t)get(20);
        j.prio = (int32_t)get(21);
        return j;
    }
};

// Once per workgroup, before the pair loops (every wavefront calls it; ends with the workgroup's only barrier).
__device__ __forceinline__ void rdv_init(RdvShared* sh, int nwaves) {
    if (threadIdx.x < 64) {
        lds_u32 b = (lds_u32)sh;
        b[threadIdx.x] = threadIdx.x == 1 ? (uint32_t)nwaves : 0u;
    }
    __syncthreads();
}

constexpr uint64_t kRdvHardTicks = 2ull * 100000000ull;  // 2 s: a taken strip that never finishes (a block's strip takes tens of microseconds)

// A qualifying strip of wavefront `w` arrives.  Returns true when the strip has been computed (by this wavefront together with a
// partner's, or by a partner): its results are in memory.  false: run it alone.
__device__ __forceinline__ void set_prio(int32_t pr) {  // (s_setprio takes an immediate)
    if (pr >= 3) __builtin_amdgcn_s_setprio(3);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}
// `my_prio`: this wavefront's issue priority (RdvParams::prio: by its pair's rank in the start order).  A fused strip runs at the HIGHER of
// the two owners' priorities: the strip of a pair on the launch's critical path must not run at the pace of a cheap pair that took it.
template <bool TAP>
__device__ __forceinline__ bool rdv_strip(const RdvLds& p, int w, const RdvParams& rp, const StripJob& sj, int tap, uint32_t* err, rdv::Counters* cnt,
                                          uint32_t* strip_units, int32_t my_prio = 0) {
    DualJob mine = dual_from<TAP>(sj, tap);
    mine.prio = my_prio;
    p.write_mail(w, mine);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the mail and everything this block's strip reads (left edge, stored row)
    int partner = -1;
    RdvLds pol = p;
    const int32_t r = rdv::arrive(pol, w, rdv::kMaxWaves, (uint64_t)rp.patience, kRdvHardTicks, &partner, cnt);
    if (r == rdv::kAlone) return false;
    if (r == rdv::kTook) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const DualJob theirs = p.read_mail(partner);
        const bool lift = theirs.prio > my_prio;
        if (lift) set_prio(theirs.prio);
        run_strip_dual<TAP>(theirs, mine);
        if (lift) set_prio(my_prio);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // both strips' results, before the partner wakes up
        rdv::finish(pol, partner);
        const int nmax = mine.n > theirs.n ? mine.n : theirs.n;
        *strip_units += (uint32_t)((((nmax + 31) >> 5) + 1) * 25);
        return true;
    }
    if (r == rdv::kStuck) {
        if (p.lane == 0) __hip_atomic_store((gu32)err, (uint32_t)PA_ERR_SPIN_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // served: the partner's stores
    return true;
}

}  // namespace pa
