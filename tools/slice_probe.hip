// Round-6 experiment: the BIT-SLICED form of the Myers/Hyyro column step ("a carry-free anti-diagonal bit-vector formulation").
//
// Transposed layout: bit p of every 32-bit register belongs to pair p of a GROUP of 32 pairs, one register per DP ROW.  Then the step of
// pa-bitpacking/src/myers.rs:27-55 needs neither the add nor the shifts: with c_0 = hm_in and c_(i+1) = hm_i the carry of
// ((eq & vp) + vp) IS the horizontal minus-delta of the row above, and `<< 1` is "take the register of the row above":
//     x_i   = eq_i | hm_(i-1)                    (= hx_i)
//     hm_i  = vp_i & x_i
//     hp_i  = vm_i | ~(x_i | vp_i)
//     vx_i  = eq_i | vm_i
//     vp'_i = hm_(i-1) | ~(vx_i | hp_(i-1))
//     vm'_i = hp_(i-1) & vx_i
// 6 boolean ops + 2 for eq = 8 two/three-input logic ops per (row x 32 pairs x 64 lanes) = 2048 cells, ALL of the fast VALU class
// (no v_add_co, no v_alignbit, no DPP per row) -- against 11.3 mixed instructions per 2048 cells in pair_kernel<8>.
// A lane owns R consecutive rows in registers; lanes are skewed one column per lane (anti-diagonal), the bottom row's (hp, hm) moves to
// the next lane through one DPP wave_shr:1 each; strips of 64 R rows run one after the other, handing the bottom row down through memory.
//
//   slice_probe <groups> <n> <m> [check]     R is a compile-time constant (-DSLICE_R=48)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#ifndef SLICE_SB
#define SLICE_SB 4
#endif
#ifndef SLICE_ASM
#define SLICE_ASM 1
#endif
#ifndef SLICE_WAVES
#define SLICE_WAVES 2
#endif
#ifndef SLICE_R
#define SLICE_R 48
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t old_, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old_, (int)src, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ uint2 ld_h(const uint2* p) {  // L1-bypassing 8-byte load (the row was written by this wavefront one strip ago)
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

constexpr int kPad = 64;  // entries in front of column 0 and behind column n - 1 of the per-column arrays: no clamping in the loop

template <int R>
__global__ __launch_bounds__(64, SLICE_WAVES) void slice_kernel(const uint2* __restrict__ A, const uint2* __restrict__ B, uint2* H0, uint2* H1,
                                                      uint2* __restrict__ Vout, int n, int nstrips, size_t a_stride, size_t b_stride,
                                                      size_t h_stride) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const uint2* Ag = A + (size_t)g * a_stride + kPad;  // column c at Ag[c], c in [-kPad, n + kPad)
    const uint2* Bg = B + (size_t)g * b_stride;
    uint2* Vg = Vout + (size_t)g * b_stride;
    for (int s = 0; s < nstrips; ++s) {
        const uint2* Hin = ((s & 1) ? H1 : H0) + (size_t)g * h_stride;
        uint2* Hout = ((s & 1) ? H0 : H1) + (size_t)g * h_stride;
        const size_t row0 = ((size_t)s * 64 + lane) * R;
        uint32_t nb0[R], nb1[R], vp[R], vm[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const uint2 b = Bg[row0 + i];
            nb0[i] = b.x;
            nb1[i] = b.y;
            vp[i] = ~0u;
            vm[i] = 0u;
        }
        uint32_t o_hp = 0, o_hm = 0;
        uint2 acol = Ag[-lane];  // column t - lane for t = 0
        uint2 hin = make_uint2(~0u, 0u);
        if (lane == 0) hin = ld_h(Hin);
        const int steps = n + 63;
        for (int t = 0; t < steps; ++t) {
            const int c = t - lane;
            // next step's inputs, issued a whole step ahead of their use
            const uint2 acol_next = Ag[c + 1];
            uint2 hin_next = make_uint2(0u, 0u);
            if (lane == 0) hin_next = ld_h(Hin + t + 1);
            uint32_t hpp = dpp_wave_shr1(hin.x, o_hp);
            uint32_t hmp = dpp_wave_shr1(hin.y, o_hm);
            if ((unsigned)c < (unsigned)n) {
                const uint32_t a0 = acol.x, a1 = acol.y;
#if SLICE_ASM
                static_assert(R % 2 == 0, "rows are stepped in pairs");
#pragma unroll
                for (int i = 0; i < R; i += 2) {
                    // two rows (A = i, B = i + 1) in an order in which no instruction reads the result of the one before it, and the
                    // chain value hm_B is ready 8 instructions before the block ends
                    uint32_t eA, eB, x, vx, hmA, hpA, hmB, hpB;
                    asm volatile(
                        "v_xor_b32 %[eA], %[a1], %[nb1A]\n\t"
                        "v_xor_b32 %[eB], %[a1], %[nb1B]\n\t"
                        "v_bitop3_b32 %[eA], %[a0], %[nb0A], %[eA] bitop3:0x28\n\t"
                        "v_bitop3_b32 %[eB], %[a0], %[nb0B], %[eB] bitop3:0x28\n\t"
                        "v_bitop3_b32 %[hmA], %[vpA], %[eA], %[hmp] bitop3:0xe0\n\t"
                        "v_or_b32 %[x], %[eA], %[hmp]\n\t"
                        "v_or_b32 %[vx], %[eA], %[vmA]\n\t"
                        "v_bitop3_b32 %[hmB], %[vpB], %[eB], %[hmA] bitop3:0xe0\n\t"
                        "v_bitop3_b32 %[hpA], %[vmA], %[x], %[vpA] bitop3:0xf1\n\t"
                        "v_bitop3_b32 %[vpA], %[hmp], %[vx], %[hpp] bitop3:0xf1\n\t"
                        "v_and_b32 %[vmA], %[hpp], %[vx]\n\t"
                        "v_or_b32 %[x], %[eB], %[hmA]\n\t"
                        "v_or_b32 %[vx], %[eB], %[vmB]\n\t"
                        "v_bitop3_b32 %[hpB], %[vmB], %[x], %[vpB] bitop3:0xf1\n\t"
                        "v_bitop3_b32 %[vpB], %[hmA], %[vx], %[hpA] bitop3:0xf1\n\t"
                        "v_and_b32 %[vmB], %[hpA], %[vx]"
                        : [eA] "=&v"(eA), [eB] "=&v"(eB), [x] "=&v"(x), [vx] "=&v"(vx), [hmA] "=&v"(hmA), [hpA] "=&v"(hpA), [hmB] "=&v"(hmB),
                          [hpB] "=&v"(hpB), [vpA] "+v"(vp[i]), [vmA] "+v"(vm[i]), [vpB] "+v"(vp[i + 1]), [vmB] "+v"(vm[i + 1])
                        : [a0] "v"(a0), [a1] "v"(a1), [nb0A] "v"(nb0[i]), [nb1A] "v"(nb1[i]), [nb0B] "v"(nb0[i + 1]), [nb1B] "v"(nb1[i + 1]),
                          [hpp] "v"(hpp), [hmp] "v"(hmp));
                    hpp = hpB;
                    hmp = hmB;
                }
#else
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const uint32_t eq = __builtin_amdgcn_bitop3_b32(a0, nb0[i], a1 ^ nb1[i], 0x28);  // (a0 ^ nb0) & (a1 ^ nb1)
                    const uint32_t x = eq | hmp;
                    const uint32_t hm = __builtin_amdgcn_bitop3_b32(vp[i], eq, hmp, 0xE0);  // vp & (eq | hmp)
                    const uint32_t hp = __builtin_amdgcn_bitop3_b32(vm[i], x, vp[i], 0xF1);  // vm | ~(x | vp)
                    const uint32_t vx = eq | vm[i];
                    vp[i] = __builtin_amdgcn_bitop3_b32(hmp, vx, hpp, 0xF1);  // hmp | ~(vx | hpp)
                    vm[i] = hpp & vx;
                    hpp = hp;
                    hmp = hm;
                    // keep the rows in order: hoisting the whole hm chain (what the scheduler does by itself) costs 2 R registers
                    if (i % SLICE_SB == SLICE_SB - 1) asm volatile("" : "+v"(hpp), "+v"(hmp), "+v"(vp[i]), "+v"(vm[i]));
                }
#endif
                o_hp = hpp;
                o_hm = hmp;
                if (lane == 63) Hout[c] = make_uint2(hpp, hmp);
            }
            acol = acol_next;
            hin = hin_next;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) Vg[row0 + i] = make_uint2(vp[i], vm[i]);
        // the next strip of this wavefront reads what lane 63 stored: the stores have to be visible to its (L1-bypassing) loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_s_waitcnt(0);
    }
}

static int edit_distance(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b) {
    std::vector<int> prev(b.size() + 1), cur(b.size() + 1);
    for (size_t j = 0; j <= b.size(); ++j) prev[j] = (int)j;
    for (size_t i = 1; i <= a.size(); ++i) {
        cur[0] = (int)i;
        for (size_t j = 1; j <= b.size(); ++j) cur[j] = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), prev[j - 1] + (a[i - 1] != b[j - 1]));
        std::swap(prev, cur);
    }
    return prev[b.size()];
}

int main(int argc, char** argv) {
    constexpr int R = SLICE_R;
    const int G = argc > 1 ? atoi(argv[1]) : 2048, n = argc > 2 ? atoi(argv[2]) : 100000, m = argc > 3 ? atoi(argv[3]) : n;
    const bool check = argc > 4;
    const int strip_rows = 64 * R, nstrips = (m + strip_rows - 1) / strip_rows;
    const size_t a_stride = (size_t)n + 2 * kPad, b_stride = (size_t)nstrips * strip_rows, h_stride = (size_t)n + 2 * kPad;
    printf("R=%d groups=%d (pairs %d) n=%d m=%d strips=%d  mem: A %.1f MB B/V %.1f MB H %.1f MB\n", R, G, 32 * G, n, m, nstrips, G * a_stride * 8 / 1e6,
           G * b_stride * 8 / 1e6, 2.0 * G * h_stride * 8 / 1e6);
    std::mt19937_64 rng(12345);
    // sequences (2-bit codes) only materialised for the checked groups; the others get random planes straight away
    std::vector<uint2> hA((size_t)G * a_stride), hB((size_t)G * b_stride);
    std::vector<std::vector<uint8_t>> sa, sb;
    const int checked_groups = check ? std::min(G, 2) : 0;
    std::vector<int> mlen;
    for (int g = 0; g < G; ++g) {
        if (g < checked_groups) {
            for (int p = 0; p < 32; ++p) {
                std::vector<uint8_t> a(n), b;
                for (auto& x : a) x = rng() & 3;
                // b = a with ~8 % edits, length kept <= m (ragged: rows beyond |b| are padding)
                for (int i = 0; i < n; ++i) {
                    const unsigned r = rng() % 100;
                    if (r < 3) continue;                        // deletion
                    if (r < 6) b.push_back(rng() & 3);          // insertion
                    b.push_back(r < 9 ? (a[i] + 1 + rng() % 3) & 3 : a[i]);
                }
                if ((int)b.size() > m) b.resize(m);
                sa.push_back(a);
                sb.push_back(b);
                mlen.push_back((int)b.size());
            }
            for (int c = 0; c < n; ++c) {
                uint32_t a0 = 0, a1 = 0;
                for (int p = 0; p < 32; ++p) {
                    a0 |= (uint32_t)(sa[g * 32 + p][c] & 1) << p;
                    a1 |= (uint32_t)(sa[g * 32 + p][c] >> 1) << p;
                }
                hA[g * a_stride + kPad + c] = make_uint2(a0, a1);
            }
            for (size_t r = 0; r < b_stride; ++r) {
                uint32_t b0 = 0, b1 = 0;
                for (int p = 0; p < 32; ++p) {
                    const auto& b = sb[g * 32 + p];
                    const unsigned code = r < b.size() ? b[r] : 0;
                    b0 |= (uint32_t)(code & 1) << p;
                    b1 |= (uint32_t)(code >> 1) << p;
                }
                hB[g * b_stride + r] = make_uint2(~b0, ~b1);
            }
        } else {
            for (size_t c = 0; c < a_stride; ++c) hA[g * a_stride + c] = make_uint2((uint32_t)rng(), (uint32_t)rng());
            for (size_t r = 0; r < b_stride; ++r) hB[g * b_stride + r] = make_uint2((uint32_t)rng(), (uint32_t)rng());
        }
    }
    uint2 *dA, *dB, *dH0, *dH1, *dV;
    CK(hipMalloc(&dA, hA.size() * 8)); CK(hipMalloc(&dB, hB.size() * 8)); CK(hipMalloc(&dV, hB.size() * 8));
    CK(hipMalloc(&dH0, (size_t)G * h_stride * 8)); CK(hipMalloc(&dH1, (size_t)G * h_stride * 8));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
    std::vector<uint2> hH((size_t)G * h_stride, make_uint2(~0u, 0u));  // the top row of the matrix: +1 everywhere
    CK(hipMemcpy(dH0, hH.data(), hH.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dH1, 0, (size_t)G * h_stride * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = check ? 1 : 3;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((slice_kernel<R>), dim3(G), dim3(64), 0, 0, dA, dB, dH0, dH1, dV, n, nstrips, a_stride, b_stride, h_stride);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double cells = (double)G * 32 * (double)n * (double)m, comp = (double)G * 32 * (double)n * (double)b_stride;
        printf("rep %d: %.3f ms  %.1f TCUPS (n*m)  %.1f TCUPS (computed rows)  ns per 8-op row-step per SIMD %.3f\n", rep, ms, cells / ms / 1e9, comp / ms / 1e9,
               ms * 1e6 / ((double)G * n * (double)b_stride / 64.0 / 1024.0) );
    }
    if (check) {
        std::vector<uint2> hV(hB.size());
        CK(hipMemcpy(hV.data(), dV, hV.size() * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int g = 0; g < checked_groups; ++g)
            for (int p = 0; p < 32; ++p) {
                int score = n;
                for (int r = 0; r < mlen[g * 32 + p]; ++r) score += (int)((hV[g * b_stride + r].x >> p) & 1) - (int)((hV[g * b_stride + r].y >> p) & 1);
                const int want = edit_distance(sa[g * 32 + p], sb[g * 32 + p]);
                if (score != want) {
                    if (bad < 5) printf("MISMATCH group %d pair %d: got %d want %d (m %d)\n", g, p, score, want, mlen[g * 32 + p]);
                    ++bad;
                }
            }
        printf("check: %d pairs, %d mismatches\n", checked_groups * 32, bad);
        return bad ? 1 : 0;
    }
    return 0;
}
