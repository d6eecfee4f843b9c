// sketch_unit.hip -- a divergence sketch per pair, for the START ORDER of the batched band search (apa2_kernel.hpp, apa2_full_kernel.hpp).
//
// Why: one wavefront runs a pair's whole band search, and a pair at 15 % divergence takes several times as long as one at 1 % (band
// doubling: 4 passes of doubling width against 1; astarpa2/src/band.rs:100-141).  A persistent grid that starts pairs in an order that
// does not know this ends with a few wavefronts finishing the expensive pairs while the rest of the chip idles: C4 (10 000 x 10 kbp at
// 1 / 5 / 10 / 15 %) 11.4 ms in the order of the lengths against 10.0 ms most-divergent-first, 8.5 against 11.4 ms with two blocks per
// strip (profiles/r05_runs/order_probe.log).  The reference aligns one pair at a time (pa-bin/src/main.rs:24-35): no counterpart.
//
// How: one wavefront per pair; lane l takes the 16-mer of a at position l (n - 16) / 63 and looks for it in b within a window around the
// scaled diagonal (rolling 32-bit key over the 2-bit codes, qgrams.rs:30-43); the number of lanes that find theirs estimates
// (1 - e)^16.  A few thousand instructions per pair, once per batch; the estimate only orders the queue -- results never depend on it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "apa2_units.hpp"

namespace pa {
namespace apa2 {

constexpr int kSketchK = 16;

__global__ __launch_bounds__(256) void sketch_kernel(const uint8_t* __restrict__ a_cat, const uint8_t* __restrict__ b_cat, const SketchDesc* __restrict__ desc, int npairs,
                                                     uint8_t* __restrict__ found_out) {
    const int pair = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (pair >= npairs) return;
    const int lane = (int)(threadIdx.x & 63);
    const SketchDesc d = desc[pair];
    const int n = d.n, m = d.m;
    bool found = false;
    if (n >= kSketchK + 63 && m >= kSketchK) {
        const uint8_t* a = a_cat + d.a_off;
        const uint8_t* b = b_cat + d.b_off;
        const int p = (int)(((long long)lane * (n - kSketchK)) / 63);
        uint32_t key = 0;
        for (int t = 0; t < kSketchK; ++t) key = (key << 2) | ((uint32_t)(a[p + t] >> 1) & 3u);
        const int jc = (int)(((long long)p * m) / n);
        int w = 48 + n / 64;
        w = w > 1024 ? 1024 : w;
        int j0 = jc - w, j1 = jc + w;  // first characters of the candidates
        j0 = j0 < 0 ? 0 : j0;
        j1 = j1 > m - kSketchK ? m - kSketchK : j1;
        if (j0 <= j1) {
            uint32_t q = 0;
            for (int t = 0; t < kSketchK - 1; ++t) q = (q << 2) | ((uint32_t)(b[j0 + t] >> 1) & 3u);
            int j = j0;
            // four candidates per (unaligned) 4-byte load; the last few one by one (the load must stay inside b)
            for (; j + 3 <= j1; j += 4) {
                uint32_t w;
                __builtin_memcpy(&w, b + j + kSketchK - 1, 4);
                const uint32_t c4 = (w >> 1) & 0x03030303u;
                q = (q << 2) | (c4 & 3u);
                found = found || q == key;
                q = (q << 2) | ((c4 >> 8) & 3u);
                found = found || q == key;
                q = (q << 2) | ((c4 >> 16) & 3u);
                found = found || q == key;
                q = (q << 2) | (c4 >> 24);
                found = found || q == key;
            }
            for (; j <= j1; ++j) {
                q = (q << 2) | ((uint32_t)(b[j + kSketchK - 1] >> 1) & 3u);
                found = found || q == key;
            }
        }
    }
    const uint64_t mask = __ballot(found);
    if (lane == 0) found_out[pair] = (uint8_t)__builtin_popcountll(mask);
}

hipError_t launch_sketch_kernel(hipStream_t s, const uint8_t* a_cat, const uint8_t* b_cat, const SketchDesc* desc, int npairs, uint8_t* found_out) {
    if (npairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(sketch_kernel, dim3((unsigned)((npairs + 3) / 4)), dim3(256), 0, s, a_cat, b_cat, desc, npairs, found_out);
    return hipGetLastError();
}

}  // namespace apa2
}  // namespace pa
