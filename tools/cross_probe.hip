// cross_probe.hip -- issue-cost probe for the crossing-chunk variants of the sweep kernel (one wavefront, wall clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../astar-pairwise-aligner_amd/csrc/sweep_kernel.hpp"
using namespace pa::sweep;
typedef DeviceWave W;

template <int VAR>
__global__ __launch_bounds__(64) void probe(uint32_t* out, uint64_t* t, int iters, int j0, int j1) {
    uint32_t lane = threadIdx.x & 63, vp = ~0u, vm = 0, nb0 = lane * 2654435761u, nb1 = ~nb0, X = 0, alo = 0, ahi = 0, sp = 0, sm = 0;
    uint32_t andm = ~0u, orm = 0, resetm = lane > 60, fpend = lane == 4;
    uint32_t XS = lane * 7 + 0x80000000u;
    const uint64_t t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (VAR == 0) W::chunk<false>(XS, X, vp, vm, nb0, nb1, alo, ahi, andm, orm);
        if (VAR == 1) W::chunk<true>(XS, X, vp, vm, nb0, nb1, alo, ahi, andm, orm);
        if (VAR == 2) W::chunk_cross<false>(XS, X, vp, vm, nb0, nb1, alo, ahi, andm, orm, lane, it & 32, sp, sm, resetm, fpend, 0, 32);
        if (VAR == 3) W::chunk_cross<true>(XS, X, vp, vm, nb0, nb1, alo, ahi, andm, orm, lane, it & 32, sp, sm, resetm, fpend, 0, 32);
        if (VAR == 4) W::chunk_cross<true>(XS, X, vp, vm, nb0, nb1, alo, ahi, andm, orm, lane, it & 32, sp, sm, resetm, fpend, j0, j1);
        if (VAR == 5) {
            W::chunk_cross<true>(XS, X, vp, vm, nb0, nb1, alo, ahi, andm, orm, lane, it & 32, sp, sm, resetm, fpend, 0, j0);
            W::chunk_cross<true>(XS, X, vp, vm, nb0, nb1, alo, ahi, andm, orm, lane, it & 32, sp, sm, resetm, fpend, j0, j1);
            W::chunk_cross<true>(XS, X, vp, vm, nb0, nb1, alo, ahi, andm, orm, lane, it & 32, sp, sm, resetm, fpend, j1, 32);
        }
        XS += X;
    }
    const uint64_t t1 = wall_clock64();
    out[threadIdx.x] = vp ^ vm ^ sp ^ sm ^ alo ^ ahi ^ andm ^ orm;
    if (threadIdx.x == 0) *t = t1 - t0;
}

int main() {
    uint32_t* d_out;
    uint64_t* d_t;
    hipMalloc(&d_out, 256);
    hipMalloc(&d_t, 8);
    const int iters = 2000;
    const char* names[] = {"chunk<plain>", "chunk<force>", "cross<noforce> full", "cross<force> full", "cross<force> switch 1..31", "cross<force> 3 switch calls 0..9,9..12,12..32"};
    for (int v = 0; v < 6; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (v) {
                case 0: hipLaunchKernelGGL(probe<0>, 1, 64, 0, 0, d_out, d_t, iters, 1, 31); break;
                case 1: hipLaunchKernelGGL(probe<1>, 1, 64, 0, 0, d_out, d_t, iters, 1, 31); break;
                case 2: hipLaunchKernelGGL(probe<2>, 1, 64, 0, 0, d_out, d_t, iters, 1, 31); break;
                case 3: hipLaunchKernelGGL(probe<3>, 1, 64, 0, 0, d_out, d_t, iters, 1, 31); break;
                case 4: hipLaunchKernelGGL(probe<4>, 1, 64, 0, 0, d_out, d_t, iters, 1, 31); break;
                case 5: hipLaunchKernelGGL(probe<5>, 1, 64, 0, 0, d_out, d_t, iters, 9, 12); break;
            }
            hipDeviceSynchronize();
        }
        uint64_t t;
        hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
        printf("%-50s %7.1f ns/step (%.2f us/chunk)\n", names[v], t * 10.0 / iters / 32, t * 0.01 / iters);
    }
    return 0;
}
