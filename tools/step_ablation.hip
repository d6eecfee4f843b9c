// Ablation probe of the strip kernel's hot step (results are garbage by design in ablated builds): W waves per SIMD of
// independent single-strip jobs (no hand-off), reports ns and shader-clock ticks per step per SIMD.
// Build variants with -DPA_ABLATE=<bits>: 1 = no per-step readlane, 2 = no DPP shift, 4 = no accumulate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../astar-pairwise-aligner_amd/csrc/strip_kernel.hpp"
using namespace pa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
__global__ void clock_probe(unsigned long long* out) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    unsigned x = threadIdx.x;
    for (int i = 0; i < 200000; ++i) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(x));
    if (threadIdx.x == 0) { out[0] = __builtin_readcyclecounter() - t0; out[1] = wall_clock64() - w0; }
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20000;
#ifndef PA_ABLATE
#define PA_ABLATE 0
#endif
    const int cw = (n + 15) / 16 + 2;
    std::vector<uint32_t> codes(cw);
    for (int i = 0; i < cw; ++i) codes[i] = 0x1B1B1B1Bu * (i + 1);
    uint32_t *d_codes, *d_prof, *d_v, *d_misc; StripJob* d_jobs; int32_t* d_sums;
    const int maxjobs = 8192;
    CK(hipMalloc(&d_codes, cw * 4)); CK(hipMalloc(&d_prof, 32 * 16)); CK(hipMalloc(&d_v, (size_t)maxjobs * 32 * 16));
    CK(hipMalloc(&d_misc, 64)); CK(hipMalloc(&d_jobs, maxjobs * sizeof(StripJob))); CK(hipMalloc(&d_sums, maxjobs * 4));
    CK(hipMemcpy(d_codes, codes.data(), cw * 4, hipMemcpyHostToDevice));
    std::vector<uint64_t> prof(64, 0x0123456789ABCDEFull);
    CK(hipMemcpy(d_prof, prof.data(), 32 * 16, hipMemcpyHostToDevice));
    for (int W : {1, 2, 4, 7, 8}) {
        const int jobsn = 1024 * W;
        std::vector<StripJob> jobs(jobsn);
        for (int i = 0; i < jobsn; ++i) {
            StripJob j; memset(&j, 0, sizeof j);
            j.a_codes = d_codes; j.b_prof = d_prof; j.v = d_v + (size_t)i * 32 * 4; j.n = n; j.word0 = 0; j.nlanes = 64;
            j.sum_out = d_sums + i; j.exact_tail = 0; j.tail_rows = -1; j.flags = kJobVInitOne;
            jobs[i] = j;
        }
        CK(hipMemcpy(d_jobs, jobs.data(), jobsn * sizeof(StripJob), hipMemcpyHostToDevice));
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(d_misc, 0, 64));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((strip_kernel<false, false>), dim3(jobsn), dim3(64), 0, 0, d_jobs, jobsn, d_misc, d_misc + 1);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // shader-clock measurement: a tiny kernel run right after (same DVFS state is not guaranteed, so also time-stamp
        // inside the strip kernel through PA_DBG-free means: use wall clock ratio of a calibration kernel)
        unsigned long long* d_clk; unsigned long long clk[2];
        CK(hipMalloc(&d_clk, 16));
        hipLaunchKernelGGL(clock_probe, dim3(8192), dim3(64), 0, 0, d_clk);
        CK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
        const double steps_per_simd = (double)W * (n + 64);
        printf("[clock under full VALU load right after: %.2f GHz] ", ghz);
        printf("ablate=%d W=%d waves/SIMD  kernel %.3f ms  %.1f ns per step per SIMD  (%.1f ns per step per wave)\n", PA_ABLATE, W, best,
               best * 1e6 / steps_per_simd, best * 1e6 / (n + 64));
    }
    return 0;
}
