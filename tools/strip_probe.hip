// Standalone probe for the strip kernel with host-visible progress markers (PA_STRIP_DEBUG).
// Runs one tiny rectangle (n columns x w words, h = v = +1) and reports where the kernel is if it stalls.
#define PA_STRIP_DEBUG 1
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <vector>
__device__ unsigned int* g_pa_dbg = nullptr;
#include "../astar-pairwise-aligner_amd/csrc/strip_kernel.hpp"
using namespace pa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16, w = argc > 2 ? atoi(argv[2]) : 1, K = argc > 3 ? atoi(argv[3]) : 1;
    unsigned int* dbg_host = nullptr;
    CK(hipHostMalloc((void**)&dbg_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(dbg_host, 0, 64);
    unsigned int* dbg_dev = nullptr;
    CK(hipHostGetDevicePointer((void**)&dbg_dev, dbg_host, 0));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pa_dbg), &dbg_dev, sizeof(dbg_dev)));
    const int cw = (n + 15) / 16 + 2, WPS = 32 * K, S = (w + WPS - 1) / WPS, G = (n + 31) / 32;
    std::vector<uint32_t> codes(cw, 0x1B1B1B1B);
    std::vector<uint64_t> prof(2 * w, 0x0123456789ABCDEFull), v(2 * w);
    for (int j = 0; j < w; ++j) { v[2 * j] = ~0ull; v[2 * j + 1] = 0; }
    uint32_t *d_codes, *d_prof, *d_v, *d_misc; uint64_t* d_gran; StripJob* d_jobs;
    CK(hipMalloc(&d_codes, cw * 4)); CK(hipMalloc(&d_prof, w * 16)); CK(hipMalloc(&d_v, w * 16));
    CK(hipMalloc(&d_misc, 64)); CK(hipMalloc(&d_gran, (size_t)(S * G + 1) * 8)); CK(hipMalloc(&d_jobs, S * sizeof(StripJob)));
    CK(hipMemcpy(d_codes, codes.data(), cw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_prof, prof.data(), w * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_v, v.data(), w * 16, hipMemcpyHostToDevice));
    CK(hipMemset(d_misc, 0, 64)); CK(hipMemset(d_gran, 0, (size_t)(S * G + 1) * 8));
    std::vector<StripJob> jobs(S);
    for (int s = 0; s < S; ++s) {
        StripJob j; memset(&j, 0, sizeof j);
        j.a_codes = d_codes; j.b_prof = d_prof; j.v = d_v; j.n = n; j.word0 = WPS * s;
        j.nlanes = 2 * (w - WPS * s < WPS ? w - WPS * s : WPS);
        if (s > 0) j.hin_gran = d_gran + (size_t)(s - 1) * G;
        if (s + 1 < S) j.hout_gran = d_gran + (size_t)s * G; else j.sum_out = (int32_t*)d_misc + 2;
        j.exact_tail = (K == 4 && (j.nlanes % 4)) ? 0 : 1;
        j.tail_rows = -1;
        jobs[s] = j;
    }
    CK(hipMemcpy(d_jobs, jobs.data(), S * sizeof(StripJob), hipMemcpyHostToDevice));
    if (K == 1) hipLaunchKernelGGL((strip_kernel<1, false, false>), dim3((S + 3) / 4), dim3(256), 0, 0, d_jobs, S, d_misc, d_misc + 1);
    else if (K == 2) hipLaunchKernelGGL((strip_kernel<2, false, false>), dim3((S + 3) / 4), dim3(256), 0, 0, d_jobs, S, d_misc, d_misc + 1);
    else hipLaunchKernelGGL((strip_kernel<4, false, false>), dim3((S + 3) / 4), dim3(256), 0, 0, d_jobs, S, d_misc, d_misc + 1);
    CK(hipGetLastError());
    hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, 0));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        if (hipEventQuery(ev) == hipSuccess) break;
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt > 10.0) {
            printf("STALL after %.1fs: dbg[0]=0x%x (ticket marker) dbg[1]=%u (stage) dbg[2]=%u (chunk q+1)\n", dt, dbg_host[0], dbg_host[1], dbg_host[2]);
            fflush(stdout);
            _exit(3);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    uint32_t misc[4];
    CK(hipMemcpy(misc, d_misc, 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(v.data(), d_v, w * 16, hipMemcpyDeviceToHost));
    printf("done n=%d w=%d K=%d: ticket=%u err=%u sum=%d v0=(%016llx,%016llx) dbg=[0x%x,%u,%u]\n", n, w, K, misc[0], misc[1], (int)misc[2],
           (unsigned long long)v[0], (unsigned long long)v[1], dbg_host[0], dbg_host[1], dbg_host[2]);
    return 0;
}
