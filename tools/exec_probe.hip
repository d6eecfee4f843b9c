// Does a wave64 VALU instruction issue faster when only half (or a quarter) of EXEC is set?  One wavefront per SIMD at most;
// prints shader clocks per instruction for an independent stream and for a dependent chain.
// hipcc --offload-arch=gfx950 -O3 -o tools/exec_probe tools/exec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int MODE>  // 0 independent xor stream, 1 dependent chain, 2 dpp + dependent mix
__device__ __forceinline__ unsigned body(unsigned x, int iters) {
    unsigned a = x, b = x * 3u + 1u, c = x * 5u + 2u, d = x * 7u + 3u, e = x * 11u, f = x * 13u, g = x * 17u, h = x * 19u;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile(
                "v_xor_b32 %0, %4, %0\n v_xor_b32 %1, %5, %1\n v_xor_b32 %2, %6, %2\n v_xor_b32 %3, %7, %3\n"
                "v_xor_b32 %0, %5, %0\n v_xor_b32 %1, %6, %1\n v_xor_b32 %2, %7, %2\n v_xor_b32 %3, %4, %3\n"
                "v_xor_b32 %0, %6, %0\n v_xor_b32 %1, %7, %1\n v_xor_b32 %2, %4, %2\n v_xor_b32 %3, %5, %3\n"
                "v_xor_b32 %0, %7, %0\n v_xor_b32 %1, %4, %1\n v_xor_b32 %2, %5, %2\n v_xor_b32 %3, %6, %3\n"
                : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h));
        } else if (MODE == 1) {
            asm volatile(
                "v_xor_b32 %0, %1, %0\n v_add_u32 %0, %2, %0\n v_xor_b32 %0, %3, %0\n v_add_u32 %0, %4, %0\n"
                "v_xor_b32 %0, %1, %0\n v_add_u32 %0, %2, %0\n v_xor_b32 %0, %3, %0\n v_add_u32 %0, %4, %0\n"
                "v_xor_b32 %0, %1, %0\n v_add_u32 %0, %2, %0\n v_xor_b32 %0, %3, %0\n v_add_u32 %0, %4, %0\n"
                "v_xor_b32 %0, %1, %0\n v_add_u32 %0, %2, %0\n v_xor_b32 %0, %3, %0\n v_add_u32 %0, %4, %0\n"
                : "+v"(a) : "v"(e), "v"(f), "v"(g), "v"(h));
        } else {
            asm volatile(
                "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96\n v_alignbit_b32 %0, %3, %0, 31\n v_bitop3_b32 %0, %1, %4, %0 bitop3:0xca\n v_add_u32 %0, %4, %0\n"
                "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96\n v_alignbit_b32 %0, %3, %0, 31\n v_bitop3_b32 %0, %1, %4, %0 bitop3:0xca\n v_add_u32 %0, %4, %0\n"
                "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96\n v_alignbit_b32 %0, %3, %0, 31\n v_bitop3_b32 %0, %1, %4, %0 bitop3:0xca\n v_add_u32 %0, %4, %0\n"
                "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96\n v_alignbit_b32 %0, %3, %0, 31\n v_bitop3_b32 %0, %1, %4, %0 bitop3:0xca\n v_add_u32 %0, %4, %0\n"
                : "+v"(a) : "v"(e), "v"(f), "v"(g), "v"(h));
        }
    }
    return a ^ b ^ c ^ d;
}

template <int MODE>
__global__ __launch_bounds__(64) void probe(unsigned* out, unsigned long long* times, int iters, int lo, int hi) {
    const int lane = threadIdx.x & 63;
    unsigned x = threadIdx.x * 2654435761u + 12345u;
    unsigned r = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (lane >= lo && lane < hi) r = body<MODE>(x, iters);
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) times[blockIdx.x] = t1 - t0;
}

int main() {
    unsigned* d_out; unsigned long long* d_t;
    const int blocks = 256;
    CK(hipMalloc(&d_out, blocks * 64 * 4)); CK(hipMalloc(&d_t, blocks * 8));
    const int iters = 20000;
    const int ranges[][2] = {{0, 64}, {32, 64}, {0, 32}, {48, 64}, {0, 16}, {16, 48}, {0, 1}};
    for (int mode = 0; mode < 3; ++mode)
        for (auto& rg : ranges) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64), 0, 0, d_out, d_t, iters, rg[0], rg[1]);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64), 0, 0, d_out, d_t, iters, rg[0], rg[1]);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(64), 0, 0, d_out, d_t, iters, rg[0], rg[1]);
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> t(blocks);
            CK(hipMemcpy(t.data(), d_t, blocks * 8, hipMemcpyDeviceToHost));
            double cyc = 0; for (auto v : t) cyc += (double)v;
            printf("mode %d (%s) lanes [%2d,%2d): %.3f ns / instruction (wall), %.2f counter ticks / instruction\n", mode,
                   mode == 0 ? "independent" : mode == 1 ? "dependent" : "dependent mixed", rg[0], rg[1], ms * 1e6 / ((double)iters * 16),
                   cyc / blocks / ((double)iters * 16));
        }
    return 0;
}
