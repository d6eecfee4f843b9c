// Ground truth for the integer-VALU issue rate and the shader clock under load (gfx950).
// Each wave runs ITER x 64 VALU ops written in inline asm (so nothing is folded away); lane 0 records s_memtime
// (shader clock ticks) and wall_clock64 (100 MHz) deltas.  Reported: cycles per op per wave, per-SIMD ops/cycle, clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
constexpr int ITER = 4000;

#define OPS8_INDEP(OP) \
    OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" \
    OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
#define OPS8_DEP(OP) \
    OP " %0, %0, %8\n" OP " %0, %0, %1\n" OP " %0, %0, %2\n" OP " %0, %0, %3\n" \
    OP " %0, %0, %4\n" OP " %0, %0, %5\n" OP " %0, %0, %6\n" OP " %0, %0, %7\n"
#define B3_8_INDEP \
    "v_bitop3_b32 %0, %0, %8, %1 bitop3:0x96\n" "v_bitop3_b32 %1, %1, %8, %2 bitop3:0x96\n" "v_bitop3_b32 %2, %2, %8, %3 bitop3:0x96\n" "v_bitop3_b32 %3, %3, %8, %4 bitop3:0x96\n" \
    "v_bitop3_b32 %4, %4, %8, %5 bitop3:0x96\n" "v_bitop3_b32 %5, %5, %8, %6 bitop3:0x96\n" "v_bitop3_b32 %6, %6, %8, %7 bitop3:0x96\n" "v_bitop3_b32 %7, %7, %8, %0 bitop3:0x96\n"

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned* out, unsigned long long* times, unsigned seed) {
    unsigned r0 = threadIdx.x + seed, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19, c = r0 ^ 0x55555555u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < ITER; ++it) {
#define BODY(S) asm volatile(S S S S S S S S : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c))
        if (MODE == 0) BODY(OPS8_INDEP("v_xor_b32"));
        if (MODE == 1) BODY(OPS8_DEP("v_xor_b32"));
        if (MODE == 2) BODY(OPS8_INDEP("v_add_u32"));
        if (MODE == 3) BODY(B3_8_INDEP);
        if (MODE == 4) BODY(OPS8_DEP("v_add_u32"));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    if (threadIdx.x == 0) { times[2 * blockIdx.x] = t1 - t0; times[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, int blocks, unsigned* d_out, unsigned long long* d_times) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, d_times, 1u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, d_times, 2u);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(2 * blocks);
    CK(hipMemcpy(t.data(), d_times, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0;
    for (int b = 0; b < blocks; ++b) { cyc += (double)t[2 * b]; wall += (double)t[2 * b + 1]; }
    cyc /= blocks; wall /= blocks;
    const double ops = (double)ITER * 64;
    const double clk_ghz = cyc / (wall * 10.0);  // wall ticks are 10 ns
    const double waves_per_simd = blocks / 1024.0;
    printf("%-22s waves/SIMD=%5.2f  kernel %8.3f ms  %6.2f memtime-ticks/op/wave  clock(memtime/wall)=%.3f GHz  => %.2f ticks per op per SIMD\n",
           name, waves_per_simd, ms, cyc / ops, clk_ghz, cyc / ops / (waves_per_simd < 1 ? 1 : waves_per_simd));
}

int main() {
    unsigned* d_out; unsigned long long* d_times;
    CK(hipMalloc(&d_out, 64 * 16384 * 4)); CK(hipMalloc(&d_times, 16384 * 16));
    for (int blocks : {256, 1024, 2048, 4096, 8192}) {
        run<0>("v_xor indep", blocks, d_out, d_times);
        run<1>("v_xor dependent", blocks, d_out, d_times);
        run<2>("v_add_u32 indep", blocks, d_out, d_times);
        run<4>("v_add_u32 dependent", blocks, d_out, d_times);
        run<3>("v_bitop3 indep", blocks, d_out, d_times);
        printf("\n");
    }
    return 0;
}
