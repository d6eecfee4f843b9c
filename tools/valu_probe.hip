// Micro-probe: issue cost of the instruction classes the strip kernel uses (gfx950).
// Each kernel runs ITER iterations of an unrolled body of 64 ops per lane; reports ns and cycles/op/wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
constexpr int ITER = 20000;

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned* out, unsigned seed) {
    unsigned a = threadIdx.x * 7 + seed, b = a ^ 0x9e3779b9u, c = a + 12345u, d = b * 3u;
    unsigned e = a + 1, f = b + 2, g = c + 3, h = d + 4;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {  // dependent chain, VOP2 logic
                a = a & b; a = a | c; a = a ^ d; a = a & e; a = a | f; a = a ^ g; a = a & h; a = a | b;
            } else if (MODE == 1) {  // 4 independent chains, VOP2 logic
                a = a ^ e; b = b ^ f; c = c ^ g; d = d ^ h; a = a | f; b = b | g; c = c | h; d = d | e;
            } else if (MODE == 2) {  // dependent chain of 3-input ops (bitop3 / and_or / or3)
                a = (a & b) | c; a = (a | d) ^ e; a = (a ^ f) & g; a = (a & h) | b; a = (a | c) ^ d; a = (a ^ e) & f; a = (a & g) | h; a = (a | b) ^ c;
            } else if (MODE == 3) {  // dependent adds
                a = a + b; a = a + c; a = a + d; a = a + e; a = a + f; a = a + g; a = a + h; a = a + b;
            } else if (MODE == 4) {  // dpp wave_shr chain
                a = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x138, 0xf, 0xf, false) ^ c;
                a = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x138, 0xf, 0xf, false) ^ d;
                a = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x138, 0xf, 0xf, false) ^ e;
                a = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x138, 0xf, 0xf, false) ^ f;
            } else if (MODE == 5) {  // readlane -> v_mov -> use
                unsigned s0 = (unsigned)__builtin_amdgcn_readlane((int)a, 5); a = (a ^ s0) + b;
                unsigned s1 = (unsigned)__builtin_amdgcn_readlane((int)a, 6); a = (a ^ s1) + c;
                unsigned s2 = (unsigned)__builtin_amdgcn_readlane((int)a, 7); a = (a ^ s2) + d;
                unsigned s3 = (unsigned)__builtin_amdgcn_readlane((int)a, 8); a = (a ^ s3) + e;
            } else if (MODE == 6) {  // shifts + bfe dependent
                a = (a >> 1) | b; a = (a << 1) ^ c; a = (unsigned)((int)a >> 31) ^ d; a = ((a >> 3) & 1u) | e; a = (a << 2) | f; a = (a >> 30) ^ g; a = (a << 1) | h; a = (a >> 2) ^ b;
            } else if (MODE == 7) {  // row_shr:1 dpp chain (for comparison)
                a = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x111, 0xf, 0xf, false) ^ c;
                a = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x111, 0xf, 0xf, false) ^ d;
                a = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x111, 0xf, 0xf, false) ^ e;
                a = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x111, 0xf, 0xf, false) ^ f;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int MODE>
void run(const char* name, int ops_per_body, int blocks, unsigned* d_out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, 2u);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)ITER * 8 * ops_per_body;
    printf("%-28s blocks=%5d  %8.3f ms  %7.3f ns/op/wave  (%.2f cycles @2.4GHz)\n", name, blocks, ms, ms * 1e6 / ops, ms * 1e6 / ops * 2.4);
}

int main() {
    unsigned* d_out; CK(hipMalloc(&d_out, 64 * 16384 * 4));
    for (int blocks : {64, 1024, 2048, 8192}) {
        run<0>("vop2 dependent", 8, blocks, d_out);
        run<1>("vop2 4 chains", 8, blocks, d_out);
        run<2>("3-input (bitop3) dependent", 8, blocks, d_out);
        run<3>("add dependent", 8, blocks, d_out);
        run<4>("dpp wave_shr + xor", 8, blocks, d_out);
        run<7>("dpp row_shr + xor", 8, blocks, d_out);
        run<5>("readlane+xor+add", 12, blocks, d_out);
        run<6>("shift/bfe+logic dependent", 16, blocks, d_out);
        printf("\n");
    }
    return 0;
}
