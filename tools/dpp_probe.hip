// Tiny probe: does v_mov_b32_dpp wave_shr:1 behave as expected on this GPU (lane i <- lane i-1, lane 0 keeps old)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned x = threadIdx.x * 3 + 1;
    unsigned y = (unsigned)__builtin_amdgcn_update_dpp((int)777, (int)x, 0x138, 0xf, 0xf, false);
    out[threadIdx.x] = y;
}
int main() {
    unsigned* d; unsigned h[64];
    if (hipMalloc(&d, 256) != hipSuccess) { printf("no device\n"); return 2; }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { unsigned want = i == 0 ? 777u : (unsigned)((i - 1) * 3 + 1); if (h[i] != want) bad++; }
    printf("lane0=%u lane1=%u lane32=%u lane63=%u bad=%d\n", h[0], h[1], h[32], h[63], bad);
    return bad != 0;
}
