// Per-opcode issue cost on gfx950 at 1 and 8 waves/SIMD (inline asm, 8 independent destination registers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
constexpr int ITER = 3000;
// 8 ops, op i writes %i reading %i, %8 (const) and %((i+1)%8)
#define R8(F) F("%0","%1") F("%1","%2") F("%2","%3") F("%3","%4") F("%4","%5") F("%5","%6") F("%6","%7") F("%7","%0")
#define F_XOR(d,s)   "v_xor_b32 " d ", " d ", %8\n"
#define F_AND(d,s)   "v_and_b32 " d ", " d ", " s "\n"
#define F_ADD(d,s)   "v_add_u32 " d ", " d ", " s "\n"
#define F_LSHL(d,s)  "v_lshlrev_b32 " d ", 1, " d "\n"
#define F_LSHR(d,s)  "v_lshrrev_b32 " d ", 31, " d "\n"
#define F_BFEI(d,s)  "v_bfe_i32 " d ", " d ", 1, 1\n"
#define F_BFEU(d,s)  "v_bfe_u32 " d ", " d ", 30, 1\n"
#define F_ALIGN(d,s) "v_alignbit_b32 " d ", " d ", " s ", 31\n"
#define F_ANDOR(d,s) "v_and_or_b32 " d ", " d ", " s ", %8\n"
#define F_OR3(d,s)   "v_or3_b32 " d ", " d ", " s ", %8\n"
#define F_LSHLOR(d,s) "v_lshl_or_b32 " d ", " d ", 1, " s "\n"
#define F_B3_3V(d,s) "v_bitop3_b32 " d ", " d ", " s ", %8 bitop3:0x96\n"
#define F_B3_2V(d,s) "v_bitop3_b32 " d ", " d ", " s ", " s " bitop3:0x96\n"
#define F_B3_SGPR(d,s) "v_bitop3_b32 " d ", %9, " d ", " s " bitop3:0xca\n"
#define F_BFI(d,s)   "v_bfi_b32 " d ", " d ", " s ", %8\n"
#define F_XOR3V(d,s) "v_xor_b32 " d ", " d ", " s "\n"
#define F_MOVS(d,s)  "v_mov_b32 " d ", %9\n"
#define F_DPP(d,s)   "v_mov_b32_dpp " d ", " s " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_DPPROW(d,s) "v_mov_b32_dpp " d ", " s " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_CNDMASK(d,s) "v_cndmask_b32 " d ", " d ", " s ", vcc\n"

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned* out, unsigned long long* times, unsigned seed) {
    unsigned r0 = threadIdx.x + seed, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19, c = r0 ^ 0x55555555u;
    unsigned sk = seed * 77u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#define BODY(F) asm volatile(R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) R8(F) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c), "s"(sk) : "vcc")
        if (MODE == 0) BODY(F_XOR);
        if (MODE == 1) BODY(F_AND);
        if (MODE == 2) BODY(F_ADD);
        if (MODE == 3) BODY(F_LSHL);
        if (MODE == 4) BODY(F_LSHR);
        if (MODE == 5) BODY(F_BFEI);
        if (MODE == 6) BODY(F_BFEU);
        if (MODE == 7) BODY(F_ALIGN);
        if (MODE == 8) BODY(F_ANDOR);
        if (MODE == 9) BODY(F_OR3);
        if (MODE == 10) BODY(F_LSHLOR);
        if (MODE == 11) BODY(F_B3_3V);
        if (MODE == 12) BODY(F_B3_2V);
        if (MODE == 13) BODY(F_B3_SGPR);
        if (MODE == 14) BODY(F_BFI);
        if (MODE == 15) BODY(F_XOR3V);
        if (MODE == 16) BODY(F_MOVS);
        if (MODE == 17) BODY(F_DPP);
        if (MODE == 18) BODY(F_DPPROW);
        if (MODE == 19) BODY(F_CNDMASK);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    if (threadIdx.x == 0) times[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned* d_out, unsigned long long* d_times) {
    double res[2];
    int idx = 0;
    for (int blocks : {1024, 8192}) {
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, d_times, 1u);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, d_times, 2u);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(blocks);
        CK(hipMemcpy(t.data(), d_times, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
        double cyc = 0;
        for (int b = 0; b < blocks; ++b) cyc += (double)t[b];
        res[idx++] = cyc / blocks / ((double)ITER * 64) / (blocks / 1024.0);
    }
    printf("%-28s  1 wave/SIMD: %5.2f ticks/op   8 waves/SIMD: %5.2f ticks/op/SIMD\n", name, res[0], res[1]);
}

int main() {
    unsigned* d_out; unsigned long long* d_times;
    CK(hipMalloc(&d_out, 64 * 16384 * 4)); CK(hipMalloc(&d_times, 16384 * 8));
    run<0>("v_xor_b32 (vop2, 2 vgpr)", d_out, d_times);
    run<15>("v_xor_b32 (2 distinct vgpr)", d_out, d_times);
    run<1>("v_and_b32", d_out, d_times);
    run<2>("v_add_u32", d_out, d_times);
    run<3>("v_lshlrev_b32 imm", d_out, d_times);
    run<4>("v_lshrrev_b32 imm", d_out, d_times);
    run<5>("v_bfe_i32", d_out, d_times);
    run<6>("v_bfe_u32", d_out, d_times);
    run<7>("v_alignbit_b32", d_out, d_times);
    run<8>("v_and_or_b32 (3 vgpr)", d_out, d_times);
    run<9>("v_or3_b32 (3 vgpr)", d_out, d_times);
    run<10>("v_lshl_or_b32", d_out, d_times);
    run<11>("v_bitop3 (3 distinct vgpr)", d_out, d_times);
    run<12>("v_bitop3 (2 distinct vgpr)", d_out, d_times);
    run<13>("v_bitop3 (sgpr + 2 vgpr)", d_out, d_times);
    run<14>("v_bfi_b32 (3 vgpr)", d_out, d_times);
    run<16>("v_mov_b32 from sgpr", d_out, d_times);
    run<17>("v_mov_dpp wave_shr:1", d_out, d_times);
    run<18>("v_mov_dpp row_shr:1", d_out, d_times);
    run<19>("v_cndmask_b32 vcc", d_out, d_times);
    return 0;
}
