/*
 * oracle/sweep_emu_tsan.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Race detection for the sweep's hand-off protocol: the wave program on host threads (sweep_emu.cpp), several passes in flight,
 * built with -fsanitize=thread (`make -C oracle tsan`, then `PA_SWEEP_EMU_DEPTH=3 [PA_SWEEP_TEST_GIVE_UP=2] oracle/_build/sweep_emu_tsan 14`).
 * Every word shared between wavefronts and between passes goes through the policy's atomics, so ThreadSanitizer sees the same
 * release / acquire pairs the device relies on (tag-carrying words, prefix words behind drained stores, done words behind the
 * merge).  Clean as of round 2.
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../include/pa_astarpa2.h"
extern "C" int pa_sweep_emu_align(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, const pa_astarpa2_params* params, int trace, int nwaves, int32_t* cost_out, char** cigar_out, pa_astarpa2_stats* stats_out, int32_t* info);
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 6;
    for (int it = 0; it < iters; ++it) {
        const int n = 2000 + rnd() % 9000;
        std::string a(n, 'A'), b;
        for (auto& c : a) c = "ACGT"[rnd() & 3];
        const int e = 5 + rnd() % 30;
        for (int i = 0; i < n; ++i) {
            const uint32_t r = rnd() % 100;
            if (r < (uint32_t)e / 3) continue;
            if (r < 2 * (uint32_t)e / 3) { b.push_back("ACGT"[rnd() & 3]); continue; }
            if (r < (uint32_t)e) b.push_back("ACGT"[rnd() & 3]);
            b.push_back(a[i]);
        }
        pa_astarpa2_params p{};
        p.domain = 3; p.heuristic = (it % 3 == 0) ? 0 : 1; p.heuristic_k = 12; p.doubling = 1; p.doubling_start = 2; p.factor = 2.0f; p.delta = 1.0f;
        p.block_width = 256; p.front.sparse = 1; p.front.simd = 1; p.front.dt_trace = 1; p.front.max_g = 40; p.front.fr_drop = 10; p.sparse_h = 1;
        int32_t cost = -1, info[16] = {0};
        char* cig = nullptr;
        pa_astarpa2_stats st{};
        const int rc = pa_sweep_emu_align((const uint8_t*)a.data(), a.size(), (const uint8_t*)b.data(), b.size(), &p, it & 1, 8, &cost, &cig, &st, info);
        printf("it %d n %d rc %d cost %d tries %llu\n", it, n, rc, cost, (unsigned long long)st.f_max_tries);
        free(cig);
    }
    return 0;
}
