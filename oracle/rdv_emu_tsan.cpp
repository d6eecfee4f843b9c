/*
 * oracle/rdv_emu_tsan.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Race detection for the rendezvous of half-wave blocks (csrc/rdv_logic.hpp) on host threads: oracle/rdv_emu.cpp built with
 * -fsanitize=thread (`make -C oracle tsan_rdv`, then `oracle/_build/rdv_emu_tsan [pairs] [groups] [patience_us]`).  The two wavefronts of
 * a fused strip share the poster's column, its sum and its mail slot; the only synchronisation is the protocol's shared word.  The run
 * also compares every pair's cost and counters with a run without any rendezvous and exits non-zero on a difference.
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
extern "C" int pa_rdv_emu_run(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len, size_t npairs, int groups,
                              double patience_us, int64_t* out, uint64_t* counters);
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 120;
    const int groups = argc > 2 ? atoi(argv[2]) : 2;
    const double patience = argc > 3 ? atof(argv[3]) : 200.0;
    std::vector<std::string> as(n), bs(n);
    for (size_t i = 0; i < n; ++i) {
        const int len = 300 + (int)(rnd() % 3000);
        const int e = (int)(rnd() % 25);
        for (int k = 0; k < len; ++k) {
            const char c = "ACGT"[rnd() & 3];
            as[i].push_back(c);
            const uint32_t r = rnd() % 100;
            if (r < (uint32_t)e / 3) continue;
            if (r < 2 * (uint32_t)e / 3) bs[i].push_back("ACGT"[rnd() & 3]);
            else if (r < (uint32_t)e) { bs[i].push_back("ACGT"[rnd() & 3]); continue; }
            bs[i].push_back(c);
        }
        if (bs[i].empty()) bs[i] = "A";
    }
    std::vector<const uint8_t*> ap(n), bp(n);
    std::vector<size_t> al(n), bl(n);
    for (size_t i = 0; i < n; ++i) {
        ap[i] = (const uint8_t*)as[i].data();
        bp[i] = (const uint8_t*)bs[i].data();
        al[i] = as[i].size();
        bl[i] = bs[i].size();
    }
    std::vector<int64_t> alone(8 * n), fused(8 * n);
    uint64_t c0[4], c1[4];
    pa_rdv_emu_run(ap.data(), al.data(), bp.data(), bl.data(), n, groups, -1.0, alone.data(), c0);
    pa_rdv_emu_run(ap.data(), al.data(), bp.data(), bl.data(), n, groups, patience, fused.data(), c1);
    int bad = 0;
    for (size_t i = 0; i < 8 * n; ++i) bad += alone[i] != fused[i];
    printf("pairs %zu groups %d patience %.0f us: fused %llu served %llu alone %llu withdrawn %llu; differences %d\n", n, groups, patience,
           (unsigned long long)c1[0], (unsigned long long)c1[1], (unsigned long long)c1[2], (unsigned long long)c1[3], bad);
    return bad || c1[0] != c1[1] ? 1 : 0;
}
