// gcsh_build_kernel.hpp -- the matches of the gap-chaining seed heuristic (GCSH) found ON THE GPU, one wavefront per pair.
//
// What it replaces (paths relative to /root/reference/pa-heuristic/src/; csrc/gcsh.hpp is the host restatement this kernel must equal,
// match for match -- tests/test_gpu_apa2_full.py compares the two):
//   seeds + exact matches   disjoint k-mers of a hashed, every k-mer of b looked up        matches/exact.rs:15-69, qgrams.rs:81-109
//   push filters            the gap-transform filter, then LOCAL PRUNING p: a match is kept only if a bounded diagonal-transition
//                           search from its end crosses the next p seeds with fewer errors than seeds, or runs into the start of a
//                           match kept before it                                           matches.rs:205-247, prepruning.rs:95-203
//   MatchPruner::new        matches by start, one window of matches per seed               prune.rs:132-201
//
// MI355X-first shape, phase by phase (a wavefront owns a pair from its sequences to its matches; nothing is shared between wavefronts):
//  A. the seed k-mers into an open-addressing table in global memory, 64 seeds per round, claimed with compare-and-swap; seeds that
//     share a k-mer hang off the table entry as a chain;
//  B. one lane per position of b: pack the k-mer, probe the table, filter by the transform; the survivors of a round are appended in
//     the reference's push order (rows descending when read backwards) with a ballot prefix;
//  C. a counting sort by seed gives the by-start order (prune.rs wants the matches of a seed side by side);
//  D. local pruning WITHOUT help from other matches, one lane per candidate: the furthest-reaching columns of the candidate's search
//     sit in the lane's column of an 8 KB LDS array (16-bit offsets from the candidate's start, round 6) -- at 5 % divergence nine
//     candidates in ten end here, kept;
//  E. the rest in the reference's order, one after the other, the search spread over the lanes (one diagonal each), with the
//     reference's `next_match_per_diag` -- the latest kept match of a diagonal -- served from a ring of the last 64 kept matches in
//     LDS (a kept match further back than that cannot be met: the search spans p seeds);
//  F. the kept matches compacted in by-start order, the windows of every seed.
// A pair whose candidates outgrow its buffers (a repeat-rich sequence) or whose ring would drop a match still in reach is flagged and
// goes to the host engine, like every other pair the batch kernels hand back.
#pragma once
#include "apa2_jobs.hpp"
#include "apa2_kernel.hpp"
#include "gcsh_dev.hpp"

namespace pa {
namespace apa2 {

typedef int32_t pa_i32x4_b __attribute__((ext_vector_type(4)));  // a GcshSeedWindow as one 16-byte store

__device__ __forceinline__ uint32_t kmer_key(const PA_GLOBAL uint8_t* s, int32_t k) {  // qgrams.rs:30-43: (c >> 1) & 3, first character highest
    uint64_t q = 0;
    int32_t t = 0;
    for (; t + 4 <= k; t += 4) {  // four characters per load: the 2-bit fields of the four bytes gathered by one multiplication
        uint32_t w;
        __builtin_memcpy(&w, (const uint8_t*)s + t, 4);
        const uint32_t y = (w >> 1) & 0x03030303u;
        q = (q << 8) | (uint64_t)((y * 0x40100401u) >> 24);
    }
    for (; t < k; ++t) q = (q << 2) | (uint64_t)((s[t] >> 1) & 3u);
    return (uint32_t)q;
}
__device__ __forceinline__ uint32_t key_hash(uint32_t key, uint32_t mask) { return (key * 0x9E3779B1u) >> 7 & mask; }
// k > 16: the 32-bit key holds the LAST 16 characters only, and that IS the reference's comparison: its map is keyed on `q as u32`
// (matches/exact.rs:47,53,56), so seeds / k-mers of b that agree on their last 16 characters match each other there and here.

constexpr int kBuildWin = 2048;  // bytes of a and of b staged in LDS for a batch of 64 candidates

struct BuildCtx {
    const GcshBuildJob& jb;
    int lane;
    uint32_t div_m;  // x / k by multiplication (exact below 2^26, apa2_full_kernel.hpp div_k)
    int32_t potn;
    // The searches of 64 neighbouring candidates read the same two stretches of a and b over and over, one dependent 4-byte load after
    // the other: the stretches sit in LDS (wa / wb, kBuildWin + 8 bytes each, starting at a0 / b0); a search that leaves them (a
    // candidate far off the others' diagonal) reads global memory.
    const uint32_t* wa = nullptr;
    const uint32_t* wb = nullptr;
    int32_t a0 = 0, b0 = 0;
    __device__ __forceinline__ uint32_t ld4(const uint32_t* w, int32_t w0, const PA_GLOBAL uint8_t* g, int32_t pos) const {
        const int32_t off = pos - w0;
        if (off >= 0 && off + 4 <= kBuildWin) {
            const uint32_t lo = w[off >> 2], hi = w[(off >> 2) + 1];
            return (uint32_t)((((uint64_t)hi << 32) | (uint64_t)lo) >> (8 * (off & 3)));
        }
        uint32_t x;
        __builtin_memcpy(&x, (const uint8_t*)g + pos, 4);
        return x;
    }
    // eight bytes at `pos`: three aligned LDS words and two byte funnel shifts (the price of four bytes through a 64-bit shift)
    __device__ __forceinline__ void ld8(const uint32_t* w, int32_t w0, const PA_GLOBAL uint8_t* g, int32_t pos, uint32_t& lo, uint32_t& hi) const {
        const int32_t off = pos - w0;
        if (off >= 0 && off + 8 <= kBuildWin) {
            const uint32_t x0 = w[off >> 2], x1 = w[(off >> 2) + 1], x2 = w[(off >> 2) + 2];
            lo = __builtin_amdgcn_alignbyte(x1, x0, (uint32_t)(off & 3));
            hi = __builtin_amdgcn_alignbyte(x2, x1, (uint32_t)(off & 3));
            return;
        }
        uint64_t x;
        __builtin_memcpy(&x, (const uint8_t*)g + pos, 8);
        lo = (uint32_t)x;
        hi = (uint32_t)(x >> 32);
    }
    // stage a[a0 .. a0 + kBuildWin) and b[b0 ..) (clipped to the sequences; what lies beyond is never compared)
    __device__ __forceinline__ void stage(uint32_t* la, uint32_t* lb, int32_t na0, int32_t nb0) {
        const PA_GLOBAL uint8_t* a = (const PA_GLOBAL uint8_t*)jb.a;
        const PA_GLOBAL uint8_t* b = (const PA_GLOBAL uint8_t*)jb.b;
        a0 = na0 < 0 ? 0 : na0;
        b0 = nb0 < 0 ? 0 : nb0;
        for (int32_t t = lane; t < kBuildWin / 4 + 2; t += 64) {
            uint32_t x = 0, y = 0;
            const int32_t pa = a0 + 4 * t, pb = b0 + 4 * t;
            if (pa + 4 <= jb.n) __builtin_memcpy(&x, (const uint8_t*)a + pa, 4);
            else
                for (int32_t q = 0; q < 4; ++q)
                    if (pa + q < jb.n) x |= (uint32_t)a[pa + q] << (8 * q);
            if (pb + 4 <= jb.m) __builtin_memcpy(&y, (const uint8_t*)b + pb, 4);
            else
                for (int32_t q = 0; q < 4; ++q)
                    if (pb + q < jb.m) y |= (uint32_t)b[pb + q] << (8 * q);
            la[t] = x;
            lb[t] = y;
        }
        wa = la;
        wb = lb;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    __device__ __forceinline__ int32_t divk(int32_t x) const { return div_m ? (int32_t)(((uint64_t)(uint32_t)x * (uint64_t)div_m) >> 31) : x / jb.k; }
    __device__ __forceinline__ int32_t P(int32_t i) const {  // seeds.rs:34-71
        if (i < 0 || i > jb.n) return 0;
        const int32_t before = divk(i + jb.k - 1);
        return before < jb.nseeds ? jb.nseeds - before : 0;
    }
    // prepruning.rs:24-66 as one statement: the first column >= i on the diagonal (i, j) where the sequences differ or one of them ends;
    // true when that column is >= end_i (the reference may run past end_i in that case; nothing reads the column then)
    __device__ __forceinline__ bool extend(int32_t& i, int32_t j, int32_t end_i) const {
        const PA_GLOBAL uint8_t* a = (const PA_GLOBAL uint8_t*)jb.a;
        const PA_GLOBAL uint8_t* b = (const PA_GLOBAL uint8_t*)jb.b;
        while (i < end_i && i < jb.n && j < jb.m) {
            if (i + 8 <= jb.n && j + 8 <= jb.m) {  // eight characters per round (round 6: half the rounds of a typical run of matches)
                uint32_t xl, xh, yl, yh;
                ld8(wa, a0, a, i, xl, xh);
                ld8(wb, b0, b, j, yl, yh);
                const uint32_t dl = xl ^ yl, dh = xh ^ yh;
                if (dl | dh) {
                    i += dl ? (__builtin_ctz(dl) >> 3) : 4 + (__builtin_ctz(dh) >> 3);
                    return i >= end_i;
                }
                i += 8;
                j += 8;
            } else if (i + 4 <= jb.n && j + 4 <= jb.m) {
                const uint32_t x = ld4(wa, a0, a, i), y = ld4(wb, b0, b, j);
                const uint32_t d = x ^ y;
                if (d) {
                    const int32_t c = __builtin_ctz(d) >> 3;
                    i += c;
                    return i >= end_i;
                }
                i += 4;
                j += 4;
            } else {
                if (a[i] != b[j]) return i >= end_i;
                i += 1;
                j += 1;
            }
        }
        return i >= end_i;
    }
};

// Local pruning of ONE candidate by ONE lane, without the matches kept before it (prepruning.rs:95-203 with an empty next_match_per_diag):
// true = the search reaches the end of the next p seeds.  fr / nx: this lane's columns of the two LDS arrays, index (d + 1) * 64.
// The furthest-reaching columns live in LDS as 16-bit offsets from the candidate's start si (a search spans p seeds: at most (p + 1) k <= 480
// columns): half the LDS of 32-bit columns, and with it twelve instead of seven wavefronts to a CU (round 6) -- the kernel is bound by
// what a wavefront waits for, not by what it issues.  kNeg16 = "no column yet"; a diagonal inside [d0, d1) always has one.
constexpr int16_t kNeg16 = INT16_MIN;
// The search as a RESUMABLE object, one per lane (round 6): begin() runs level 0, level() one further level.  Phase D keeps every lane
// busy -- a lane whose search has ended takes the next candidate at once instead of idling until the deepest search of its round of 64 is
// through (the lanes used 45 % of the levels their rounds lasted: PA_BUILD_CLOCKS).  The decisions are those of the loop they replace.
struct AloneSearch {
    int32_t si, ei, ej, start_pot, end_i, pd, d0, d1, g;
    int16_t* fr;
    int16_t* nx;
#define FR(d) fr[((d) + 1) * 64]
#define NX(d) nx[((d) + 1) * 64]
    // +1: kept at once, -1: dropped at once (no level to run), 0: level() has to go on
    __device__ __forceinline__ int begin(const BuildCtx& cx, int32_t si_, int32_t sj, int16_t* fa, int16_t* fb) {
        const GcshBuildJob& jb = cx.jb;
        si = si_;
        ei = si + jb.k;
        ej = sj + jb.k;
        start_pot = cx.P(si);
        const int32_t seed_idx = cx.divk(si);
        int32_t last_seed = seed_idx + jb.p - 1;
        if (last_seed > jb.nseeds - 1) last_seed = jb.nseeds - 1;
        end_i = last_seed * jb.k + jb.k;
        pd = start_pot - cx.P(end_i);
        fr = fa;
        nx = fb;
        for (int32_t d = -1; d <= 2 * pd + 1; ++d) {
            FR(d) = kNeg16;
            NX(d) = kNeg16;
        }
        d0 = pd;
        d1 = pd + 1;
        int32_t i = ei;
        if (cx.extend(i, ej, end_i)) return 1;
        FR(pd) = (int16_t)(i - si);
        g = 1;
        return g < pd ? 0 : -1;
    }
    // level g (prepruning.rs:137-200): +1 kept, -1 dropped, 0 another level follows
    __device__ __forceinline__ int level(const BuildCtx& cx) {
        FR(d0 - 1) = kNeg16;
        FR(d1) = kNeg16;
        NX(d0 - 1) = kNeg16;
        NX(d1) = kNeg16;
        for (int32_t d = d0; d < d1; ++d) {
            const int32_t f = FR(d);
            if (NX(d - 1) < f) NX(d - 1) = (int16_t)f;
            if (NX(d) < f + 1) NX(d) = (int16_t)(f + 1);
            if (NX(d + 1) < f + 1) NX(d + 1) = (int16_t)(f + 1);
        }
        int16_t* t = fr;
        fr = nx;
        nx = t;
        d0 -= 1;
        d1 += 1;
        while (d0 < d1 && g + cx.P(si + FR(d0)) >= start_pot) d0 += 1;
        while (d0 < d1 && g + cx.P(si + FR(d1 - 1)) >= start_pot) d1 -= 1;
        if (d0 >= d1) return -1;
        for (int32_t d = d0; d < d1; ++d) {
            int32_t i = si + FR(d);
            const int32_t dd = ei - ej + (d - pd);
            if (cx.extend(i, i - dd, end_i)) return 1;
            FR(d) = (int16_t)(i - si);
        }
        g += 1;
        return g < pd ? 0 : -1;
    }
#undef FR
#undef NX
};

// The same search with the matches kept so far, spread over the wavefront: lane l owns diagonal l (0 .. 2 pd) of the search.
// ring_i / ring_d: the last kept matches (start column, diagonal), `nring` of them valid.  The reference's next_match_per_diag[dd] is
// the LATEST kept match of diagonal dd = the one with the smallest start column among those in reach.
__device__ __forceinline__ bool prune_with_kept(const BuildCtx& cx, int32_t si, int32_t sj, const int32_t* ring_i, const int32_t* ring_d, int32_t nring, int32_t* nm_lds) {
    const GcshBuildJob& jb = cx.jb;
    const int lane = cx.lane;
    const int32_t ei = si + jb.k, ej = sj + jb.k;
    const int32_t start_pot = cx.P(si);
    const int32_t seed_idx = cx.divk(si);
    int32_t last_seed = seed_idx + jb.p - 1;
    if (last_seed > jb.nseeds - 1) last_seed = jb.nseeds - 1;
    const int32_t end_i = last_seed * jb.k + jb.k;
    const int32_t pd = start_pot - cx.P(end_i);
    const int32_t dd = ei - ej + (lane - pd);  // this lane's diagonal
    // lane e of the ring drops its start column onto the diagonal it belongs to (one LDS atomic min each), lane l picks up its diagonal's
    nm_lds[lane] = INT32_MAX;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (lane < nring) {
        const int32_t rel = ring_d[lane] - (ei - ej) + pd;
        if (rel >= 0 && rel <= 2 * pd) atomicMin(nm_lds + rel, ring_i[lane]);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const int32_t nm = nm_lds[lane];
    int32_t fr = kBuildNeg, nx = kBuildNeg;  // this lane's entries of the reference's two arrays
    int32_t d0 = pd, d1 = pd + 1;
    {
        int32_t i = ei;
        bool hit = false;
        if (lane == pd) {
            hit = cx.extend(i, ej, end_i);
            fr = i;
            if (!hit && nm <= i) hit = true;  // prepruning.rs:134
        }
        if (__ballot(hit)) return true;
    }
    for (int32_t g = 1; g < pd; ++g) {
        if (lane == d0 - 1 || lane == d1) {
            fr = kBuildNeg;
            nx = kBuildNeg;
        }
        const bool in = lane >= d0 && lane < d1;
        const int32_t f = in ? fr : kBuildNeg;
        const int32_t up = __builtin_amdgcn_update_dpp(f, f, 0x130, 0xf, 0xf, false);  // wave_shl:1 = fr[d + 1] (lane 63: itself, never in range)
        const int32_t dn = __builtin_amdgcn_update_dpp(f, f, 0x138, 0xf, 0xf, false);  // wave_shr:1 = fr[d - 1]
        int32_t v = nx;
        if (lane + 1 >= d0 && lane + 1 < d1 && lane < 63 && v < up) v = up;           // next[d] = max(.., fr[d + 1])
        if (in && v < f + 1) v = f + 1;                                                // max(.., fr[d] + 1)
        if (lane >= 1 && lane - 1 >= d0 && lane - 1 < d1 && v < dn + 1) v = dn + 1;   // max(.., fr[d - 1] + 1)
        nx = fr;  // (swap: the old furthest-reaching columns are the next level's scratch, stale values and all -- as in the reference)
        fr = v;
        d0 -= 1;
        d1 += 1;
        const bool drop = g + cx.P(fr) >= start_pot;
        const uint64_t keep = __ballot(lane >= d0 && lane < d1 && !drop);
        if (!keep) return false;
        d0 = __builtin_ctzll(keep);
        d1 = 64 - __builtin_clzll(keep);
        bool hit = false;
        if (lane >= d0 && lane < d1) {
            int32_t i = fr;
            const int32_t old_i = i;
            hit = cx.extend(i, i - dd, end_i);
            fr = i;
            if (!hit && old_i <= nm && nm <= i) hit = true;  // prepruning.rs:156-158
        }
        if (__ballot(hit)) return true;
    }
    return false;
}

#ifdef PA_UNIT_GCSH_BUILD  // (the kernel is compiled in a translation unit of its own: csrc/apa2_units.hpp)
__global__ __launch_bounds__(64) void gcsh_build_kernel(const GcshBuildJob* __restrict__ jobs, int npairs, uint32_t* ticket) {
    // One pool, used phase by phase: the sieve of phases A / B (2048 words) lies over the search columns (2 x kBuildFr x 64 16-bit offsets)
    // and the head of the sequence windows, which phases D / E fill before they use them.  12.8 KB with the ring: twelve wavefronts to a CU.
    constexpr int kFrWords = kBuildFr * 64;          // 2 arrays x kBuildFr x 64 int16 = kFrWords 32-bit words
    constexpr int kWinWords = kBuildWin / 4 + 2;
    constexpr int kPoolWords = kFrWords + 2 * kWinWords > 2048 ? kFrWords + 2 * kWinWords : 2048;
    __shared__ __attribute__((aligned(16))) uint32_t lds_pool[kPoolWords];
    __shared__ int32_t ring_i[64], ring_d[64], nm_lds[64];
    int16_t* const lds_fr0 = reinterpret_cast<int16_t*>(lds_pool);
    int16_t* const lds_fr1 = lds_fr0 + kBuildFr * 64;
    uint32_t* const lds_wa = lds_pool + kFrWords;
    uint32_t* const lds_wb = lds_wa + kWinWords;
    const int lane = (int)(threadIdx.x & 63);
    for (;;) {
        uint32_t tk = atomicAdd(ticket, lane == 0 ? 1u : 0u);
        tk = rfl(tk);
        if (tk >= (uint32_t)npairs) break;
        const GcshBuildJob jb = jobs[tk];
        BuildCtx cx{jb, lane, 0u, 0, nullptr, nullptr, 0, 0};
        if (jb.k >= 1 && jb.k < 32 && jb.n < (1 << 26) - 64) cx.div_m = (uint32_t)((1ull << 31) / (uint64_t)jb.k) + 1u;
        cx.potn = cx.P(jb.n);
        const PA_GLOBAL uint8_t* a = (const PA_GLOBAL uint8_t*)jb.a;
        const PA_GLOBAL uint8_t* b = (const PA_GLOBAL uint8_t*)jb.b;
        PA_GLOBAL uint32_t* keys = (PA_GLOBAL uint32_t*)jb.keys;
        PA_GLOBAL int32_t* slot = (PA_GLOBAL int32_t*)jb.slot;
        PA_GLOBAL int32_t* next_same = (PA_GLOBAL int32_t*)jb.next_same;
        PA_GLOBAL int32_t* cnt = (PA_GLOBAL int32_t*)jb.cnt;
        PA_GLOBAL int32_t* fill = (PA_GLOBAL int32_t*)jb.fill;
        uint32_t status = kBuildOk;
        int32_t nmatch = 0;
        uint64_t tk0 = jb.clocks ? wall_clock64() : 0;
        auto lap = [&](int slot_) {
            if (!jb.clocks) return;
            const uint64_t t1 = wall_clock64();
            if (lane == 0) atomicAdd(jb.clocks + slot_, (unsigned long long)(t1 - tk0));
            tk0 = t1;
        };
        uint32_t n_alone = 0, n_search = 0;
        if (jb.nseeds > 0 && jb.m >= jb.k) {
            const uint32_t mask = (uint32_t)jb.tsize - 1u;
            // ---- A. seeds into the table ----
            for (int32_t t = lane; t < jb.tsize; t += 64) slot[t] = -1;
            for (int32_t s = lane; s < jb.nseeds; s += 64) {
                keys[s] = kmer_key(a + (size_t)s * jb.k, jb.k);
                cnt[s] = 0;
                fill[s] = 0;
            }
            if (lane == 0) cnt[jb.nseeds] = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            for (int32_t s = lane; s < jb.nseeds; s += 64) {
                const uint32_t key = keys[s];
                uint32_t h = key_hash(key, mask);
                for (;;) {
                    int32_t cur = __hip_atomic_load(slot + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (cur == -1) {
                        next_same[s] = -1;
                        int32_t expect = -1;
                        if (__hip_atomic_compare_exchange_strong(slot + h, &expect, s, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                        cur = expect;
                    }
                    if (keys[cur] == key) {  // the k-mer is there already: this seed becomes the head of its chain
                        next_same[s] = cur;
                        int32_t expect = cur;
                        if (__hip_atomic_compare_exchange_strong(slot + h, &expect, s, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                        continue;  // (somebody else got in first: look at the entry again)
                    }
                    h = (h + 1u) & mask;
                }
            }
            // In front of the table: one bit per hashed seed key (round 5; csrc/gcsh.hpp does the same on the host).  Almost every position
            // of b matches no seed at all, and half of those still find their table slot occupied (two dependent loads from global
            // memory): 64 K bits in the LDS array phase D will use later stop seven of eight positions with one LDS read.
            uint32_t* const sieve = lds_pool;  // 2048 words
            for (int32_t t = lane; t < 2048; t += 64) sieve[t] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            for (int32_t s = lane; s < jb.nseeds; s += 64) {
                const uint32_t hb = (keys[s] * 0x85EBCA6Bu) >> 16;
                atomicOr(sieve + (hb >> 5), 1u << (hb & 31u));
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            lap(0);
            // ---- B. every k-mer of b: candidates in rows ascending, seeds DESCENDING within a row (read backwards that is the order the
            //      reference pushes them in: rows descending, seeds ascending -- exact.rs:40-69) ----
            const int32_t ttx = jb.n - jb.m - cx.potn, tty = jb.m - jb.n - cx.potn;
            int32_t ncand = 0;
            PA_GLOBAL int32_t* tmp_s = (PA_GLOBAL int32_t*)jb.tmp_s;
            PA_GLOBAL int32_t* tmp_j = (PA_GLOBAL int32_t*)jb.tmp_j;
            for (int32_t base = 0; base <= jb.m - jb.k; base += 64) {
                const int32_t j = base + lane;
                const bool valid = j <= jb.m - jb.k;
                int32_t head = -1;
                if (valid) {
                    const uint32_t key = kmer_key(b + j, jb.k);
                    uint32_t h = key_hash(key, mask);
                    const uint32_t hb = (key * 0x85EBCA6Bu) >> 16;
                    for (; (sieve[hb >> 5] >> (hb & 31u)) & 1u;) {
                        const int32_t cur = slot[h];
                        if (cur < 0) break;
                        if (keys[cur] == key) {
                            head = cur;
                            break;
                        }
                        h = (h + 1u) & mask;
                    }
                }
                auto ok_at = [&](int32_t s) -> bool {  // matches.rs:205-215: T(start) <= T(target)
                    const int32_t i = s * jb.k, ps = cx.P(i);
                    return i - j - ps <= ttx && j - i - ps <= tty;
                };
                int32_t mine = 0;
                for (int32_t s = head; s >= 0; s = next_same[s]) mine += ok_at(s) ? 1 : 0;
                const int32_t incl = wave_scan_add(mine);
                const int32_t total = __builtin_amdgcn_readlane(incl, 63);
                if (ncand + total > jb.cap) {
                    status = kBuildOverflow;
                    break;
                }
                if (mine) {
                    const int32_t at = ncand + incl - mine;
                    int32_t w = 0;
                    for (int32_t s = head; s >= 0; s = next_same[s])
                        if (ok_at(s)) {
                            // insertion by seed, descending (chains are in no particular order; they hold one seed almost always)
                            int32_t q = w;
                            while (q > 0 && tmp_s[at + q - 1] < s) {
                                tmp_s[at + q] = tmp_s[at + q - 1];
                                q -= 1;
                            }
                            tmp_s[at + q] = s;
                            tmp_j[at + w] = j;
                            w += 1;
                            atomicAdd((int32_t*)(cnt + s), 1);
                        }
                }
                ncand += total;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            lap(1);
            if (status == kBuildOk) {
                // ---- C. by-start order: exclusive prefix of the per-seed counts, scatter, rows ascending within a seed ----
                int32_t carry = 0;
                for (int32_t base = 0; base <= jb.nseeds; base += 64) {
                    const int32_t s = base + lane;
                    const int32_t v = s < jb.nseeds ? cnt[s] : 0;
                    const int32_t incl = wave_scan_add(v);
                    if (s <= jb.nseeds) cnt[s] = carry + incl - v;
                    carry += __builtin_amdgcn_readlane(incl, 63);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                PA_GLOBAL int32_t* gpos = (PA_GLOBAL int32_t*)jb.gpos;
                PA_GLOBAL int32_t* cj = (PA_GLOBAL int32_t*)jb.cj;
                // Rows ascend in tmp, so a seed's candidates meet this loop in row order: the slot inside the seed's range is the number
                // of its candidates seen so far -- `fill` from the rounds before plus the earlier lanes of this round with the same seed
                // (almost never any), counted, not raced for.
                for (int32_t base = 0; base < ncand; base += 64) {
                    const int32_t t = base + lane;
                    int32_t s = -1, j = 0;
                    if (t < ncand) {
                        s = tmp_s[t];
                        j = tmp_j[t];
                    }
                    int32_t before = 0;
                    bool last = true;
                    for (int l = 0; l < 64; ++l) {
                        const int32_t sl = __builtin_amdgcn_readlane(s, l);
                        if (sl == s && l < lane) before += 1;
                        if (sl == s && l > lane) last = false;
                    }
                    int32_t f0 = 0;
                    if (t < ncand) {
                        f0 = fill[s];
                        const int32_t q = cnt[s] + f0 + before;
                        gpos[t] = q;
                        cj[q] = j;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (every lane has read the mark before anyone moves it)
                    if (t < ncand && last) fill[s] = f0 + before + 1;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
                lap(2);
                // ---- D. local pruning alone, one lane per candidate ----
                PA_GLOBAL uint8_t* flag = (PA_GLOBAL uint8_t*)jb.flag;
                if (jb.p == 0) {
                    for (int32_t t = lane; t < ncand; t += 64) flag[t] = 1;
                } else {
                    // ---- D0. chains (round 5).  Matches along the alignment come in runs on one diagonal: candidate (seed s, row j) is followed
                    // by (s + 1, j + k).  In the reference's push order the successor is decided first, and if it is kept the candidate meets it at
                    // its own end -- next_match_per_diag[diagonal] == i + k <= the column its first extension reaches (prepruning.rs:134) -- and is
                    // kept at once.  So a candidate WITH a successor candidate is not searched here at all (flag 2): phase E keeps it when the
                    // successor was kept and only otherwise runs its search.  The expensive search "alone" (through all p seeds: no kept match
                    // ends it early) is left for the run ends and the stray candidates -- about 45 % of them at 5 % divergence.
                    // (succ / todo live in the mi / mj arrays, which phase F fills only afterwards)
                    PA_GLOBAL int32_t* succ = (PA_GLOBAL int32_t*)jb.mi;
                    PA_GLOBAL int32_t* todo = (PA_GLOBAL int32_t*)jb.mj;
                    int32_t ntodo = 0;
                    for (int32_t base = 0; base < ncand; base += 64) {
                        const int32_t t = base + lane;
                        int32_t sc = -1;
                        if (t < ncand) {
                            const int32_t s = tmp_s[t], jr = tmp_j[t] + jb.k;
                            if (s + 1 < jb.nseeds && jr <= jb.m - jb.k) {
                                int32_t lo = t + 1, hi = ncand;  // rows ascend with the index: first index whose row is >= jr
                                while (lo < hi) {
                                    const int32_t mid = (lo + hi) >> 1;
                                    if (tmp_j[mid] < jr) lo = mid + 1;
                                    else hi = mid;
                                }
                                for (int32_t x = lo; x < ncand && tmp_j[x] == jr; ++x)
                                    if (tmp_s[x] == s + 1) {
                                        sc = x;
                                        break;
                                    }
                            }
                            succ[t] = sc;
                            flag[t] = sc >= 0 ? 2 : 0;
                        }
                        const uint64_t need = __ballot(t < ncand && sc < 0);
                        if (t < ncand && sc < 0) todo[ntodo + __builtin_popcountll(need & ((1ull << lane) - 1ull))] = t;
                        ntodo += __builtin_popcountll(need);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    // ---- D. local pruning alone, one lane per candidate that has no successor; a lane that is through takes the next one ----
                    {
                        AloneSearch S{};
                        bool active = false;
                        int32_t t = -1, next = 0, staged = -64;  // next: first candidate of `todo` not handed out yet; staged: `next` at the last staging
                        uint64_t lv_sum = 0, lv_rounds = 0;
                        for (;;) {
                            for (;;) {  // hand out candidates while there are idle lanes and candidates
                                const uint64_t idle = __ballot(!active);
                                const int32_t remaining = ntodo - next;
                                if (idle == 0 || remaining <= 0) break;
                                const int32_t nidle = __builtin_popcountll(idle), give = nidle < remaining ? nidle : remaining;
                                if (next - staged >= 32) {
                                    // the windows follow the candidates (rows ascend along `todo`): b from the row of the oldest search still
                                    // running or the first new candidate, a along the new candidates' middle diagonal; a search that leaves
                                    // them reads global memory
                                    const int32_t tf = todo[next], tm = todo[next + give / 2];
                                    int32_t lowj = active ? S.ej - jb.k : tmp_j[tf];
                                    for (int o = 32; o > 0; o >>= 1) lowj = min(lowj, __shfl_xor(lowj, o));
                                    const int32_t firstj = tmp_j[tf];
                                    if (lowj < firstj - kBuildWin / 2) lowj = firstj - kBuildWin / 2;  // (a straggler far behind does not hold the window back)
                                    const int32_t nb0 = (lowj + jb.k - 16) & ~3;
                                    cx.stage(lds_wa, lds_wb, (nb0 + tmp_s[tm] * jb.k - tmp_j[tm]) & ~3, nb0);
                                    staged = next;
                                }
                                const int32_t rank = __builtin_popcountll(idle & ((1ull << lane) - 1ull));
                                if (!active && rank < give) {
                                    t = todo[next + rank];
                                    const int r = S.begin(cx, tmp_s[t] * jb.k, tmp_j[t], lds_fr0 + lane, lds_fr1 + lane);
                                    if (r != 0) flag[t] = r > 0 ? 1 : 0;
                                    else active = true;
                                }
                                next += give;
                            }
                            const uint64_t running = __ballot(active);
                            if (running == 0) break;
                            if (active) {
                                const int r = S.level(cx);
                                if (r != 0) {
                                    flag[t] = r > 0 ? 1 : 0;
                                    active = false;
                                }
                            }
                            lv_sum += (uint64_t)__builtin_popcountll(running);
                            lv_rounds += 64;
                        }
                        if (jb.clocks && lane == 0) {  // diagnostics: search levels run, and lane slots the loop's iterations offered
                            atomicAdd(jb.clocks + 9, (unsigned long long)lv_sum);
                            atomicAdd(jb.clocks + 10, (unsigned long long)lv_rounds);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    lap(3);
                    // ---- E. the reference's order (tmp read backwards), the kept matches of the last rows in a ring ----
                    int32_t nkept = 0;
                    const int32_t reach = (jb.p + 2) * jb.k + 64 + jb.p;  // rows a kept match can still be met from
                    for (int32_t top = ncand - 1; top >= 0 && status == kBuildOk; top -= 64) {
                        const int32_t t = top - lane;
                        int32_t s = 0, j = 0, f = 0, sc = -1;
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (the final flags of the batches before)
                        if (t >= 0) {
                            s = tmp_s[t];
                            j = tmp_j[t];
                            f = flag[t];
                            sc = succ[t];
                        }
                        const int cntl = top + 1 < 64 ? top + 1 : 64;
                        {  // (lane cntl - 1 holds the batch's lowest row)
                            const int32_t nb0 = (__builtin_amdgcn_readlane(j, cntl - 1) + jb.k - 16) & ~3;
                            cx.stage(lds_wa, lds_wb, (nb0 + (__builtin_amdgcn_readlane(s, cntl / 2) * jb.k - __builtin_amdgcn_readlane(j, cntl / 2))) & ~3, nb0);
                        }
                        uint64_t kept_mask = 0;  // lane l of this batch was kept
                        for (int l = 0; l < cntl; ++l) {
                            const int32_t si = __builtin_amdgcn_readlane(s, l) * jb.k, sj = __builtin_amdgcn_readlane(j, l);
                            const int32_t fl = __builtin_amdgcn_readlane(f, l);
                            bool keep = fl == 1;
                            if (fl == 2) {  // a run on one diagonal: kept with its successor (D0)
                                const int32_t x = __builtin_amdgcn_readlane(sc, l);  // (x > top - l: decided earlier)
                                if (x <= top) keep = ((kept_mask >> (top - x)) & 1ull) != 0;
                                else keep = rfl((uint32_t)flag[x]) == 1u;
                            }
                            n_alone += keep ? 1u : 0u;
                            if (!keep) {
                                n_search += 1;
                                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                                keep = prune_with_kept(cx, si, sj, ring_i, ring_d, nkept < 64 ? nkept : 64, nm_lds);
                            }
                            if (keep) {  // MatchBuilder::push: next_match_per_diag[i - j] = i (matches.rs:229-244)
                                kept_mask |= 1ull << l;
                                const int e = nkept & 63;
                                if (nkept >= 64) {
                                    const int32_t oj = ring_i[e] - ring_d[e];
                                    if (oj - sj <= reach) status = kBuildRing;  // the match falling out of the ring could still be met
                                }
                                if (lane == 0) {
                                    ring_i[e] = si;
                                    ring_d[e] = si - sj;
                                }
                                nkept += 1;
                            }
                        }
                        if (t >= 0) flag[t] = (uint8_t)((kept_mask >> lane) & 1ull);  // (final: phase F and the runs that end in later batches read it)
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                lap(4);
                if (jb.clocks && lane == 0) {
                    atomicAdd(jb.clocks + 6, (unsigned long long)ncand);
                    atomicAdd(jb.clocks + 7, (unsigned long long)n_alone);
                    atomicAdd(jb.clocks + 8, (unsigned long long)n_search);
                }
                if (status == kBuildOk) {
                    // ---- F. kept matches by start, the windows of the seeds ----
                    PA_GLOBAL uint8_t* keptg = (PA_GLOBAL uint8_t*)jb.keptg;
                    for (int32_t t = lane; t < ncand; t += 64) keptg[gpos[t]] = flag[t];
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    PA_GLOBAL int32_t* mi = (PA_GLOBAL int32_t*)jb.mi;
                    PA_GLOBAL int32_t* mj = (PA_GLOBAL int32_t*)jb.mj;
                    int32_t carry2 = 0;
                    for (int32_t base = 0; base < jb.nseeds; base += 64) {
                        const int32_t s = base + lane;
                        int32_t g0 = 0, g1 = 0, kc = 0;
                        if (s < jb.nseeds) {
                            g0 = cnt[s];
                            g1 = cnt[s + 1];
                            for (int32_t q = g0; q < g1; ++q) kc += keptg[q];
                        }
                        const int32_t incl = wave_scan_add(kc);
                        if (s < jb.nseeds) {
                            int32_t o = carry2 + incl - kc;
                            PA_GLOBAL pa_i32x4_b* wp = (PA_GLOBAL pa_i32x4_b*)jb.win0;
                            const pa_i32x4_b wv = {o, o + kc, -1, 0};
                            wp[s] = wv;
                            for (int32_t q = g0; q < g1; ++q)
                                if (keptg[q]) {
                                    mi[o] = s * jb.k;
                                    mj[o] = cj[q];
                                    o += 1;
                                }
                        }
                        carry2 += __builtin_amdgcn_readlane(incl, 63);
                    }
                    nmatch = carry2;
                }
            }
        } else {
            const pa_i32x4_b wv = {0, 0, -1, 0};
            for (int32_t s = lane; s < jb.nseeds; s += 64) ((PA_GLOBAL pa_i32x4_b*)jb.win0)[s] = wv;
        }
        lap(5);
        if (status != kBuildOk) nmatch = -1;  // (the band search hands such a pair back to the host engine)
        if (lane == 0) {
            *(PA_GLOBAL int32_t*)jb.nmatch_out = nmatch;
            *(PA_GLOBAL uint32_t*)jb.status = status;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}
#endif  // PA_UNIT_GCSH_BUILD

}  // namespace apa2
}  // namespace pa
