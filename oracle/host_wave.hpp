/*
 * oracle/host_wave.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A 64-lane wavefront emulated with arrays, as the wave policy of astar-pairwise-aligner_amd/csrc/sweep_wave.hpp: the same
 * wave program the GPU runs (band decisions, hand-off protocol, block-boundary bookkeeping) executes here on host threads,
 * one thread per wavefront, so that the protocol can be tested without a GPU.  The Myers step is restated from
 * pa-bitpacking/src/myers.rs:27-55 on 32-row subwords (K = 1 of strip_kernel.hpp).
 */
#pragma once
#include <atomic>
#include <cstdint>
#include <thread>

namespace pa_host_wave {

struct HVec {
    uint32_t a[64];
};
struct HMask {
    bool a[64];
};
#define HV_BIN(op)                                                   \
    inline HVec operator op(const HVec& x, const HVec& y) {          \
        HVec r;                                                      \
        for (int i = 0; i < 64; ++i) r.a[i] = x.a[i] op y.a[i];      \
        return r;                                                    \
    }                                                                \
    inline HVec operator op(const HVec& x, uint32_t y) {             \
        HVec r;                                                      \
        for (int i = 0; i < 64; ++i) r.a[i] = x.a[i] op y;           \
        return r;                                                    \
    }
HV_BIN(&)
HV_BIN(|)
HV_BIN(^)
HV_BIN(+)
HV_BIN(-)
HV_BIN(*)
#undef HV_BIN

struct HostWave {
    using vec = HVec;
    using mask = HMask;

    static vec splat(uint32_t x) {
        vec r;
        for (int i = 0; i < 64; ++i) r.a[i] = x;
        return r;
    }
    static vec lane_ids() {
        vec r;
        for (int i = 0; i < 64; ++i) r.a[i] = (uint32_t)i;
        return r;
    }
    static vec select(const mask& m, const vec& x, const vec& y) {
        vec r;
        for (int i = 0; i < 64; ++i) r.a[i] = m.a[i] ? x.a[i] : y.a[i];
        return r;
    }
#define HM_CMP(name, T, expr)                              \
    static mask name(const vec& x, T y) {                  \
        mask r;                                            \
        for (int i = 0; i < 64; ++i) {                     \
            const T xi = (T)x.a[i];                        \
            r.a[i] = (expr);                               \
        }                                                  \
        return r;                                          \
    }
    HM_CMP(eq_u, uint32_t, xi == y)
    HM_CMP(ne_u, uint32_t, xi != y)
    HM_CMP(le_u, uint32_t, xi <= y)
    HM_CMP(ge_i, int32_t, xi >= y)
    HM_CMP(lt_i, int32_t, xi < y)
    HM_CMP(gt_i, int32_t, xi > y)
    HM_CMP(le_i, int32_t, xi <= y)
#undef HM_CMP
    static mask and_m(const mask& x, const mask& y) {
        mask r;
        for (int i = 0; i < 64; ++i) r.a[i] = x.a[i] && y.a[i];
        return r;
    }
    static vec shr_v(const vec& x, const vec& s) {
        vec r;
        for (int i = 0; i < 64; ++i) r.a[i] = x.a[i] >> (s.a[i] & 31);
        return r;
    }
    static vec shl_v(const vec& x, const vec& s) {
        vec r;
        for (int i = 0; i < 64; ++i) r.a[i] = x.a[i] << (s.a[i] & 31);
        return r;
    }
    static uint32_t popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
    static vec popc_v(const vec& x) {
        vec r;
        for (int i = 0; i < 64; ++i) r.a[i] = (uint32_t)__builtin_popcount(x.a[i]);
        return r;
    }
    static uint32_t readlane(const vec& x, int i) { return x.a[i & 63]; }
    static int32_t readlane_i(const vec& x, int i) { return (int32_t)x.a[i & 63]; }
    static uint32_t reduce_add(const vec& x) {
        uint32_t s = 0;
        for (int i = 0; i < 64; ++i) s += x.a[i];
        return s;
    }
    static vec prefix_excl(const vec& x) {
        vec r;
        uint32_t s = 0;
        for (int i = 0; i < 64; ++i) {
            r.a[i] = s;
            s += x.a[i];
        }
        return r;
    }

    // ---- memory (host threads: sequentially consistent atomics are more than the device's relaxed agent scope) ----
    static uint32_t load_u32(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
    static uint64_t load_u64(const uint64_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
    static void store_u64(uint64_t* p, uint64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
    static bool cas_u32(uint32_t* p, uint32_t expect, uint32_t v) {
        return __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
    }
    static void add_u64(uint64_t* p, uint64_t v) { __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
    static void nap(uint32_t) { std::this_thread::yield(); }
    static uint64_t clock() { return 0; }
    static void stretch_marker() {}
    static void drain_stores() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
    static uint32_t ticket(uint32_t* p) { return __atomic_fetch_add(p, 1u, __ATOMIC_ACQ_REL); }

    // lane l (< n) reads the 8-byte word p[l]
    static void load_words(const uint64_t* p, int n, vec& lo, vec& hi) {
        for (int i = 0; i < 64; ++i) {
            lo.a[i] = hi.a[i] = 0;
            if (i < n) {
                const uint64_t w = __atomic_load_n(p + i, __ATOMIC_ACQUIRE);
                lo.a[i] = (uint32_t)w;
                hi.a[i] = (uint32_t)(w >> 32);
            }
        }
    }
    static void load_i32s(const int32_t* p, int n, vec& v) {
        for (int i = 0; i < 64; ++i) v.a[i] = i < n ? (uint32_t)p[i] : 0u;
    }
    static void load_codes2(const uint32_t* codes, int32_t q, uint32_t& lo, uint32_t& hi) {
        lo = codes[2 * (int64_t)q];
        hi = codes[2 * (int64_t)q + 1];
    }
    static void load_profile(const uint32_t* prof, uint32_t word0, int32_t wtot, const vec& lane, vec& nb0, vec& nb1) {
        for (int i = 0; i < 64; ++i) {
            const uint32_t w = word0 + lane.a[i] / 2, half = lane.a[i] & 1;
            nb0.a[i] = nb1.a[i] = 0;
            if ((int32_t)w < wtot) {
                nb0.a[i] = prof[(size_t)w * 4 + half];
                nb1.a[i] = prof[(size_t)w * 4 + 2 + half];
            }
        }
    }
    // V word w = (p:u64, m:u64) at col[2w], col[2w+1]
    static void load_v_words(const uint64_t* col, const vec& widx, const mask& inr, vec& plo, vec& phi, vec& mlo, vec& mhi) {
        for (int i = 0; i < 64; ++i) {
            plo.a[i] = phi.a[i] = mlo.a[i] = mhi.a[i] = 0;
            if (!inr.a[i]) continue;
            const uint64_t p = __atomic_load_n(col + 2 * (int64_t)widx.a[i], __ATOMIC_ACQUIRE);
            const uint64_t m = __atomic_load_n(col + 2 * (int64_t)widx.a[i] + 1, __ATOMIC_ACQUIRE);
            plo.a[i] = (uint32_t)p;
            phi.a[i] = (uint32_t)(p >> 32);
            mlo.a[i] = (uint32_t)m;
            mhi.a[i] = (uint32_t)(m >> 32);
        }
    }
    // lane l holds half (l & 1) of word word0 + l / 2
    static void store_v_halves(uint64_t* col, uint32_t word0, const vec& lane, const mask& act, const vec& sp, const vec& sm) {
        for (int i = 0; i < 64; i += 2) {
            if (!act.a[i]) continue;  // both lanes of a word are active together (ranges are rounded to 64)
            const int64_t w = (int64_t)word0 + lane.a[i] / 2;
            const uint64_t p = (uint64_t)sp.a[i] | ((uint64_t)sp.a[i + 1] << 32);
            const uint64_t m = (uint64_t)sm.a[i] | ((uint64_t)sm.a[i + 1] << 32);
            __atomic_store_n(col + 2 * w, p, __ATOMIC_RELEASE);
            __atomic_store_n(col + 2 * w + 1, m, __ATOMIC_RELEASE);
        }
    }

    // One Myers step for all 64 lanes (myers.rs:27-55 on 32-row subwords; the packed pipeline register of
    // strip_kernel.hpp: bit31 = h.p, bit30 = h.m, bits 1:0 = base code).
    template <bool FORCE>
    static void myers(uint32_t s_x, vec& X, vec& vp, vec& vm, const vec& nb0, const vec& nb1, vec& acc, const vec& andm, const vec& orm) {
        vec Xo;
        for (int l = 0; l < 64; ++l) {
            acc.a[l] = (acc.a[l] << 2) | (X.a[l] >> 30);
            uint32_t Xin = l == 0 ? s_x : X.a[l - 1];
            if (FORCE) Xin = (Xin & andm.a[l]) | orm.a[l];
            const uint32_t a0 = (Xin & 1u) ? 0xFFFFFFFFu : 0u, a1 = (Xin & 2u) ? 0xFFFFFFFFu : 0u;
            const uint32_t hp0 = Xin >> 31, hm0 = (Xin >> 30) & 1u;
            uint32_t eq = (a0 ^ nb0.a[l]) & (a1 ^ nb1.a[l]);
            const uint32_t p = vp.a[l], m = vm.a[l];
            const uint32_t vx = eq | m;
            eq |= hm0;
            const uint32_t hx = (((eq & p) + p) ^ p) | eq;
            const uint32_t hp = m | ~(hx | p), hm = p & hx;
            Xo.a[l] = (hp & 0x80000000u) | ((hm >> 1) & 0x40000000u) | (Xin & 0x3FFFFFFFu);
            const uint32_t hp2 = (hp << 1) | hp0, hm2 = (hm << 1) | hm0;
            vp.a[l] = hm2 | ~(vx | hp2);
            vm.a[l] = hp2 & vx;
        }
        X = Xo;
    }
    template <bool FORCE>
    static void chunk(const vec& XS, vec& X, vec& vp, vec& vm, const vec& nb0, const vec& nb1, vec& acc_lo, vec& acc_hi, const vec& andm,
                      const vec& orm) {
        for (int j = 0; j < 32; ++j) myers<FORCE>(XS.a[j], X, vp, vm, nb0, nb1, j < 16 ? acc_lo : acc_hi, andm, orm);
    }
    // steps [j0, j1) of a chunk in which lane cl0 + j leaves its block at step j: snapshot of its V, pending +1 forcing,
    // V::one() if it was below the band.  (The device has several specialised copies; they all mean this.)
    static void chunk_cross_range(const vec& XS, vec& X, vec& vp, vec& vm, const vec& nb0, const vec& nb1, vec& acc_lo, vec& acc_hi, vec& andm,
                                  vec& orm, const vec& lane, int32_t cl0, vec& snap_p, vec& snap_m, const vec& resetm, const vec& fpend, int32_t j0,
                                  int32_t j1) {
        for (int j = j0; j < j1; ++j) {
            const int32_t cl = cl0 + j;
            if (cl >= 0 && cl < 64) {
                snap_p.a[cl] = vp.a[cl];
                snap_m.a[cl] = vm.a[cl];
                if (fpend.a[cl]) {
                    andm.a[cl] = 3u;
                    orm.a[cl] = 0x80000000u;
                }
                if (resetm.a[cl]) {
                    vp.a[cl] = 0xFFFFFFFFu;
                    vm.a[cl] = 0;
                }
            }
            myers<true>(XS.a[j], X, vp, vm, nb0, nb1, j < 16 ? acc_lo : acc_hi, andm, orm);
        }
        (void)lane;
    }
    template <bool FORCE, bool EXTRA>
    static void chunk_cross(const vec& XS, vec& X, vec& vp, vec& vm, const vec& nb0, const vec& nb1, vec& acc_lo, vec& acc_hi, vec& andm,
                            vec& orm, const vec& lane, int32_t cl0, vec& snap_p, vec& snap_m, const vec& resetm, const vec& fpend) {
        chunk_cross_range(XS, X, vp, vm, nb0, nb1, acc_lo, acc_hi, andm, orm, lane, cl0, snap_p, snap_m, resetm, fpend, 0, 32);
    }
    static bool any(const vec& x) {
        for (int i = 0; i < 64; ++i)
            if (x.a[i]) return true;
        return false;
    }
};

}  // namespace pa_host_wave
