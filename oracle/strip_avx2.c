/*
 * oracle/strip_avx2.c -- TEST INFRASTRUCTURE / CPU BASELINE (see pa_oracle.h).
 *
 * Port of the reference's SIMD schedule `simd::compute::<2,(u64,u64),4>`:
 * 8 word-rows per strip, lane k owns word-row rev(k)=7-k and runs k columns ahead (anti-diagonal
 * skew), two independent 4x64-bit vectors for ILP, scalar triangles at both ends.
 * Follows pa-bitpacking/src/simd.rs:98-315 and myers.rs:61-91.  This is what bench.py times as
 * `cpu_baseline` (kind "port"); compile with -O3 -march=native (256-bit vectors, as the reference's
 * Simd<u64,4> under -C target-cpu=native).
 */
#include "pa_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef uint64_t u64x4 __attribute__((vector_size(32)));
typedef uint64_t u64x4u __attribute__((vector_size(32), aligned(8)));

static inline u64x4 splat(uint64_t x) { return (u64x4){x, x, x, x}; }

/* myers.rs:61-91 */
static inline void block_simd(u64x4* hp0, u64x4* hm0, u64x4* vp, u64x4* vm, u64x4 eq) {
    u64x4 vx = eq | *vm;
    eq = eq | *hm0;
    u64x4 hx = (((eq & *vp) + *vp) ^ *vp) | eq;
    u64x4 hp = *vm | ~(hx | *vp);
    u64x4 hm = *vp & hx;
    u64x4 hpw = hp >> 63;
    u64x4 hmw = hm >> 63;
    hp = (hp << 1) | *hp0;
    hm = (hm << 1) | *hm0;
    *hp0 = hpw;
    *hm0 = hmw;
    *vp = hm | ~(vx | hp);
    *vm = hp & vx;
}

/* rotate_left over the 8 lanes (lane k <- lane k+1, lane 7 <- carry), returns old lane 0. simd.rs:76-87 */
static inline uint64_t rotate8(u64x4* lo, u64x4* hi, uint64_t carry) {
    uint64_t out = (*lo)[0];
    u64x4 l = *lo, h = *hi;
    const u64x4 rot = {1, 2, 3, 0};
    l[0] = h[0];
    h[0] = carry;
    *lo = __builtin_shuffle(l, rot);
    *hi = __builtin_shuffle(h, rot);
    return out;
}

/* compute_block_of_rows::<2,H,4>, simd.rs:228-315.  ap0/ap1 are the unzipped bits of a. */
static void strip8(const pa_bits_t* a, const uint64_t* ap0, const uint64_t* ap1, size_t n,
                   const pa_bits_t* cbs, pa_h_t* h, pa_v_t* v) {
    /* top-left triangle, simd.rs:243-247 */
    for (size_t j = 0; j < 8; ++j)
        for (size_t i = 0; i < 8 - j; ++i) pa_or_compute_block(&h[i], &v[j], a[i], cbs[j]);

    u64x4 b0[2], b1[2], ph[2], mh[2], pv[2], mv[2];
    for (int k = 0; k < 8; ++k) {
        int r = 7 - k;
        b0[k / 4][k % 4] = cbs[r].b0;
        b1[k / 4][k % 4] = cbs[r].b1;
        ph[k / 4][k % 4] = h[k].p;
        mh[k / 4][k % 4] = h[k].m;
        pv[k / 4][k % 4] = v[r].p;
        mv[k / 4][k % 4] = v[r].m;
    }
    /* steady state, simd.rs:265-293 */
    for (size_t i = 0; i + 8 < n; ++i) {
        u64x4 a0l = *(const u64x4u*)(ap0 + i + 1), a0h = *(const u64x4u*)(ap0 + i + 5);
        u64x4 a1l = *(const u64x4u*)(ap1 + i + 1), a1h = *(const u64x4u*)(ap1 + i + 5);
        u64x4 eq0 = (a0l ^ b0[0]) & (a1l ^ b1[0]);
        u64x4 eq1 = (a0h ^ b0[1]) & (a1h ^ b1[1]);
        uint64_t pc = rotate8(&ph[0], &ph[1], h[i + 8].p);
        uint64_t mc = rotate8(&mh[0], &mh[1], h[i + 8].m);
        h[i].p = pc;
        h[i].m = mc;
        block_simd(&ph[0], &mh[0], &pv[0], &mv[0], eq0);
        block_simd(&ph[1], &mh[1], &pv[1], &mv[1], eq1);
    }
    for (int k = 0; k < 8; ++k) { /* write back, simd.rs:296-308 */
        h[n - 8 + k].p = ph[k / 4][k % 4];
        h[n - 8 + k].m = mh[k / 4][k % 4];
        v[7 - k].p = pv[k / 4][k % 4];
        v[7 - k].m = mv[k / 4][k % 4];
    }
    /* bottom-right triangle, simd.rs:310-314 */
    for (size_t j = 0; j < 8; ++j)
        for (size_t i = n - j; i < n; ++i) pa_or_compute_block(&h[i], &v[j], a[i], cbs[j]);
}

int32_t pa_or_strip_compute_avx2(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                                 pa_h_t* h, pa_v_t* v, int exact_end) {
    /* Small shapes take the reference's scalar/narrow paths (simd.rs:112-134): not timed, use the
     * schedule-independent restatement. */
    if (n < 16 || w == 1) return pa_or_simd_compute(a, n, b, w, h, v, exact_end, 2);

    uint64_t* ap0 = (uint64_t*)malloc(2 * n * sizeof(uint64_t)); /* unzip, simd.rs:137-138 */
    uint64_t* ap1 = ap0 + n;
    for (size_t i = 0; i < n; ++i) { ap0[i] = a[i].b0; ap1[i] = a[i].b1; }

    size_t j = 0;
    for (; j + 8 <= w; j += 8) strip8(a, ap0, ap1, n, b + j, h, v + j);
    size_t rem = w - j;
    int32_t ret;
    if (rem == 0) {
        ret = 0;
        for (size_t i = 0; i < n; ++i) ret += (int32_t)h[i].p - (int32_t)h[i].m;
    } else if (!exact_end && rem >= 5) {
        /* pad to 8 rows with Bits(0,0), V(0,0) and subtract the pad rows' right edge, simd.rs:211-218 */
        pa_bits_t bt[8];
        pa_v_t vt[8];
        for (size_t k = 0; k < 8; ++k) {
            bt[k] = k < rem ? b[j + k] : (pa_bits_t){0, 0};
            vt[k] = k < rem ? v[j + k] : (pa_v_t){0, 0};
        }
        strip8(a, ap0, ap1, n, bt, h, vt);
        memcpy(v + j, vt, rem * sizeof(pa_v_t));
        ret = 0;
        for (size_t i = 0; i < n; ++i) ret += (int32_t)h[i].p - (int32_t)h[i].m;
        for (size_t k = rem; k < 8; ++k) ret -= pa_or_v_value(vt[k]);
    } else {
        /* 1..4 remaining rows (or exact mode): narrower reference paths; values are schedule
         * independent, so finish with the restatement on the remaining rows. */
        ret = pa_or_simd_compute(a, n, b + j, rem, h, v + j, exact_end, 1);
    }
    free(ap0);
    return ret;
}
