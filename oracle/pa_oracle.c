/*
 * oracle/pa_oracle.c -- TEST INFRASTRUCTURE (see pa_oracle.h).  CPU restatement of pa-bitpacking.
 * Reference paths are relative to /root/reference/.
 */
#include "pa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

/* ---- encoding.rs:21-38 ------------------------------------------------------------------- */
int32_t pa_or_v_value(pa_v_t v) {
    return (int32_t)__builtin_popcountll(v.p) - (int32_t)__builtin_popcountll(v.m);
}
int32_t pa_or_v_value_of_prefix(pa_v_t v, int32_t j) {
    uint64_t mask = (1ull << j) - 1; /* j < 64 */
    return (int32_t)__builtin_popcountll(v.p & mask) - (int32_t)__builtin_popcountll(v.m & mask);
}
int32_t pa_or_v_value_of_suffix(pa_v_t v, int32_t j) {
    /* !((1 << (64-j)).wrapping_sub(1)); j==64 -> shift by 0 -> mask = !0 */
    uint64_t mask = ~((1ull << (64 - j)) - 1);
    return (int32_t)__builtin_popcountll(v.p & mask) - (int32_t)__builtin_popcountll(v.m & mask);
}

/* ---- profile.rs:112-133 (bio RankTransform over "ACGT": A0 C1 G2 T3) ---------------------- */
static int rank_acgt(uint8_t c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}

int pa_or_bitprofile_build(const uint8_t* a, size_t n, const uint8_t* b, size_t m,
                           pa_bits_t* pa, pa_bits_t* pb) {
    for (size_t i = 0; i < n; ++i) {
        int r = rank_acgt(a[i]);
        if (r < 0) return -1;
        pa[i].b0 = 0ull - (uint64_t)(r & 1);
        pa[i].b1 = 0ull - (uint64_t)((r >> 1) & 1);
    }
    size_t w = (m + 63) / 64;
    for (size_t j = 0; j < w; ++j) pb[j].b0 = pb[j].b1 = 0;
    for (size_t j = 0; j < m; ++j) {
        int r = rank_acgt(b[j]);
        if (r < 0) return -1;
        pb[j / 64].b0 |= (uint64_t)((r & 1) ^ 1) << (j % 64);
        pb[j / 64].b1 |= (uint64_t)(((r >> 1) & 1) ^ 1) << (j % 64);
    }
    return 0;
}

/* ---- myers.rs:27-55 ------------------------------------------------------------------------ */
static inline void myers_step(uint64_t eq, pa_h_t* h0, pa_v_t* v) {
    uint64_t vp = v->p, vm = v->m;
    uint64_t vx = eq | vm;
    eq |= h0->m;
    uint64_t hx = (((eq & vp) + vp) ^ vp) | eq;
    uint64_t hp = vm | ~(hx | vp);
    uint64_t hm = vp & hx;
    uint64_t hpw = hp >> 63, hmw = hm >> 63;
    hp = (hp << 1) | h0->p;
    hm = (hm << 1) | h0->m;
    h0->p = hpw;
    h0->m = hmw;
    v->p = hm | ~(vx | hp);
    v->m = hp & vx;
}

void pa_or_compute_block(pa_h_t* h0, pa_v_t* v, pa_bits_t ca, pa_bits_t cb) {
    uint64_t eq = (ca.b0 ^ cb.b0) & (ca.b1 ^ cb.b1); /* profile.rs:141-144 */
    myers_step(eq, h0, v);
}

static int32_t sum_h(const pa_h_t* h, size_t n) {
    int32_t s = 0;
    for (size_t i = 0; i < n; ++i) s += (int32_t)h[i].p - (int32_t)h[i].m;
    return s;
}

/* ---- scalar.rs:37-46 ----------------------------------------------------------------------- */
int32_t pa_or_scalar_row(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                         pa_h_t* h, pa_v_t* v) {
    for (size_t j = 0; j < w; ++j)
        for (size_t i = 0; i < n; ++i) pa_or_compute_block(&h[i], &v[j], a[i], b[j]);
    return sum_h(h, n);
}
/* ---- scalar.rs:9-18 ------------------------------------------------------------------------ */
int32_t pa_or_scalar_col(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                         pa_h_t* h, pa_v_t* v) {
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < w; ++j) pa_or_compute_block(&h[i], &v[j], a[i], b[j]);
    return sum_h(h, n);
}
/* ---- scalar.rs:405-425 --------------------------------------------------------------------- */
int32_t pa_or_scalar_fill(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                          pa_h_t* h, pa_v_t* v, pa_v_t* values) {
    for (size_t i = 0; i < n; ++i) {
        for (size_t j = 0; j < w; ++j) pa_or_compute_block(&h[i], &v[j], a[i], b[j]);
        memcpy(values + i * w, v, w * sizeof(pa_v_t));
    }
    return sum_h(h, n);
}

/* ---- simd.rs:112-134,184-218: how many pad rows the reference's dispatch appends ------------ */
size_t pa_or_simd_pad_rows(size_t n, size_t w, int exact_end, int ilp_n) {
    if (exact_end) return 0;
    size_t L = 4, N = (size_t)ilp_n;
    while (n < 2 * L * N) { /* simd.rs:112-126 */
        if (N > 1) { N = 1; continue; }
        if (L > 2) { L = 2; continue; }
        return 0; /* pure scalar double loop */
    }
    if (w == 1) return 0; /* simd.rs:129-134 */
    size_t rem = w % (L * N);
    switch (rem) { /* simd.rs:190-223 */
        case 0: case 1: case 2: return 0;
        case 3: case 4: return 4 - rem;
        default: return 8 - rem; /* 5,6,7 */
    }
}

int32_t pa_or_simd_compute(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                           pa_h_t* h, pa_v_t* v, int exact_end, int ilp_n) {
    size_t pad = pa_or_simd_pad_rows(n, w, exact_end, ilp_n);
    if (pad == 0) return pa_or_scalar_row(a, n, b, w, h, v);
    /* Non-exact tail: the last `l` real rows run together with `pad` rows of Bits(0,0), V(0,0);
     * returned sum = sum(bottom h of padded block) - sum(value of pad v's). simd.rs:202-224 */
    pa_or_scalar_row(a, n, b, w, h, v);
    pa_bits_t zb = {0, 0};
    int32_t correction = 0;
    for (size_t k = 0; k < pad; ++k) {
        pa_v_t vt = {0, 0};
        for (size_t i = 0; i < n; ++i) pa_or_compute_block(&h[i], &vt, a[i], zb);
        correction += pa_or_v_value(vt);
    }
    return sum_h(h, n) - correction;
}

int32_t pa_or_simd_fill(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                        pa_h_t* h, pa_v_t* v, pa_v_t* values) {
    /* exact mode only (simd.rs:434-436); values identical to scalar::fill. */
    return pa_or_scalar_fill(a, n, b, w, h, v, values);
}

/* ---- ScatterProfile, profile.rs:25-75 ------------------------------------------------------ */
static int scatter_char(uint8_t c) { /* get_char, profile.rs:30-38 */
    switch (c) {
        case 'a': case 'A': return 0;
        case 'c': case 'C': return 1;
        case 't': case 'T': return 2;
        case 'g': case 'G': return 3;
        default: return -1;
    }
}
static int scatter_mask(uint8_t c, int mask[4]) { /* get_mask, profile.rs:39-50 */
    static const int A[4] = {1, 0, 0, 0}, C[4] = {0, 1, 0, 0}, T[4] = {0, 0, 1, 0}, G[4] = {0, 0, 0, 1},
                     N[4] = {1, 1, 1, 1}, Y[4] = {0, 1, 1, 0}, R[4] = {1, 0, 0, 1};
    const int* s;
    switch (c) {
        case 'a': case 'A': s = A; break;
        case 'c': case 'C': s = C; break;
        case 't': case 'T': s = T; break;
        case 'g': case 'G': s = G; break;
        case 'n': case 'N': case '*': s = N; break;
        case 'y': case 'Y': s = Y; break;
        case 'r': case 'R': s = R; break;
        default: return -1;
    }
    memcpy(mask, s, sizeof(int) * 4);
    return 0;
}

/* ---- search.rs:46-120 ---------------------------------------------------------------------- */
int pa_or_search(const uint8_t* pattern, size_t plen, const uint8_t* text, size_t tlen,
                 float unmatched_cost, int32_t* out) {
    size_t w = (plen + 63) / 64;
    uint8_t* t = (uint8_t*)malloc(tlen ? tlen : 1);
    uint64_t (*p)[4] = (uint64_t (*)[4])calloc(w ? w : 1, sizeof(uint64_t[4]));
    pa_v_t* v0 = (pa_v_t*)calloc(w ? w : 1, sizeof(pa_v_t));
    pa_v_t* v = (pa_v_t*)calloc(w ? w : 1, sizeof(pa_v_t));
    pa_h_t* h = (pa_h_t*)calloc(tlen ? tlen : 1, sizeof(pa_h_t));
    int rc = -1;
    /* ScatterProfile::build(text, pattern): a = text, b = pattern. */
    for (size_t i = 0; i < tlen; ++i) {
        int c = scatter_char(text[i]);
        if (c < 0) goto done;
        t[i] = (uint8_t)c;
    }
    for (size_t j = 0; j < plen; ++j) {
        int mask[4];
        if (scatter_mask(pattern[j], mask) < 0) goto done;
        for (int k = 0; k < 4; ++k) p[j / 64][k] |= (uint64_t)mask[k] << (j % 64);
    }
    for (size_t j = plen; j < w * 64; ++j) /* padding rows match everything, profile.rs:58-62 */
        for (int k = 0; k < 4; ++k) p[j / 64][k] |= 1ull << (j % 64);

    size_t padding = w * 64 - plen;
    if (!(unmatched_cost >= 0.0f && unmatched_cost <= 1.0f)) goto done;
    if (unmatched_cost > 0.0f) { /* search.rs:57-65 */
        for (size_t i = 0;; ++i) {
            size_t idx = (size_t)ceilf((float)i / unmatched_cost);
            if (idx >= plen) break;
            v0[idx / 64].p |= 1ull << (idx % 64);
        }
    }
    memcpy(v, v0, w * sizeof(pa_v_t));
    int32_t bot_left = 0;
    for (size_t j = 0; j < w; ++j) bot_left += pa_or_v_value(v[j]);

    /* scatter_profile::compute::<2,_,4,false>(t,p,h=zeros,v,exact_end=true): schedule independent. */
    for (size_t j = 0; j < w; ++j)
        for (size_t i = 0; i < tlen; ++i) myers_step(p[j][t[i]], &h[i], &v[j]);

    {
        size_t k = 0, skipped = 0;
        int32_t bsum = bot_left;
        out[k++] = bsum;
        for (size_t i = 0; i < tlen; ++i) { /* search.rs:76-83 */
            bsum += (int32_t)h[i].p - (int32_t)h[i].m;
            if (skipped < padding) skipped++;
            else out[k++] = bsum;
        }
        for (size_t jj = w; jj-- > 0;) { /* search.rs:86-99 */
            for (int j = 1; j <= 64; ++j) {
                int32_t delta = pa_or_v_value_of_suffix(v[jj], j);
                int32_t unmatched = pa_or_v_value_of_suffix(v0[jj], j);
                int32_t val = bsum - delta + unmatched;
                if (skipped < padding) skipped++;
                else out[k++] = val;
            }
            bsum -= pa_or_v_value(v[jj]);
            bsum += pa_or_v_value(v0[jj]);
        }
        rc = (k == plen + tlen + 1) ? 0 : -2;
    }
done:
    free(t); free(p); free(v0); free(v); free(h);
    return rc;
}

/* ---- SearchResult::trace, search.rs:104-228 ------------------------------------------------- */
static int32_t vec_value_to(const pa_v_t* v, int64_t j) { /* V::value_to, encoding.rs:57-66 */
    int32_t s = 0;
    for (int64_t k = 0; k < j / 64; ++k) s += pa_or_v_value(v[k]);
    if (j % 64 != 0) s += pa_or_v_value_of_prefix(v[j / 64], (int)(j % 64));
    return s;
}
static int32_t vec_value_from(const pa_v_t* v, size_t w, int64_t j) { /* V::value_from, encoding.rs:67-76 */
    int32_t s = 0;
    if (j % 64 != 0) s += pa_or_v_value_of_suffix(v[j / 64], (int)(64 - j % 64));
    for (size_t k = (size_t)((j + 63) / 64); k < w; ++k) s += pa_or_v_value(v[k]);
    return s;
}

int pa_or_search_trace(const uint8_t* pattern, size_t plen, const uint8_t* text, size_t tlen, float unmatched_cost,
                       size_t idx, char* cigar_buf, size_t cigar_cap, int32_t* path_buf, size_t path_cap, size_t* npos_out) {
    size_t w = (plen + 63) / 64;
    int rc = -1;
    int32_t* out = (int32_t*)malloc((plen + tlen + 1) * sizeof(int32_t));
    uint8_t* t = (uint8_t*)malloc(tlen ? tlen : 1);
    uint64_t (*p)[4] = (uint64_t (*)[4])calloc(w ? w : 1, sizeof(uint64_t[4]));
    pa_v_t* v0 = (pa_v_t*)calloc(w ? w : 1, sizeof(pa_v_t));
    pa_v_t* fill = NULL;
    uint8_t* ops = NULL;   /* one op per step, end -> start */
    int32_t* poss = NULL;  /* (i, j) per visited position, end -> start */
    if (idx > plen + tlen || w == 0) goto done;
    if (pa_or_search(pattern, plen, text, tlen, unmatched_cost, out) != 0) goto done;
    for (size_t i = 0; i < tlen; ++i) t[i] = (uint8_t)scatter_char(text[i]);
    for (size_t j = 0; j < plen; ++j) {
        int mask[4];
        scatter_mask(pattern[j], mask);
        for (int k = 0; k < 4; ++k) p[j / 64][k] |= (uint64_t)mask[k] << (j % 64);
    }
    for (size_t j = plen; j < w * 64; ++j)
        for (int k = 0; k < 4; ++k) p[j / 64][k] |= 1ull << (j % 64);
    if (unmatched_cost > 0.0f)
        for (size_t i = 0;; ++i) {
            size_t k = (size_t)ceilf((float)i / unmatched_cost);
            if (k >= plen) break;
            v0[k / 64].p |= 1ull << (k % 64);
        }
    {
        /* idx_to_pos, search.rs:105-115 */
        int64_t pi, pj;
        if (idx <= tlen) { pi = (int64_t)idx; pj = (int64_t)plen; }
        else { pi = (int64_t)tlen; pj = (int64_t)plen - ((int64_t)idx - (int64_t)tlen); }
        int32_t target = out[idx];
        if ((size_t)pi == tlen) target -= vec_value_from(v0, w, pj);
        size_t width = 2 * plen, end = (size_t)pi, start;
        for (;;) { /* search.rs:140-177 */
            start = end > width ? end - width : 0;
            size_t cols = end - start;
            free(fill);
            fill = (pa_v_t*)malloc((cols + 1) * w * sizeof(pa_v_t));
            for (size_t k = 0; k < w; ++k) fill[k] = start == 0 ? v0[k] : (pa_v_t){~0ull, 0};
            for (size_t c = 0; c < cols; ++c) { /* h = zero along the top; one column at a time */
                pa_h_t h = {0, 0};
                for (size_t k = 0; k < w; ++k) {
                    pa_v_t vv = fill[c * w + k];
                    myers_step(p[k][t[start + c]], &h, &vv);
                    fill[(c + 1) * w + k] = vv;
                }
            }
            int32_t cost = vec_value_to(fill + cols * w, pj);
            if (cost < target) goto done; /* "found trace path of cost < target" assert */
            if (cost == target) break;
            if (start == 0) goto done;
            width *= 2;
        }
        size_t cap = (size_t)pi - start + (size_t)pj + 2, nops = 0, np = 0;
        ops = (uint8_t*)malloc(cap);
        poss = (int32_t*)malloc(2 * cap * sizeof(int32_t));
#define COSTAT(i_, j_) vec_value_to(fill + ((size_t)(i_) - start) * w, (j_))
        int32_t g = target;
        poss[0] = (int32_t)pi; poss[1] = (int32_t)pj; np = 1;
        while (pi > (int64_t)start && pj > 0) { /* search.rs:185-224 */
            int cnt = 0;
            while (pi > (int64_t)start && pj > 0 && ((p[(pj - 1) / 64][t[pi - 1]] >> ((pj - 1) % 64)) & 1)) {
                ++cnt; --pi; --pj;
                ops[nops++] = '=';
                poss[2 * np] = (int32_t)pi; poss[2 * np + 1] = (int32_t)pj; ++np;
            }
            if (cnt > 0) continue;
            if (COSTAT(pi - 1, pj) == g - 1) { --g; --pi; ops[nops++] = 'D'; }
            else if (COSTAT(pi, pj - 1) == g - 1) { --g; --pj; ops[nops++] = 'I'; }
            else if (COSTAT(pi - 1, pj - 1) == g - 1) { --g; --pi; --pj; ops[nops++] = 'X'; }
            else goto done; /* "Bad trace!" */
            poss[2 * np] = (int32_t)pi; poss[2 * np + 1] = (int32_t)pj; ++np;
        }
#undef COSTAT
        if (!(pi == 0 || g == 0)) goto done;
        /* reverse into the outputs; CIGAR in "=I4=X=" form */
        size_t pos = 0;
        for (size_t k = nops; k > 0;) {
            size_t run = 1;
            uint8_t op = ops[k - 1];
            while (k - run > 0 && ops[k - run - 1] == op) ++run;
            char tmp[32];
            int len = run == 1 ? snprintf(tmp, sizeof tmp, "%c", op) : snprintf(tmp, sizeof tmp, "%zu%c", run, op);
            if (pos + (size_t)len + 1 > cigar_cap) goto done;
            memcpy(cigar_buf + pos, tmp, (size_t)len);
            pos += (size_t)len;
            k -= run;
        }
        cigar_buf[pos] = 0;
        if (np > path_cap) goto done;
        for (size_t k = 0; k < np; ++k) {
            path_buf[2 * k] = poss[2 * (np - 1 - k)];
            path_buf[2 * k + 1] = poss[2 * (np - 1 - k) + 1];
        }
        *npos_out = np;
        rc = 0;
    }
done:
    free(out); free(t); free(p); free(v0); free(fill); free(ops); free(poss);
    return rc;
}

/* ---- plain Levenshtein --------------------------------------------------------------------- */
int32_t pa_or_levenshtein(const uint8_t* a, size_t n, const uint8_t* b, size_t m) {
    int32_t* row = (int32_t*)malloc((m + 1) * sizeof(int32_t));
    for (size_t j = 0; j <= m; ++j) row[j] = (int32_t)j;
    for (size_t i = 1; i <= n; ++i) {
        int32_t diag = row[0];
        row[0] = (int32_t)i;
        for (size_t j = 1; j <= m; ++j) {
            int32_t up = row[j];
            int32_t best = diag + (a[i - 1] != b[j - 1]);
            if (up + 1 < best) best = up + 1;
            if (row[j - 1] + 1 < best) best = row[j - 1] + 1;
            row[j] = best;
            diag = up;
        }
    }
    int32_t d = row[m];
    free(row);
    return d;
}

/* ---- Cigar::verify at unit cost ------------------------------------------------------------ */
int32_t pa_or_cigar_verify(const char* cigar, const uint8_t* a, size_t n, const uint8_t* b, size_t m) {
    size_t i = 0, j = 0;
    int32_t cost = 0;
    const char* c = cigar;
    while (*c) {
        size_t cnt = 0;
        int have = 0;
        while (*c >= '0' && *c <= '9') { cnt = cnt * 10 + (size_t)(*c - '0'); have = 1; ++c; }
        if (!have) cnt = 1;
        if (cnt == 0) return -1;
        char op = *c++;
        switch (op) {
            case '=':
                for (size_t k = 0; k < cnt; ++k, ++i, ++j)
                    if (i >= n || j >= m || a[i] != b[j]) return -1;
                break;
            case 'X':
                for (size_t k = 0; k < cnt; ++k, ++i, ++j)
                    if (i >= n || j >= m || a[i] == b[j]) return -1;
                cost += (int32_t)cnt;
                break;
            case 'I': /* advances b */
                j += cnt; if (j > m) return -1; cost += (int32_t)cnt; break;
            case 'D': /* advances a */
                i += cnt; if (i > n) return -1; cost += (int32_t)cnt; break;
            default: return -1;
        }
    }
    if (i != n || j != m) return -1;
    return cost;
}

/* ---- NW cost-only, driven like AstarPa2Params::nw() cost mode ------------------------------ */
int32_t pa_or_nw_cost(const uint8_t* a, size_t n, const uint8_t* b, size_t m, int use_avx2) {
    size_t w = (m + 63) / 64;
    pa_bits_t* pa = (pa_bits_t*)malloc((n ? n : 1) * sizeof(pa_bits_t));
    pa_bits_t* pb = (pa_bits_t*)malloc((w ? w : 1) * sizeof(pa_bits_t));
    pa_v_t* v = (pa_v_t*)malloc((w ? w : 1) * sizeof(pa_v_t));
    pa_h_t h[256];
    int32_t bot = (int32_t)(w * 64); /* bot_val of the first column = rounded len, blocks.rs:171 */
    if (pa_or_bitprofile_build(a, n, b, m, pa, pb) != 0) { bot = -1; goto done; }
    for (size_t j = 0; j < w; ++j) { v[j].p = ~0ull; v[j].m = 0; }
    for (size_t i = 0; i < n; i += 256) {
        size_t len = n - i < 256 ? n - i : 256;
        for (size_t k = 0; k < len; ++k) { h[k].p = 1; h[k].m = 0; }
        if (w == 0) { bot += (int32_t)len; continue; }
        bot += use_avx2 ? pa_or_strip_compute_avx2(pa + i, len, pb, w, h, v, 0)
                        : pa_or_simd_compute(pa + i, len, pb, w, h, v, 0, 2);
    }
    /* last_block.get(|b|): bot_val minus the suffix below row m in the last word, block.rs:110-120 */
    if (m % 64 != 0 && w > 0) bot -= pa_or_v_value_of_suffix(v[w - 1], (int32_t)(w * 64 - m));
done:
    free(pa); free(pb); free(v);
    return bot;
}
