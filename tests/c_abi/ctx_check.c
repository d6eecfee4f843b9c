/* The device-resident operator handles (pa_bp_ctx_*, include/pa_bitpacking_hip.h) from plain C, the way a host engine calls them
 * block by block (astarpa2/src/blocks.rs:686-748): a pair of 600 x 500 characters is computed
 *   (1) in one piece by the stateless operator pa_bp_compute (top row +1, left column +1), and
 *   (2) by the handle as 3 column blocks x 2 row ranges: the upper range with HMode::Output (top +1, bottom row stored), the lower
 *       with HMode::Input (top row = the stored row); then the upper range again with HMode::Update on a fresh left column, which
 *       must reproduce the same stored row -- the chain incremental doubling runs (blocks.rs:392-468).
 * Both must give the same right-edge column and the same bottom-right value.  pa_bp_ctx_fill then re-fills the last block and
 * its last column must equal the compute result.  Prints "ctx_check ok cost=<edit distance>". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pa_bitpacking_hip.h"

#define N 600
#define M 500

int main(void) {
    static uint8_t a[N], b[M];
    unsigned s = 12345;
    for (int i = 0; i < N; ++i) {
        s = s * 1103515245u + 12345u;
        a[i] = "ACGT"[(s >> 16) & 3];
    }
    for (int j = 0; j < M; ++j) b[j] = j < N ? a[j] : 'A';
    for (int j = 7; j < M; j += 23) b[j] = b[j] == 'A' ? 'C' : 'A'; /* some substitutions */
    const size_t w = (M + 63) / 64, wtop = 3;
    /* (1) stateless */
    uint64_t* a2 = calloc(2 * N, 8);
    uint64_t* b2 = calloc(2 * w, 8);
    uint64_t* h2 = calloc(2 * N, 8);
    uint64_t* v_ref = calloc(2 * w, 8);
    if (pa_bp_profile_build(a, N, b, M, a2, b2) != 0) {
        fprintf(stderr, "profile: %s\n", pa_last_error());
        return 1;
    }
    for (int i = 0; i < N; ++i) h2[2 * i] = 1;
    for (size_t j = 0; j < w; ++j) v_ref[2 * j] = ~0ull;
    const int32_t bottom_ref = pa_bp_compute(a2, N, b2, w, h2, v_ref, 1);
    if (bottom_ref == INT32_MIN) {
        fprintf(stderr, "compute: %s\n", pa_last_error());
        return 1;
    }
    /* (2) the handle, block by block */
    pa_bp_ctx* ctx = pa_bp_ctx_create(a, N, b, M);
    if (!ctx) {
        fprintf(stderr, "ctx_create: %s\n", pa_last_error());
        return 1;
    }
    uint64_t* v = calloc(2 * w, 8);
    for (size_t j = 0; j < w; ++j) v[2 * j] = ~0ull;
    const int32_t cuts[4] = {0, 256, 512, N};
    int32_t bottom = 0;
    for (int k = 0; k < 3; ++k) {
        int32_t s_top = 0, s_bot = 0, s_again = 0;
        uint64_t keep[2 * 3];
        memcpy(keep, v, sizeof keep);
        if (pa_bp_ctx_compute(ctx, cuts[k], cuts[k + 1], 0, wtop, v, /*Output*/ 3, &s_top) != 0 ||
            pa_bp_ctx_compute(ctx, cuts[k], cuts[k + 1], wtop, w, v + 2 * wtop, /*Input*/ 1, &s_bot) != 0) {
            fprintf(stderr, "ctx_compute: %s\n", pa_last_error());
            return 1;
        }
        /* the upper range once more from its old left column, reading and rewriting the stored row: same row, same words */
        uint64_t again[2 * 3];
        memcpy(again, keep, sizeof again);
        if (pa_bp_ctx_compute(ctx, cuts[k], cuts[k + 1], 0, 0, again, /*Update*/ 2, &s_again) != 0) { /* zero rows: the stored row passes through */
            fprintf(stderr, "ctx_compute(update): %s\n", pa_last_error());
            return 1;
        }
        if (s_again != s_top) {
            fprintf(stderr, "block %d: stored row sums to %d, the Output pass returned %d\n", k, s_again, s_top);
            return 1;
        }
        bottom += s_bot;
    }
    if (memcmp(v, v_ref, 16 * w) != 0 || bottom != bottom_ref) {
        fprintf(stderr, "handle and stateless operator disagree (bottom %d vs %d)\n", bottom, bottom_ref);
        return 1;
    }
    /* fill: the last block again from the column at 512 */
    uint64_t* v512 = calloc(2 * w, 8);
    for (size_t j = 0; j < w; ++j) v512[2 * j] = ~0ull;
    int32_t dummy = 0;
    if (pa_bp_ctx_compute(ctx, 0, 512, 0, w, v512, /*None*/ 0, &dummy) != 0) return 1;
    const size_t cols = N - 512;
    uint64_t* values = calloc(2 * cols * w, 8);
    int8_t* hb = calloc(cols, 1);
    if (pa_bp_ctx_fill(ctx, 512, N, 0, w, v512, values, hb) != 0) {
        fprintf(stderr, "ctx_fill: %s\n", pa_last_error());
        return 1;
    }
    if (memcmp(values + 2 * (cols - 1) * w, v_ref, 16 * w) != 0) {
        fprintf(stderr, "the last filled column differs from the computed one\n");
        return 1;
    }
    pa_bp_ctx_destroy(ctx);
    /* edit distance = D[N][M] = (value at the bottom-left, M... here rows are padded to 64: use the right column instead) */
    long cost = N; /* D[N][0] = N, then down the right edge: + sum of vertical deltas of rows 0..M-1 */
    for (int j = 0; j < M; ++j) cost += (long)((v_ref[2 * (j / 64)] >> (j % 64)) & 1) - (long)((v_ref[2 * (j / 64) + 1] >> (j % 64)) & 1);
    printf("ctx_check ok cost=%ld\n", cost);
    return 0;
}
