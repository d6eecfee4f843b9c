/* The C side of the struct layouts the Rust mirrors in rust-shim/ declare (#[repr(C)], same field order), pinned with
 * _Static_assert, plus a run of the aligner and operator entry points from plain C (include/pa_astarpa2.h,
 * include/pa_bitpacking_hip.h).  Built with gcc by tests/test_gpu_engine.py::test_c_layout_program and (compile only, no
 * GPU) by tests/test_capi_symbols.py. */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "astarpa.h"
#include "pa_astarpa2.h"
#include "pa_bitpacking_hip.h"

#define OFF(T, f, o) _Static_assert(offsetof(T, f) == (o), #T "." #f " offset")

/* pa_block_params: 7 x i32 */
OFF(pa_block_params, sparse, 0);
OFF(pa_block_params, simd, 4);
OFF(pa_block_params, no_ilp, 8);
OFF(pa_block_params, incremental_doubling, 12);
OFF(pa_block_params, dt_trace, 16);
OFF(pa_block_params, max_g, 20);
OFF(pa_block_params, fr_drop, 24);
_Static_assert(sizeof(pa_block_params) == 28, "pa_block_params size");

/* pa_astarpa2_params */
OFF(pa_astarpa2_params, domain, 0);
OFF(pa_astarpa2_params, heuristic, 4);
OFF(pa_astarpa2_params, heuristic_k, 8);
OFF(pa_astarpa2_params, heuristic_p, 12);
OFF(pa_astarpa2_params, doubling, 16);
OFF(pa_astarpa2_params, doubling_start, 20);
OFF(pa_astarpa2_params, factor, 24);
OFF(pa_astarpa2_params, delta, 28);
OFF(pa_astarpa2_params, block_width, 32);
OFF(pa_astarpa2_params, front, 36);
OFF(pa_astarpa2_params, sparse_h, 64);
OFF(pa_astarpa2_params, prune, 68);
_Static_assert(sizeof(pa_astarpa2_params) == 72, "pa_astarpa2_params size");
_Static_assert(sizeof(float) == 4 && sizeof(int32_t) == 4, "scalar sizes");

/* pa_astarpa2_stats: 12 x u64, 8 x f64 */
OFF(pa_astarpa2_stats, num_blocks, 0);
OFF(pa_astarpa2_stats, num_incremental_blocks, 8);
OFF(pa_astarpa2_stats, computed_lanes, 16);
OFF(pa_astarpa2_stats, unique_lanes, 24);
OFF(pa_astarpa2_stats, dt_trace_tries, 32);
OFF(pa_astarpa2_stats, dt_trace_success, 40);
OFF(pa_astarpa2_stats, dt_trace_fallback, 48);
OFF(pa_astarpa2_stats, fill_tries, 56);
OFF(pa_astarpa2_stats, fill_success, 64);
OFF(pa_astarpa2_stats, fill_fallback, 72);
OFF(pa_astarpa2_stats, f_max_tries, 80);
OFF(pa_astarpa2_stats, sanity_violations, 88);
OFF(pa_astarpa2_stats, t_compute, 96);
OFF(pa_astarpa2_stats, t_dt, 104);
OFF(pa_astarpa2_stats, t_fill, 112);
OFF(pa_astarpa2_stats, t_precomp, 120);
OFF(pa_astarpa2_stats, t_j_range, 128);
OFF(pa_astarpa2_stats, t_fixed_j_range, 136);
OFF(pa_astarpa2_stats, t_pruning, 144);
OFF(pa_astarpa2_stats, t_contours_update, 152);
_Static_assert(sizeof(pa_astarpa2_stats) == 160, "pa_astarpa2_stats size");

int main(void) {
    const char* a = "ACTCGCT";
    const char* b = "AACTCGTT";
    pa_astarpa2_params p;
    const char* names[3] = {"nw", "simple", "full"};
    for (int which = 0; which < 3; ++which) {
        memset(&p, 0xAB, sizeof(p));
        if (which == 0) pa_params_nw(&p);
        if (which == 1) pa_params_simple(&p);
        if (which == 2) pa_params_full(&p);
        int32_t cost = -1;
        char* cigar = NULL;
        pa_astarpa2_stats st;
        memset(&st, 0, sizeof(st));
        const int rc = pa_align((const uint8_t*)a, strlen(a), (const uint8_t*)b, strlen(b), &p, 1, &cost, &cigar, &st);
        printf("pa_align %s rc=%d cost=%d cigar=%s block_width=%d tries=%llu\n", names[which], rc, (int)cost, cigar ? cigar : "(null)", (int)p.block_width,
               (unsigned long long)st.f_max_tries);
        if (rc != 0) return 1;
        free(cigar);
    }
    /* the operator boundary from C: profile, then one 7-column x 1-word rectangle with h = v = +1 */
    uint64_t a2[2 * 7], b2[2 * 1], h2[2 * 7], v2[2 * 1];
    if (pa_bp_profile_build((const uint8_t*)a, 7, (const uint8_t*)b, 8, a2, b2) != 0) return 2;
    for (int i = 0; i < 7; ++i) {
        h2[2 * i] = 1;
        h2[2 * i + 1] = 0;
    }
    v2[0] = ~(uint64_t)0;
    v2[1] = 0;
    const int32_t sum = pa_bp_compute(a2, 7, b2, 1, h2, v2, 1);
    /* bottom row of the 64-row word: D[7][64] - D[0][64]; the cost at row 8 follows from v: D[7][8] = 7 + sum of the first 8 deltas */
    int d = 7;
    for (int k = 0; k < 8; ++k) d += (int)((v2[0] >> k) & 1) - (int)((v2[1] >> k) & 1);
    printf("pa_bp_compute sum=%d cost=%d\n", (int)sum, d);
    return d == 2 ? 0 : 3;
}
