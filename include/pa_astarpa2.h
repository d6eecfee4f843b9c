/*
 * pa_astarpa2.h -- C mirror of the A*PA2 aligner surface (parameters, stats, one entry point).
 *
 * Replaces, for the hot path only, the Rust surface
 *   AstarPa2Params::{nw,simple,full}().make_aligner(trace) -> Box<dyn AstarPa2StatsAligner>
 *   AstarPa2StatsAligner::align_with_stats(a, b) -> (Cost, Option<Cigar>, AstarPa2Stats)
 * (astarpa2/src/params.rs:8-42,46-128,132; astarpa2/src/lib.rs:200-214; blocks.rs:31-84; domain.rs:31-43).
 * Field names and meanings follow the reference structs one for one.
 */
#ifndef PA_ASTARPA2_H
#define PA_ASTARPA2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PA_DOMAIN_FULL = 0, PA_DOMAIN_GAP_START = 1, PA_DOMAIN_GAP_GAP = 2, PA_DOMAIN_ASTAR = 3 }; /* params.rs:231-242 */
enum { PA_HEURISTIC_NONE = 0, PA_HEURISTIC_GAP = 1, PA_HEURISTIC_SH = 2, PA_HEURISTIC_GCSH = 3 }; /* NoCost / GapCost / SH / GCSH (pa-heuristic) */
enum { PA_DOUBLING_NONE = 0, PA_DOUBLING_BAND = 1, PA_DOUBLING_LINEAR = 2 };                     /* band.rs:26-44 */
enum { PA_START_ZERO = 0, PA_START_GAP = 1, PA_START_H0 = 2 };                                   /* band.rs:5-10 */

typedef struct pa_block_params { /* BlockParams, blocks.rs:31-60 */
    int32_t sparse;
    int32_t simd;   /* accepted for compatibility: the GPU library always runs its own strip schedule */
    int32_t no_ilp; /* idem */
    int32_t incremental_doubling;
    int32_t dt_trace;
    int32_t max_g;
    int32_t fr_drop;
} pa_block_params;

typedef struct pa_astarpa2_params { /* AstarPa2Params, params.rs:8-42 */
    int32_t domain;
    int32_t heuristic;
    int32_t heuristic_k; /* HeuristicParams.k: seed length of SH / GCSH (exact matches, r = 1) */
    int32_t heuristic_p; /* HeuristicParams.p: local-pruning look-ahead of GCSH, 0 = off */
    int32_t doubling;
    int32_t doubling_start;
    float factor; /* BandDoubling: must be finite and > 1 (a band that does not grow never ends the search: rejected as invalid) */
    float delta;  /* LinearSearch: 1 <= delta <= 2^30 */
    int32_t block_width;
    pa_block_params front;
    int32_t sparse_h;
    int32_t prune;
} pa_astarpa2_params;

typedef struct pa_astarpa2_stats { /* AstarPa2Stats + BlockStats + TraceStats */
    /* The four block counters are reported after a band doubling only: the reference copies them into its result in that arm of
     * cost_or_align alone (astarpa2/src/lib.rs:158), so doubling = none (the nw preset) and the linear search return zeros here. */
    uint64_t num_blocks, num_incremental_blocks, computed_lanes, unique_lanes;
    uint64_t dt_trace_tries, dt_trace_success, dt_trace_fallback, fill_tries, fill_success, fill_fallback;
    uint64_t f_max_tries;
    uint64_t sanity_violations; /* times the reference's band.rs:117-135 asserts would have aborted */
    double t_compute, t_dt, t_fill, t_precomp, t_j_range, t_fixed_j_range, t_pruning, t_contours_update;
} pa_astarpa2_stats;

/* Presets, params.rs:46-128 (full = GCSH k=12 r=1 p=14 Prune::Start, incremental doubling, pruning). */
void pa_params_nw(pa_astarpa2_params* p);
void pa_params_simple(pa_astarpa2_params* p);
void pa_params_full(pa_astarpa2_params* p);
/* nw() with front.sparse = 1 and no DT-trace: the parameter set whose alignments pa_batch_align (pa_bitpacking_hip.h) returns. */
void pa_params_batch_align(pa_astarpa2_params* p);

/* align_with_stats.  trace != 0 => *cigar_out receives a malloc'ed NUL-terminated CIGAR ("=I4=X=" style,
 * free() it or use astarpa_free_cigar).  Returns 0, or a PA_E_* code.  All DP rectangles run on the GPU.
 * trace == 0 with Domain::Astar, sparse blocks and no incremental doubling (`simple` and its relatives): the cost is the edit
 * distance, computed over the band of the TRACED mode, and the block statistics are those of the traced band (trace statistics 0).
 * The reference's own cost-only mode for these parameters (astarpa2/src/blocks.rs:252-277: ONE block updated in place, its
 * fixed_j_range the union over all columns; no entry point or test of the reference uses it) is not reproduced: it RETURNS UPPER
 * BOUNDS on some inputs -- two independent restatements of it (csrc/engine.hpp and tests/tools/cost_only_restatement.py, pure
 * Python from the Rust text) both give 11353 for a pair whose distance is 11325, with the same passes and computed lanes
 * (tests/test_cost_only_mode.py) -- whereas this library returns the distance. */
int pa_align(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, const pa_astarpa2_params* params,
             int trace, int32_t* cost_out, char** cigar_out, pa_astarpa2_stats* stats_out);
/* The other choice for trace == 0, process-wide: on != 0 (or PA_COST_ONLY_MODE=reference in the environment) makes pa_align(trace = 0) run
 * the REFERENCE's cost-only arm for every parameter set -- blocks.rs:252-277 restated in csrc/engine.hpp, its rectangles on the GPU, one
 * launch per 256-column block: the value `AstarPa2Params::simple().make_aligner(false)` returns, upper bounds included (11353 for the pair
 * of tests/golden/cost_only_pair.json, tests/test_gpu_engine.py::test_reference_cost_only_mode_on_the_gpu).  Off by default: see above. */
void pa_set_reference_cost_only(int on);

/* Callers that are inside pa_align (or an astarpa-c symbol) AT THE SAME TIME with the same parameters, sequences shorter than 32 768
 * bases and a parameter set the batch kernels take (pa_batch_params_supported) are COMBINED: one of them aligns all queued pairs as one
 * batch on the GPU and hands every caller its own cost, CIGAR and statistics -- the values the single-pair path returns (timers 0).
 * Below a dozen concurrent callers (PA_COMBINE_MIN) everybody keeps the single-pair path and its latency; once a dozen are inside at a
 * time, everybody is combined until 20 ms after the crowd was last seen.  PA_COMBINE=0 switches it off.  pa_combine_stats: calls served that way so far and
 * the batches they went out in (either argument may be NULL). */
void pa_combine_stats(uint64_t* calls, uint64_t* batches);

/* Optional, once, BEFORE anything starts the HIP runtime in this process (torch, another library, the first pa_* call): asks the
 * runtime for 16 hardware queues (GPU_MAX_HW_QUEUES, default 4) unless the variable is set already, so that the pipelined passes
 * of one pa_align call -- each on its own stream -- run side by side (C3 `simple`: 14.5 ms with, ~20 ms without).  Exporting the
 * variable does the same.  The library never touches the environment by itself.  Returns 1 if it set the variable, else 0. */
int pa_runtime_hints(void);

/* pa_align and the astarpa-c symbols keep device buffers, pinned staging and streams in per-thread pools that only grow (after one
 * 10 Mbp alignment: gigabytes).  This frees the calling thread's pools; the next call builds them again.
 * It also returns the library's cache of large device buffers to the driver: a device buffer of 16 MB or more is kept when its
 * batch or pool lets go of it and handed to the next request it fits (hipMalloc + hipFree of a 40 GB block-column store cost about
 * a second; the cache is bounded PER DEVICE at half of the device's memory and at most 16 GB -- PA_ALLOC_CACHE_MAX bytes overrides --,
 * emptied when an allocation fails, off with PA_NO_ALLOC_CACHE=1). */
void pa_release_pools(void);
/* Diagnostics of that cache: requests served from it / not served from it, bytes it holds now.  Any argument may be NULL. */
void pa_alloc_cache_stats(uint64_t* hits, uint64_t* misses, uint64_t* cached_bytes);

#ifdef __cplusplus
}
#endif
#endif
