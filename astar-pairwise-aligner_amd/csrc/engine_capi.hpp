// engine_capi.hpp -- glue between the C structs of include/pa_astarpa2.h and engine.hpp.
#pragma once
#include "../../include/pa_astarpa2.h"
#include "engine.hpp"

namespace pa {
namespace engine {

inline AstarPa2Params params_from_c(const pa_astarpa2_params& c) {
    AstarPa2Params p;
    p.domain = (DomainKind)c.domain;
    p.heuristic = (HeuristicKind)c.heuristic;
    p.heuristic_k = c.heuristic_k;
    p.heuristic_p = c.heuristic_p;
    p.doubling = (DoublingKind)c.doubling;
    p.start = (DoublingStart)c.doubling_start;
    p.factor = c.factor;
    p.delta = c.delta;
    p.block_width = c.block_width;
    p.front.sparse = c.front.sparse != 0;
    p.front.simd = c.front.simd != 0;
    p.front.no_ilp = c.front.no_ilp != 0;
    p.front.incremental_doubling = c.front.incremental_doubling != 0;
    p.front.dt_trace = c.front.dt_trace != 0;
    p.front.max_g = c.front.max_g;
    p.front.fr_drop = c.front.fr_drop;
    p.sparse_h = c.sparse_h != 0;
    p.prune = c.prune != 0;
    return p;
}

inline void params_to_c(const AstarPa2Params& p, pa_astarpa2_params* c) {
    c->domain = (int32_t)p.domain;
    c->heuristic = (int32_t)p.heuristic;
    c->heuristic_k = p.heuristic_k;
    c->heuristic_p = p.heuristic_p;
    c->doubling = (int32_t)p.doubling;
    c->doubling_start = (int32_t)p.start;
    c->factor = p.factor;
    c->delta = p.delta;
    c->block_width = p.block_width;
    c->front.sparse = p.front.sparse;
    c->front.simd = p.front.simd;
    c->front.no_ilp = p.front.no_ilp;
    c->front.incremental_doubling = p.front.incremental_doubling;
    c->front.dt_trace = p.front.dt_trace;
    c->front.max_g = p.front.max_g;
    c->front.fr_drop = p.front.fr_drop;
    c->sparse_h = p.sparse_h;
    c->prune = p.prune;
}

inline void stats_to_c(const AstarPa2Stats& s, pa_astarpa2_stats* c) {
    c->num_blocks = s.block_stats.num_blocks;
    c->num_incremental_blocks = s.block_stats.num_incremental_blocks;
    c->computed_lanes = s.block_stats.computed_lanes;
    c->unique_lanes = s.block_stats.unique_lanes;
    c->dt_trace_tries = s.trace_stats.dt_trace_tries;
    c->dt_trace_success = s.trace_stats.dt_trace_success;
    c->dt_trace_fallback = s.trace_stats.dt_trace_fallback;
    c->fill_tries = s.trace_stats.fill_tries;
    c->fill_success = s.trace_stats.fill_success;
    c->fill_fallback = s.trace_stats.fill_fallback;
    c->f_max_tries = s.f_max_tries;
    c->sanity_violations = s.sanity_violations;
    c->t_compute = s.block_stats.t_compute;
    c->t_dt = s.trace_stats.t_dt;
    c->t_fill = s.trace_stats.t_fill;
    c->t_precomp = s.t_precomp;
    c->t_j_range = s.t_j_range;
    c->t_fixed_j_range = s.t_fixed_j_range;
    c->t_pruning = s.t_pruning;
    c->t_contours_update = s.t_contours_update;
}

inline bool params_valid(const pa_astarpa2_params& c) {
    return c.domain >= 0 && c.domain <= 3 && c.heuristic >= 0 && c.heuristic <= 3 && !(c.heuristic >= PA_HEURISTIC_SH && (c.heuristic_k < 1 || c.heuristic_k > 31)) &&
           c.heuristic_p >= 0 && c.doubling >= 0 && c.doubling <= 2 &&
           c.doubling_start >= 0 && c.doubling_start <= 2 && c.block_width >= 1 &&
           !(c.doubling == PA_DOUBLING_NONE && c.domain != PA_DOMAIN_FULL) &&
           // a band that does not grow never ends the search (band.rs:138: factor <= 1 or NaN; a delta below 1 or beyond i32)
           !(c.doubling == PA_DOUBLING_BAND && !(c.factor > 1.0f && c.factor <= 1.0e6f)) &&
           !(c.doubling == PA_DOUBLING_LINEAR && !(c.delta >= 1.0f && c.delta <= 1073741824.0f)) && c.front.max_g >= 0 &&
           !(c.front.sparse == 0 && c.doubling != PA_DOUBLING_NONE && c.front.incremental_doubling);
}

}  // namespace engine
}  // namespace pa
