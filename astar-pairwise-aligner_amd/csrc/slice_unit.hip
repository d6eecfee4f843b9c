// slice_unit.hip -- plan + launches of the bit-sliced full-DP kernel (slice_kernel.hpp; interface: slice_plan.hpp).
#include "slice_plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <numeric>
#include <vector>

#include "pa_hip_internal.hpp"
#include "slice_kernel.hpp"

namespace pa {
namespace slice {

struct Plan {
    int R = 52;
    size_t pairs = 0;
    std::vector<SliceGroup> groups;
    std::vector<SliceJob> jobs;
    size_t a_elems = 0, b_elems = 0, h_elems = 0;
    unsigned max_col_blocks = 0, max_row_blocks = 0;  // grid.y of the transposes
    double valu = 0, computed = 0;
    DeviceBuf d_groups, d_events, d_jobs, d_spairs, d_A, d_B, d_V, d_H;
};

static int strips_for(size_t m, int R) { return (int)((m + (size_t)64 * R - 1) / ((size_t)64 * R)); }

// ns per row step (8 instructions) per wavefront at two wavefronts per SIMD, and per step outside the rows (24 instructions = three row
// steps): the bench batch's 8192 jobs of 100 063 steps x (50 rows + 3) take their wave slots 4 x 95.2 ms (profiles/r06_runs/slice_variants.log);
// kChainPenalty: what a strip loses per strip of its group's chain (asleep behind the strip above, filling and draining the chain) -- 36 strips
// of 44 rows against 32 of 50 for 16 384 x 100 kbp: 789 against 760 ms where the plain count of instructions calls it a tie.
static constexpr double kNsPerRowStep = 17.9, kNsStepOverhead = 3 * 17.9, kChainPenalty = 0.0015;

int choose_rows_per_lane(const size_t* a_len, const size_t* b_len, size_t pairs, double simds, double* est_ns) {
    if (const char* e = getenv("PA_SLICE")) {
        if (atoi(e) == 0) return 0;
    }
    size_t live = 0;
    for (size_t i = 0; i < pairs; ++i) {
        live += a_len[i] > 0 && b_len[i] > 0;
        // (the boundary rows are addressed through a buffer descriptor of n * 8 bytes, the columns through 32-bit counters: pairs beyond
        //  2^27 columns stay with the strip kernels)
        if (a_len[i] >= (size_t(1) << 27) || b_len[i] >= (size_t(1) << 27)) return 0;
    }
    const bool forced = getenv("PA_SLICE") && atoi(getenv("PA_SLICE")) > 0;
    if (live < 64 && !forced) return 0;
    // groups of 32 in the order of the lengths; a group costs its longest a times the strips of its longest b
    std::vector<uint32_t> order;
    for (size_t i = 0; i < pairs; ++i)
        if (a_len[i] > 0 && b_len[i] > 0) order.push_back((uint32_t)i);
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return a_len[x] != a_len[y] ? a_len[x] < a_len[y] : (b_len[x] != b_len[y] ? b_len[x] < b_len[y] : x < y); });
    int best_r = 0;
    double best = -1;
    const double slots = simds * 2.0;
    for (const int R : kRowsPerLane) {
        if (forced && atoi(getenv("PA_SLICE")) > 1 && atoi(getenv("PA_SLICE")) != R) continue;
        double work = 0, longest = 0, jobs = 0;
        for (size_t g = 0; g * 32 < order.size(); ++g) {
            size_t n = 0, m = 0;
            for (size_t t = g * 32; t < std::min(order.size(), g * 32 + 32); ++t) {
                n = std::max(n, a_len[order[t]]);
                m = std::max(m, b_len[order[t]]);
            }
            const int S = strips_for(m, R);
            const double per_strip = ((double)n + 63.0) * (R * kNsPerRowStep + kNsStepOverhead) * (1.0 + kChainPenalty * (S - 1));
            work += per_strip * S;
            jobs += S;
            longest = std::max(longest, per_strip + 128.0 * (S - 1) * (R * kNsPerRowStep + kNsStepOverhead));  // the chain of a group's strips
        }
        // a wave slot runs one job at a time; a wavefront alone on its SIMD steps 1.165x as fast as one of two (18.3 against 21.3 ns per
        // row step); equal jobs finish in whole rounds of `slots`
        const double waves = std::min(jobs, slots);
        const double speed = waves <= simds ? 1.165 : 1.0 + 0.165 * (slots - waves) / simds;
        const double rounds = jobs > slots ? std::ceil(jobs / slots) * (work / jobs) : work / waves;
        const double t = std::max(longest / 1.165, rounds / speed);
        if (best < 0 || t < best) {
            best = t;
            best_r = R;
        }
    }
    if (est_ns) *est_ns = best;
    return best_r;
}

Plan* create(const size_t* a_len, const size_t* b_len, size_t pairs, const size_t* code_off, const size_t* prof_off, int rows_per_lane) {
    auto p = std::make_unique<Plan>();
    p->R = rows_per_lane;
    p->pairs = pairs;
    const int R = rows_per_lane;
    std::vector<uint32_t> order;
    for (size_t i = 0; i < pairs; ++i)
        if (a_len[i] > 0 && b_len[i] > 0) order.push_back((uint32_t)i);
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return a_len[x] != a_len[y] ? a_len[x] < a_len[y] : (b_len[x] != b_len[y] ? b_len[x] < b_len[y] : x < y); });
    std::vector<SlicePair> spairs(order.size());
    std::vector<SliceEvent> events;
    size_t a_at = 0, b_at = 0, h_at = 0;
    size_t max_n = 0, max_rows = 0;
    // the heaviest groups first: the launch ends with the short ones
    const size_t ngroups = (order.size() + 31) / 32;
    for (size_t gi = 0; gi < ngroups; ++gi) {
        const size_t g = ngroups - 1 - gi;  // (sorted ascending: the last group is the longest)
        SliceGroup grp{};
        grp.first_pos = (uint32_t)(g * 32);
        grp.npairs = (int32_t)std::min<size_t>(32, order.size() - g * 32);
        size_t n = 0, m = 0;
        for (int t = 0; t < grp.npairs; ++t) {
            const uint32_t i = order[g * 32 + t];
            n = std::max(n, a_len[i]);
            m = std::max(m, b_len[i]);
            spairs[g * 32 + t] = SlicePair{(uint64_t)code_off[i], (uint64_t)prof_off[i], (int32_t)a_len[i], (int32_t)b_len[i], i, 0u};
        }
        grp.n = (int32_t)n;
        grp.nstrips = strips_for(m, R);
        grp.a_off = a_at;
        grp.b_off = b_at;
        grp.h_off = h_at;
        grp.h_stride = (uint32_t)(n + 2 * kPad);
        a_at += n + 2 * kPad;
        b_at += (size_t)grp.nstrips * 64 * R;
        h_at += (size_t)(grp.nstrips - 1) * grp.h_stride;
        // capture events: one per distinct |a| of the group, increasing (the pairs are sorted by |a|)
        grp.ev_first = (uint32_t)events.size();
        for (int t = 0; t < grp.npairs; ++t) {
            const int32_t col = (int32_t)a_len[order[g * 32 + t]];
            if (events.size() > grp.ev_first && events.back().col == col) events.back().mask |= 1u << t;
            else events.push_back(SliceEvent{col, 1u << t});
        }
        grp.ev_count = (uint32_t)events.size() - grp.ev_first;
        max_n = std::max(max_n, n);
        max_rows = std::max(max_rows, (size_t)grp.nstrips * 64 * R);
        for (int s = 0; s < grp.nstrips; ++s) p->jobs.push_back(SliceJob{(uint32_t)p->groups.size(), (uint32_t)s});
        p->valu += (double)grp.nstrips * ((double)n + 63.0) * (8.0 * R + kStepOverheadInstr);
        p->computed += (double)grp.nstrips * 64.0 * R * (double)n * 32.0;
        p->groups.push_back(grp);
    }
    p->a_elems = a_at;
    p->b_elems = b_at;
    p->h_elems = h_at;
    p->max_col_blocks = (unsigned)((max_n + 127) / 128);
    p->max_row_blocks = (unsigned)((max_rows / 64 + 3) / 4);
    if (p->groups.empty()) return p.release();
    if (events.empty()) events.push_back(SliceEvent{0, 0});
    if (!p->d_groups.alloc(p->groups.size() * sizeof(SliceGroup)) || !p->d_events.alloc(events.size() * sizeof(SliceEvent)) ||
        !p->d_jobs.alloc(p->jobs.size() * sizeof(SliceJob)) || !p->d_spairs.alloc(spairs.size() * sizeof(SlicePair)) || !p->d_A.alloc(a_at * 8) ||
        !p->d_B.alloc(b_at * 8) || !p->d_V.alloc(b_at * 8) || !p->d_H.alloc(std::max<size_t>(h_at, 8) * 8))
        return nullptr;
    if (!hip_ok(hipMemcpy(p->d_groups.ptr, p->groups.data(), p->groups.size() * sizeof(SliceGroup), hipMemcpyHostToDevice), "H2D slice groups") ||
        !hip_ok(hipMemcpy(p->d_events.ptr, events.data(), events.size() * sizeof(SliceEvent), hipMemcpyHostToDevice), "H2D slice events") ||
        !hip_ok(hipMemcpy(p->d_jobs.ptr, p->jobs.data(), p->jobs.size() * sizeof(SliceJob), hipMemcpyHostToDevice), "H2D slice jobs") ||
        !hip_ok(hipMemcpy(p->d_spairs.ptr, spairs.data(), spairs.size() * sizeof(SlicePair), hipMemcpyHostToDevice), "H2D slice pairs") ||
        // the pads of the column planes are read (by lanes that are not at a column yet) and never used; they are written once all the same
        !hip_ok(hipMemset(p->d_A.ptr, 0, a_at * 8), "memset slice A"))
        return nullptr;
    return p.release();
}

void destroy(Plan* p) { delete p; }

template <int R>
static hipError_t launch_slice(int grid, hipStream_t s, const Plan* p, uint32_t* d_ticket_err, unsigned long long* dbg) {
    hipLaunchKernelGGL((slice_kernel<R>), dim3((unsigned)grid), dim3(64), 0, s, p->d_jobs.as<SliceJob>(), (int)p->jobs.size(), p->d_groups.as<SliceGroup>(),
                       p->d_events.as<SliceEvent>(), p->d_A.as<uint2>(), p->d_B.as<uint2>(), p->d_H.as<uint2>(), p->d_V.as<uint2>(), d_ticket_err, dbg);
    return hipGetLastError();
}

int run(Plan* p, hipStream_t s, const uint32_t* d_codes, const uint64_t* d_prof, int32_t* d_costs, uint32_t* d_ticket_err, hipEvent_t ev0, hipEvent_t ev1) {
    if (!hip_ok(hipMemsetAsync(d_ticket_err, 0, 8, s), "memset slice ticket")) return PA_E_HIP;
    if (p->groups.empty()) {
        if (ev0 && (!hip_ok(hipEventRecord(ev0, s), "event") || !hip_ok(hipEventRecord(ev1, s), "event"))) return PA_E_HIP;
        return 0;
    }
    const unsigned G = (unsigned)p->groups.size();
    hipLaunchKernelGGL(slice_pack_a_kernel, dim3(G, p->max_col_blocks), dim3(256), 0, s, p->d_groups.as<SliceGroup>(), p->d_spairs.as<SlicePair>(), d_codes,
                       p->d_A.as<uint2>());
    hipLaunchKernelGGL(slice_pack_b_kernel, dim3(G, p->max_row_blocks), dim3(256), 0, s, p->d_groups.as<SliceGroup>(), p->d_spairs.as<SlicePair>(), d_prof,
                       p->d_B.as<uint2>(), 64 * p->R);
    if (!hip_ok(hipGetLastError(), "slice transposes")) return PA_E_HIP;
    if (!hip_ok(hipMemsetAsync(p->d_V.ptr, 0, p->b_elems * 8, s), "memset slice V")) return PA_E_HIP;  // captured columns are OR-ed in
    // boundary rows: "not written yet" = hp = hm = ~0 in every pair
    if (p->h_elems && !hip_ok(hipMemsetAsync(p->d_H.ptr, 0xFF, p->h_elems * 8, s), "memset slice boundaries")) return PA_E_HIP;
    int cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    const int grid = (int)std::min<size_t>(p->jobs.size(), (size_t)cus * 8);  // two wavefronts per SIMD, one wavefront per workgroup
    // diagnostics: PA_SLICE_JOBTIMES=1 prints, per pass, how long the (group, strip) jobs took their wavefronts and how much of that they slept
    static const bool jobtimes = getenv("PA_SLICE_JOBTIMES") != nullptr;
    unsigned long long* dbg = nullptr;
    if (jobtimes && hipMalloc((void**)&dbg, 192) == hipSuccess) {
        const unsigned long long init[24] = {~0ull, 0, 0, 0, 0, 0, ~0ull, 0};
        if (hipMemcpyAsync(dbg, init, 192, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) dbg = nullptr;
    }
    if (ev0 && !hip_ok(hipEventRecord(ev0, s), "event")) return PA_E_HIP;
    hipError_t e = hipSuccess;
    switch (p->R) {
        case 52: e = launch_slice<52>(grid, s, p, d_ticket_err, dbg); break;
        case 50: e = launch_slice<50>(grid, s, p, d_ticket_err, dbg); break;
        case 48: e = launch_slice<48>(grid, s, p, d_ticket_err, dbg); break;
        case 46: e = launch_slice<46>(grid, s, p, d_ticket_err, dbg); break;
        case 44: e = launch_slice<44>(grid, s, p, d_ticket_err, dbg); break;
        case 42: e = launch_slice<42>(grid, s, p, d_ticket_err, dbg); break;
        case 40: e = launch_slice<40>(grid, s, p, d_ticket_err, dbg); break;
        case 36: e = launch_slice<36>(grid, s, p, d_ticket_err, dbg); break;
        case 32: e = launch_slice<32>(grid, s, p, d_ticket_err, dbg); break;
        case 28: e = launch_slice<28>(grid, s, p, d_ticket_err, dbg); break;
        default: set_error("slice: no kernel for %d rows per lane", p->R); return PA_E_INTERNAL;
    }
    if (!hip_ok(e, "slice_kernel")) return PA_E_HIP;
    if (ev1 && !hip_ok(hipEventRecord(ev1, s), "event")) return PA_E_HIP;
    if (dbg) {
        unsigned long long h[24] = {0};
        if (hipMemcpyAsync(h, dbg, 192, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess && h[3]) {
            std::fprintf(stderr, "[slice jobs] awake us per job by XCD:");
            for (int x = 0; x < 8; ++x) std::fprintf(stderr, " %d: %.0f (%llu)", x, h[9 + 2 * x] ? h[8 + 2 * x] * 0.01 / (double)h[9 + 2 * x] : 0.0, h[9 + 2 * x]);
            std::fprintf(stderr, "\n");
            std::fprintf(stderr, "[slice jobs] R %d: %llu jobs, us per job: min %.1f mean %.1f max %.1f (first strips: min %.1f max %.1f); asleep behind the strip above: %.2f %% of the job time, %.1f waits per job\n",
                         p->R, h[3], h[0] * 0.01, h[2] * 0.01 / (double)h[3], h[1] * 0.01, h[6] * 0.01, h[7] * 0.01, 100.0 * (double)h[4] / (double)h[2], (double)h[5] / (double)h[3]);
        }
        (void)hipFree(dbg);
    }
    hipLaunchKernelGGL(slice_score_kernel, dim3(G, getenv("PA_SCORE_ONE") ? 1u : (p->max_row_blocks * 256u + kScoreSpan - 1) / kScoreSpan), dim3(64), 0, s, p->d_groups.as<SliceGroup>(), p->d_spairs.as<SlicePair>(), p->d_V.as<uint2>(), d_costs);
    if (!hip_ok(hipGetLastError(), "slice_score_kernel")) return PA_E_HIP;
    return 0;
}

Info info(const Plan* p) {
    Info i{};
    i.rows_per_lane = p->R;
    i.groups = p->groups.size();
    i.jobs = p->jobs.size();
    i.valu_instructions = p->valu;
    i.computed_rows_cells = p->computed;
    i.device_bytes = (double)(p->a_elems + 2 * p->b_elems + p->h_elems) * 8.0;
    i.boundary_bytes = (double)p->h_elems * 8.0;
    return i;
}

}  // namespace slice
}  // namespace pa
