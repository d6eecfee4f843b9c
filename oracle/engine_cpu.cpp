/*
 * oracle/engine_cpu.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Instantiates the host block engine (astar-pairwise-aligner_amd/csrc/engine.hpp) over the oracle's
 * CPU kernels (pa_oracle.c).  Used by tests to (1) check the host logic without a GPU against the
 * reference's own acceptance rules (cost == Levenshtein, CIGAR valid, incremental == from scratch;
 * pa-test/src/lib.rs:65-99, blocks.rs:471-543) and (2) supply the expected cost / CIGAR / band
 * statistics that the HIP-backed engine must reproduce exactly (the kernels are bit-exact, so every
 * band decision and the traceback must coincide).  The shipped library never links this file.
 *
 * Note: the engine template is shared with the product, so this is an oracle for the *kernels under
 * the engine*, not an independent restatement of the host logic; the host logic is pinned by the
 * acceptance rules above.  Exact A*PA2 CIGAR strings remain "parity unpinned" (no Rust toolchain).
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../astar-pairwise-aligner_amd/csrc/engine_capi.hpp"
#include "pa_oracle.h"

using namespace pa::engine;

namespace {

struct CpuBackend {
    std::vector<uint8_t> a_, b_;
    std::vector<pa_bits_t> pa_, pb_;
    std::vector<pa_h_t> h_;
    bool ok = true;

    CpuBackend(const uint8_t* a, size_t n, const uint8_t* b, size_t m) : a_(a, a + n), b_(b, b + m) {
        pa_.resize(n ? n : 1);
        pb_.resize((m + 63) / 64 ? (m + 63) / 64 : 1);
        ok = pa_or_bitprofile_build(a, n, b, m, pa_.data(), pb_.data()) == 0;
    }
    I n() const { return (I)a_.size(); }
    I m() const { return (I)b_.size(); }
    const uint8_t* a() const { return a_.data(); }
    const uint8_t* b() const { return b_.data(); }
    void enable_h_row() { h_.assign(a_.size(), pa_h_t{0, 0}); }  // blocks.rs:119-123

    Cost run(I i0, I i1, size_t w0, size_t w1, V* v, pa_h_t* h, bool exact, const BlockParams& p) {
        const size_t n = (size_t)(i1 - i0), w = w1 - w0;
        pa_v_t* vv = reinterpret_cast<pa_v_t*>(v);
        if (!p.simd) return pa_or_scalar_row(pa_.data() + i0, n, pb_.data() + w0, w, h, vv);
        return pa_or_simd_compute(pa_.data() + i0, n, pb_.data() + w0, w, h, vv, exact, p.no_ilp ? 1 : 2);
    }

    Cost compute(I i0, I i1, size_t w0, size_t w1, V* v, HMode mode, const BlockParams& p) {  // blocks.rs:728-747
        const size_t n = (size_t)(i1 - i0);
        switch (mode) {
            case HMode::None: {
                std::vector<pa_h_t> h(n, pa_h_t{1, 0});
                return run(i0, i1, w0, w1, v, h.data(), false, p);
            }
            case HMode::Input: {
                std::vector<pa_h_t> h(h_.begin() + i0, h_.begin() + i1);
                return run(i0, i1, w0, w1, v, h.data(), false, p);
            }
            case HMode::Update:
                return run(i0, i1, w0, w1, v, h_.data() + i0, true, p);
            case HMode::Output:
                for (I i = i0; i < i1; ++i) h_[i] = pa_h_t{1, 0};
                return run(i0, i1, w0, w1, v, h_.data() + i0, true, p);
        }
        return 0;
    }

    void fill(I i0, I i1, size_t w0, size_t w1, V* v, V* values, int8_t* hbot, const BlockParams& p) {  // blocks.rs:627-648
        const size_t n = (size_t)(i1 - i0), w = w1 - w0;
        std::vector<pa_h_t> h(n, pa_h_t{1, 0});
        pa_v_t* vv = reinterpret_cast<pa_v_t*>(v);
        pa_v_t* vals = reinterpret_cast<pa_v_t*>(values);
        if (p.simd) pa_or_simd_fill(pa_.data() + i0, n, pb_.data() + w0, w, h.data(), vv, vals);
        else pa_or_scalar_fill(pa_.data() + i0, n, pb_.data() + w0, w, h.data(), vv, vals);
        for (size_t i = 0; i < n; ++i) hbot[i] = (int8_t)((int)h[i].p - (int)h[i].m);
    }

    std::vector<int8_t> debug_read_h(I i0, I i1) {
        std::vector<int8_t> r;
        for (I i = i0; i < i1; ++i) r.push_back((int8_t)((int)h_[i].p - (int)h_[i].m));
        return r;
    }
    void debug_write_h(I i0, I i1, const std::vector<int8_t>& x) {
        for (I i = i0; i < i1; ++i) h_[i] = pa_h_t{(uint64_t)(x[i - i0] > 0), (uint64_t)(x[i - i0] < 0)};
    }
};

}  // namespace

extern "C" int pa_cpu_align(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len,
                            const pa_astarpa2_params* params, int trace, int self_check, int32_t* cost_out,
                            char** cigar_out, pa_astarpa2_stats* stats_out) {
    if (!params || !params_valid(*params)) return -4;
    CpuBackend be(a, a_len, b, b_len);
    if (!be.ok) return -1;
    const AstarPa2Params p = params_from_c(*params);
    AlignResult r;
    try {
        r = cost_or_align(p, be, trace != 0, self_check != 0);
    } catch (const EnginePanic& e) {
        std::fprintf(stderr, "astarpa2 engine panic: %s\n", e.what());
        return -5;
    }
    if (cost_out) *cost_out = r.cost;
    if (cigar_out) {
        *cigar_out = nullptr;
        if (r.has_cigar) {
            const std::string s = r.cigar.to_string();
            *cigar_out = (char*)std::malloc(s.size() + 1);
            std::memcpy(*cigar_out, s.c_str(), s.size() + 1);
        }
    }
    if (stats_out) stats_to_c(r.stats, stats_out);
    return 0;
}

extern "C" void pa_cpu_free(char* p) { std::free(p); }

// Test hook: the SH heuristic values h(i, *) for i = 0..n (engine.hpp SeedHeuristicH).
extern "C" void pa_cpu_sh_h(const uint8_t* a, size_t n, const uint8_t* b, size_t m, int k, int32_t* out) {
    SeedHeuristicH h(a, (I)n, b, (I)m, (I)k);
    for (size_t i = 0; i <= n; ++i) out[i] = h.h((I)i, 0);
}

// Test hook: GCSH h(i, j) at the given positions (before any pruning), plus the kept matches.
extern "C" int pa_cpu_gcsh_probe(const uint8_t* a, size_t n, const uint8_t* b, size_t m, int k, int p, const int32_t* qi,
                                 const int32_t* qj, size_t nq, int32_t* h_out, int32_t* match_out, size_t match_cap) {
    GcshHeuristic h(a, (I)n, b, (I)m, (I)k, p, true);
    for (size_t t = 0; t < nq; ++t) h_out[t] = h.h(qi[t], qj[t]);
    size_t cnt = 0;
    for (const auto& mt : h.by_start) {
        if (cnt < match_cap) {
            match_out[2 * cnt] = mt.i;
            match_out[2 * cnt + 1] = mt.j;
        }
        cnt++;
    }
    return (int)cnt;
}
