/*
 * oracle/cpu_backend.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * The kernel backend of the host block engine over the oracle's CPU kernels (pa_oracle.c); shared by engine_cpu.cpp and
 * sweep_emu.cpp.
 */
#pragma once
#include <vector>

#include "../astar-pairwise-aligner_amd/csrc/engine_capi.hpp"
#include "pa_oracle.h"

namespace pa_oracle_cpu {
using namespace pa::engine;

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

}  // namespace pa_oracle_cpu
