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

#include "cpu_backend.hpp"

using namespace pa::engine;

using pa_oracle_cpu::CpuBackend;

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

// Test hook: the blocks of the LAST completed pass (before the traceback) as flat arrays, so that tests can check the band
// logic against a dense DP without going through the engine's own accessors.  rec[k] = {i0, i1, ojs, oje, js, je, fs, fe,
// top_val, bot_val, v_offset, v_words}; v = concatenated V words (p, m).  Returns the number of blocks, or < 0.
extern "C" int pa_cpu_align_blocks(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, const pa_astarpa2_params* params,
                                   int32_t* cost_out, int32_t* f_max_out, int32_t* rec, size_t rec_cap, uint64_t* v, size_t v_cap) {
    if (!params || !params_valid(*params)) return -4;
    CpuBackend be(a, a_len, b, b_len);
    if (!be.ok) return -1;
    const AstarPa2Params p = params_from_c(*params);
    int nblocks = 0;
    size_t vused = 0;
    bool overflow = false;
    try {
        const AlignResult r = cost_or_align<CpuBackend>(p, be, true, false, [&](const Blocks<CpuBackend>& bl, std::optional<Cost> f_max, Cost) {
            nblocks = 0;
            vused = 0;
            if (f_max_out) *f_max_out = f_max.value_or(-1);
            for (size_t k = 0; k <= bl.last_block_idx; ++k) {
                const Block& x = bl.blocks[k];
                if ((size_t)(nblocks + 1) * 12 > rec_cap || vused + 2 * x.v.size() > v_cap) {
                    overflow = true;
                    return;
                }
                int32_t* o = rec + (size_t)nblocks * 12;
                o[0] = x.i_range.s;
                o[1] = x.i_range.e;
                o[2] = x.original_j_range.s;
                o[3] = x.original_j_range.e;
                o[4] = x.j_range.s;
                o[5] = x.j_range.e;
                o[6] = x.fixed_j_range ? x.fixed_j_range->s : -1;
                o[7] = x.fixed_j_range ? x.fixed_j_range->e : -2;
                o[8] = x.top_val;
                o[9] = x.bot_val;
                o[10] = (int32_t)(vused / 2);
                o[11] = (int32_t)x.v.size();
                for (const V& w : x.v) {
                    v[vused++] = w.p;
                    v[vused++] = w.m;
                }
                nblocks += 1;
            }
        });
        if (cost_out) *cost_out = r.cost;
    } catch (const EnginePanic& e) {
        std::fprintf(stderr, "astarpa2 engine panic: %s\n", e.what());
        return -5;
    }
    return overflow ? -6 : nblocks;
}

// ---- the CPU baselines on every host core (bench.py `cpu_baseline_nproc`) ------------------------------------------------
// Independent pairs pulled from one atomic counter by `nthreads` std::threads: no Python between the calls, so the number is
// what the host's cores do with the oracle kernels, not what a thread pool of an interpreter lets through.
#include <atomic>
#include <thread>

// mode 0: full DP, cost only (pa_or_nw_cost, the AVX2 strip port); mode 1: A*PA2 with `params` and traceback (the CPU-kernel engine).
// costs[i] receives the cost of pair i.  Returns 0, or the first error code.
extern "C" int pa_cpu_many(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len, size_t pairs,
                           const pa_astarpa2_params* params, int mode, int nthreads, int32_t* costs) {
    if (nthreads < 1) nthreads = 1;
    if (mode == 1 && (!params || !params_valid(*params))) return -4;
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= pairs || err.load(std::memory_order_relaxed)) return;
            if (mode == 0) {
                costs[i] = pa_or_nw_cost(a[i], a_len[i], b[i], b_len[i], 1);
            } else {
                int32_t c = 0;
                char* cg = nullptr;
                const int rc = pa_cpu_align(a[i], a_len[i], b[i], b_len[i], params, 1, 0, &c, &cg, nullptr);
                std::free(cg);
                if (rc != 0) err.store(rc, std::memory_order_relaxed);
                costs[i] = c;
            }
        }
    };
    std::vector<std::thread> ts;
    for (int t = 1; t < nthreads; ++t) ts.emplace_back(work);
    work();
    for (auto& t : ts) t.join();
    return err.load();
}
