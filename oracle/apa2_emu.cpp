/*
 * oracle/apa2_emu.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Runs the product's per-pair A*PA2 program (astar-pairwise-aligner_amd/csrc/apa2_logic.hpp: the band search one wavefront
 * executes per pair in the batched mode) WITHOUT a GPU: the backend below computes the blocks with the oracle's CPU kernels
 * and runs the reference's LITERAL probing loops for fixed_j_range (domain.rs:306-328), so the CPU test-suite checks (a) the
 * restated pass / search logic against the host engine (cost, every statistic) and (b) the claim that the probing loops end on
 * the first / last row with f <= f_max, which the device backend's wave-parallel scans rely on.  The traceback then runs
 * engine.hpp's Blocks::trace over the blocks the program left behind -- what trace_kernel.hpp reads on the device.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../astar-pairwise-aligner_amd/csrc/apa2_logic.hpp"
#include "cpu_backend.hpp"

using namespace pa::engine;
using namespace pa::apa2;
using pa_oracle_cpu::CpuBackend;

namespace {

struct EmuBackend {
    CpuBackend& cb;
    pa::sweep::HeurParams hp;
    int sparse_h;
    size_t wtot;
    std::vector<BlockRec> rec;
    std::vector<std::vector<V>> col;  // slot k, absolute words
    uint64_t scans = 0, scan_mismatch = 0;
    BlockParams bp;

    EmuBackend(CpuBackend& c, const pa::sweep::HeurParams& h, int sh, int nblk) : cb(c), hp(h), sparse_h(sh) {
        wtot = (size_t)(c.m() + 63) / 64;
        rec.resize((size_t)nblk + 2);
        col.assign((size_t)nblk + 2, std::vector<V>(wtot, V::one()));
        bp.simd = true;
        bp.no_ilp = false;
    }
    bool failed() const { return false; }
    void mark(int, uint32_t) const {}
    int32_t uniform(int32_t x) const { return x; }
    uint64_t strip_instructions() const { return 0; }
    BlockRec load_rec(int32_t k) const { return rec[(size_t)k]; }
    void store_rec(int32_t k, const BlockRec& r) { rec[(size_t)k] = r; }
    int32_t index(int32_t k, const BlockRec& r, int32_t j) const {  // block.rs:69-122 (from the top)
        if (k == 0) return j;
        if (j > r.je) return r.bot_val + (j - r.je);
        int32_t v = r.top_val, j0 = r.js;
        while (j0 + 64 <= j) {
            v += col[(size_t)k][(size_t)j0 / 64].value();
            j0 += 64;
        }
        if (j > j0) v += col[(size_t)k][(size_t)j0 / 64].value_of_prefix(j - j0);
        return v;
    }
    int32_t compute(int32_t k, const BlockRec& prev, const BlockRec& cur, int32_t i0, int32_t i1) {
        const size_t w0 = (size_t)cur.js / 64, w1 = (size_t)cur.je / 64;
        for (size_t w = w0; w < w1; ++w) {
            const bool in_prev = k > 1 && (int32_t)(w * 64) >= prev.js && (int32_t)(w * 64) < prev.je;
            col[(size_t)k][w] = in_prev ? col[(size_t)k - 1][w] : V::one();
        }
        if (w1 == w0) return i1 - i0;
        return cb.compute(i0, i1, w0, w1, col[(size_t)k].data() + w0, HMode::None, bp);
    }
    int32_t f(int32_t k, const BlockRec& r, int32_t i, int32_t j) const { return index(k, r, j) + pa::sweep::heur_h(hp, i, j); }
    // the literal loops of domain.rs:306-328 (engine.hpp fixed_j_range), cross-checked against plain first / last searches
    bool scan_first(int32_t k, const BlockRec& r, int32_t i, int32_t f_max, int32_t lo, int32_t hi, int32_t* out) {
        int32_t start = lo;
        while (start <= hi) {
            const int32_t fv = f(k, r, i, start);
            if (fv <= f_max) break;
            start += sparse_h ? pa::sweep::div_ceil_pos(fv - f_max, 2) : 1;
        }
        int32_t plain = lo;
        while (plain <= hi && f(k, r, i, plain) > f_max) ++plain;
        scans += 1;
        if ((start <= hi) != (plain <= hi) || (start <= hi && start != plain)) scan_mismatch += 1;
        *out = start;
        return start <= hi;
    }
    bool scan_last(int32_t k, const BlockRec& r, int32_t i, int32_t f_max, int32_t lo, int32_t hi, int32_t* out) {
        int32_t end = hi;
        while (end >= lo) {
            const int32_t fv = f(k, r, i, end);
            if (fv <= f_max) break;
            end -= sparse_h ? pa::sweep::div_ceil_pos(fv - f_max, 2) : 1;
        }
        int32_t plain = hi;
        while (plain >= lo && f(k, r, i, plain) > f_max) --plain;
        scans += 1;
        if ((end >= lo) != (plain >= lo) || (end >= lo && end != plain)) scan_mismatch += 1;
        *out = end;
        return end >= lo;
    }
};

}  // namespace

// The band-proportional slots of the batch kernels' column store (csrc/sweep_logic.hpp SlotGeom): the addressing the GPU kernels and the
// host share, exposed for tests/test_host_logic_slots.py.
extern "C" int32_t pa_emu_slot_off(int32_t n, int32_t m, int32_t win, uint32_t ratio, int32_t k) {
    return pa::sweep::slot_off(pa::sweep::SlotGeom{n, m, win, ratio}, k);
}
extern "C" int pa_emu_slot_holds(int32_t n, int32_t m, int32_t win, uint32_t ratio, int32_t k, int32_t w0, int32_t w1) {
    return pa::sweep::slot_holds(pa::sweep::SlotGeom{n, m, win, ratio}, k, w0, w1) ? 1 : 0;
}

// rc 0 = ran; 1 = parameters not supported by the batched program; 2 = the program handed the pair back (info[0] = status).
// info[1] = scans run, info[2] = scans whose jumping probes did NOT end on the first / last row with f <= f_max.
extern "C" int pa_apa2_emu_align(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, const pa_astarpa2_params* params,
                                 int32_t* cost_out, char** cigar_out, pa_astarpa2_stats* stats_out, int32_t* info) {
    if (!params || !params_valid(*params)) return -4;
    const AstarPa2Params p = params_from_c(*params);
    const bool ok = p.domain == DomainKind::Astar &&
                    (p.heuristic == HeuristicKind::None || p.heuristic == HeuristicKind::Gap || p.heuristic == HeuristicKind::SH) &&
                    p.block_width == 256 && p.front.sparse && !p.front.incremental_doubling && !p.prune &&
                    (p.doubling == DoublingKind::BandDoubling || p.doubling == DoublingKind::LinearSearch);
    if (!ok || a_len == 0 || b_len == 0) return 1;
    CpuBackend cb(a, a_len, b, b_len);
    if (!cb.ok) return -1;
    std::vector<int32_t> sh;
    pa::sweep::HeurParams hp;
    hp.kind = p.heuristic == HeuristicKind::Gap ? pa::sweep::kHeurGap : p.heuristic == HeuristicKind::SH ? pa::sweep::kHeurSH : pa::sweep::kHeurNone;
    hp.n = (int32_t)a_len;
    hp.m = (int32_t)b_len;
    hp.sh_h = nullptr;
    if (hp.kind == pa::sweep::kHeurSH) {
        SeedHeuristicH h(a, (I)a_len, b, (I)b_len, p.heuristic_k);
        sh.assign(h.h_by_i.begin(), h.h_by_i.end());
        hp.sh_h = sh.data();
    }
    SearchParams sp;
    sp.heur = hp.kind;
    sp.sparse_h = p.sparse_h ? 1 : 0;
    sp.doubling = p.doubling == DoublingKind::LinearSearch ? kDoublingLinear : kDoublingBand;
    sp.start = (int32_t)p.start;
    sp.factor = p.factor;
    sp.delta = (int32_t)p.delta;
    const int nblk = ((int)a_len + 255) / 256;
    EmuBackend be(cb, hp, sp.sparse_h, nblk);
    PairProg<EmuBackend> prog(be, hp, sp);
    PairResult res;
    prog.run(&res);
    if (info) {
        info[0] = res.status;
        info[1] = (int32_t)be.scans;
        info[2] = (int32_t)be.scan_mismatch;
        info[3] = res.f_max;
    }
    if (res.status != kOk) return 2;
    AstarPa2Stats st;
    if (sp.doubling == kDoublingBand) {  // lib.rs:158: the other arms of cost_or_align leave the block counters at zero
        st.block_stats.num_blocks = res.num_blocks;
        st.block_stats.num_incremental_blocks = res.num_incremental_blocks;
        st.block_stats.computed_lanes = res.computed_lanes;
        st.block_stats.unique_lanes = res.unique_lanes;
    }
    st.f_max_tries = res.f_max_tries;
    st.sanity_violations = res.sanity_violations;
    std::string cig;
    try {  // Blocks::trace over what the program left behind (what the device traceback reads)
        Blocks<CpuBackend> blocks(p.front, true, cb);
        blocks.blocks.resize((size_t)nblk + 1);
        const BlockRec& r0 = be.rec[0];
        blocks.blocks[0] = Block::first_col(JRange{r0.ojs, r0.oje}, JRange{r0.js, r0.je});
        for (int k = 1; k <= nblk; ++k) {
            const BlockRec& r = be.rec[(size_t)k];
            Block& x = blocks.blocks[(size_t)k];
            x.i_range = IRange{(k - 1) * 256, k * 256 < (int)a_len ? k * 256 : (int)a_len};
            x.original_j_range = JRange{r.ojs, r.oje};
            x.j_range = JRange{r.js, r.je};
            x.fixed_j_range = JRange{r.fs, r.fe};
            x.offset = r.js;
            x.top_val = r.top_val;
            x.bot_val = r.bot_val;
            x.v.assign(be.col[(size_t)k].begin() + r.js / 64, be.col[(size_t)k].begin() + r.je / 64);
        }
        blocks.last_block_idx = (size_t)nblk;
        blocks.i_range = IRange{-1, (I)a_len};
        auto [cg, ts] = blocks.trace(0, 0, (I)a_len, (I)b_len);
        st.trace_stats = ts;
        cig = cg.to_string();
    } catch (const EnginePanic& e) {
        std::fprintf(stderr, "apa2 emu: engine panic in trace: %s\n", e.what());
        return -5;
    }
    if (cost_out) *cost_out = res.cost;
    if (cigar_out) {
        *cigar_out = (char*)std::malloc(cig.size() + 1);
        std::memcpy(*cigar_out, cig.c_str(), cig.size() + 1);
    }
    if (stats_out) stats_to_c(st, stats_out);
    return 0;
}
