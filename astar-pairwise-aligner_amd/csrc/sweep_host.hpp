// sweep_host.hpp -- host side of the device-resident A*PA2 sweep: AstarPa2::cost_or_align (astarpa2/src/lib.rs:122-175)
// with every align_for_bounded_dist pass (domain.rs:356-541) executed as ONE launch of the sweep kernel (sweep_wave.hpp).
//
// The host keeps what the reference keeps between passes (the first column, `last_block_idx`, the statistics), decides the
// first column and block 1 (they need no DP), launches the pass, commits the pass's block records and runs the band search
// (band.rs:100-182).  With traceback on, the blocks of the successful pass are read back and handed to the engine's own
// Blocks::trace (engine.hpp), so DT-trace / re-fill / parent are the same code as on the host-driven path.
//
// Templated over a Launcher (device memory + one pass): the HIP launcher lives in engine_hip.hip, tests/tools/sweep_emu has
// one that runs the same wave program on host threads.
#pragma once
#include <cstring>
#include <optional>
#include <stdexcept>
#include <vector>

#include "engine.hpp"
#include "sweep_wave.hpp"

namespace pa {
namespace sweep {

struct SweepFallback : std::runtime_error {  // the pass met a case the kernel does not handle: use the host-driven engine
    int reason;
    SweepFallback(const char* what, int r) : std::runtime_error(what), reason(r) {}
};

inline bool sweep_supported(const engine::AstarPa2Params& p, size_t n, size_t m) {
    using namespace engine;
    return n > 0 && m > 0 && n < (1u << 30) && m < (1u << 30) && p.domain == DomainKind::Astar &&
           (p.heuristic == HeuristicKind::None || p.heuristic == HeuristicKind::Gap || p.heuristic == HeuristicKind::SH) &&
           p.block_width == kBlockW && p.front.sparse && !p.front.incremental_doubling && !p.prune &&
           (p.doubling == DoublingKind::BandDoubling || p.doubling == DoublingKind::LinearSearch);
}

// Sizes of one pass's buffers for a given f_max (everything is a window around the diagonal: a cell with f <= f_max has
// |i - j| <= g <= f_max, and the range estimate of domain.rs:160-235 adds at most f_max + 256 below the fixed end).
struct PassGeometry {
    int32_t nblk, wtot, nstrips, win;
    int64_t gran_stride, col_stride, pr_stride;
    int32_t col_ring;  // blocks in the column ring of a cost-only pass (power of two)
};
inline PassGeometry pass_geometry(int32_t n, int32_t m, int32_t f_max) {
    PassGeometry g;
    g.nblk = (n + kBlockW - 1) / kBlockW;
    g.wtot = (m + 63) / 64;
    g.nstrips = (g.wtot * 64 + kStripRows - 1) / kStripRows;
    const int64_t lim = (int64_t)(n > m ? n : m) + 1024;
    int64_t win = 2ll * f_max + 1024;
    if (win > lim) win = lim;
    g.win = (int32_t)win;
    int64_t gs = (2 * win) / 32 + 16;
    if (gs > n / 32 + 16) gs = n / 32 + 16;
    g.gran_stride = gs;
    int64_t cs = (2 * win + kBlockW) / 64 + 8;
    if (cs > g.wtot + 8) cs = g.wtot + 8;
    g.col_stride = cs;
    int64_t ps = (2 * win + 2 * kStripRows) / kBlockW + 8;
    if (ps > g.nblk + 2) ps = g.nblk + 2;
    g.pr_stride = ps;
    int32_t ring = 1;
    while (ring < g.nblk + 2 && ring < 1024) ring *= 2;
    g.col_ring = ring;
    return g;
}

// What the host writes before a pass (one small upload / kernel argument).
struct PassInit {
    int32_t js1, je1, ojs1, oje1, flags1;  // block 1's range (BRec[1])
    int32_t top1, fs0;                     // block 1's top-edge record (TRec[1])
    int32_t last_strip;                    // strips 0..last_strip start at block 1
};

template <class Backend, class Launcher>
class SweepAligner {
   public:
    using I = engine::I;
    using Cost = engine::Cost;
    const engine::AstarPa2Params& params;
    Backend& be;
    Launcher& dev;
    HeurParams hp;
    std::vector<int32_t> sh_h;  // host copy of the SH table
    engine::AstarPa2Stats stats;
    bool trace;
    int32_t n, m, nblk;
    int32_t last_block_idx = 0;  // Blocks::last_block_idx as the previous pass left it (domain.rs:395-404 reads through it)
    BlockRec rec0;               // the first column as the previous pass left it
    bool have_rec0 = false;
    int64_t blocks_len = 0;      // Blocks::blocks.len()
    std::optional<engine::Cigar> cigar;

    SweepAligner(const engine::AstarPa2Params& p, Backend& backend, Launcher& launcher, bool trace_)
        : params(p), be(backend), dev(launcher), trace(trace_) {
        n = be.n();
        m = be.m();
        nblk = (n + kBlockW - 1) / kBlockW;
        const double t0 = engine::now_s();
        hp.kind = p.heuristic == engine::HeuristicKind::Gap ? kHeurGap : p.heuristic == engine::HeuristicKind::SH ? kHeurSH : kHeurNone;
        hp.n = n;
        hp.m = m;
        hp.sh_h = nullptr;
        if (hp.kind == kHeurSH) {
            engine::SeedHeuristicH sh(be.a(), n, be.b(), m, p.heuristic_k);
            sh_h.assign(sh.h_by_i.begin(), sh.h_by_i.end());
            hp.sh_h = sh_h.data();
        }
        stats.t_precomp = engine::now_s() - t0;
        dev.begin_pair(n, m, nblk, hp.kind == kHeurSH ? sh_h.data() : nullptr, trace);
    }

    Cost h0() const { return heur_h(hp, 0, 0); }

    // One pass.  nullopt = no path for this bound (domain.rs returns None).
    std::optional<std::pair<Cost, std::optional<engine::Cigar>>> pass(Cost f_max) {
        stats.f_max_tries += 1;
        if (f_max < 0) engine::engine_panic("f_max >= 0");
        // ---- first column (domain.rs:395-413, blocks.rs:146-179) ----
        BlockRec stale;  // `blocks.next_block_j_range()` before init(): the block after the previous pass's last one
        stale.js = kNone;
        if ((int64_t)last_block_idx + 1 < blocks_len) stale = dev.read_old(last_block_idx + 1);
        JRangeOut jr0;
        if (!next_j_range(hp, -1, 0, -1, -1, 0, f_max, params.sparse_h ? 1 : 0, stale.js, stale.je, &jr0) || jr0.ojs > 0) return std::nullopt;
        BlockRec r0;
        r0.ojs = jr0.ojs;
        r0.oje = jr0.oje;
        r0.js = 0;
        r0.je = jr0.je;
        if (have_rec0 && rec0.je > r0.je) r0.je = rec0.je;  // initial_j_range.union(blocks[0].j_range), rounded
        r0.fs = jr0.ojs;
        r0.fe = jr0.oje;
        r0.top_val = 0;
        r0.bot_val = r0.je;
        rec0 = r0;
        have_rec0 = true;
        if (blocks_len < 1) blocks_len = 1;
        last_block_idx = 0;
        dev.write_old(0, r0);
        // ---- block 1 from the first column: index_0(j) = j ----
        BlockRec old1;
        old1.js = kNone;
        if (blocks_len > 1) old1 = dev.read_old(1);
        const NextDecision nd = decide_next(hp, f_max, params.sparse_h ? 1 : 0, 0, n < kBlockW ? n : kBlockW, r0.fs, r0.fe, r0.fe, old1, true);
        if (!nd.ok) {
            if (old1.js != kNone) engine::engine_panic("empty j_range with existing next block");
            return std::nullopt;
        }
        stats.block_stats.num_blocks += nd.d_num_blocks;
        stats.block_stats.unique_lanes += nd.d_unique_add - nd.d_unique_sub;
        stats.block_stats.computed_lanes += nd.d_computed;
        stats.block_stats.num_incremental_blocks += nd.d_incremental;
        PassInit init;
        init.js1 = nd.jr.js;
        init.je1 = nd.jr.je;
        init.ojs1 = nd.jr.ojs;
        init.oje1 = nd.jr.oje;
        init.flags1 = nd.flags;
        init.top1 = nd.jr.js + (n < kBlockW ? n : kBlockW);
        init.fs0 = r0.fs;
        const int32_t mrows = ((m + 63) / 64) * 64;
        init.last_strip = ((nd.jr.je < mrows ? nd.jr.je : mrows) - 1) / kStripRows;
        if (init.last_strip < 0) init.last_strip = 0;
        // ---- the pass ----
        const double t0 = engine::now_s();
        const Status st = dev.run_pass(f_max, params.sparse_h ? 1 : 0, init);
        stats.block_stats.t_compute += engine::now_s() - t0;
        if (st.state == kStAbort || st.state == kStTimeout || st.state == kStRunning)
            throw SweepFallback(st.state == kStAbort ? "sweep pass aborted" : "sweep pass timed out", st.state == kStAbort ? st.value : -1);
        stats.block_stats.num_blocks += st.stats.num_blocks;
        stats.block_stats.unique_lanes += st.stats.unique_lanes;
        stats.block_stats.computed_lanes += st.stats.computed_lanes;
        stats.block_stats.num_incremental_blocks += st.stats.num_incremental_blocks;
        dev.commit(st.k_end, st.k_fixed);
        last_block_idx = st.k_end;
        if ((int64_t)st.k_end + 1 > blocks_len) blocks_len = (int64_t)st.k_end + 1;
        if (st.state == kStNoPath) return std::nullopt;
        const Cost dist = st.value;
        if (trace && dist <= f_max) {
            cigar = do_trace(r0);
            return std::make_pair(dist, std::optional<engine::Cigar>(*cigar));
        }
        return std::make_pair(dist, std::optional<engine::Cigar>());
    }

    // Blocks::trace (blocks/trace.rs:21-135) over the blocks of the pass that just succeeded.
    engine::Cigar do_trace(const BlockRec& r0) {
        using namespace engine;
        Blocks<Backend> blocks(params.front, true, be);
        blocks.blocks.resize((size_t)nblk + 1);
        blocks.blocks[0] = Block::first_col(JRange{r0.ojs, r0.oje}, JRange{r0.js, r0.je});
        dev.read_blocks(blocks.blocks);
        blocks.last_block_idx = (size_t)nblk;
        blocks.i_range = IRange{-1, n};
        auto [cg, ts] = blocks.trace(0, 0, n, m);
        stats.trace_stats = ts;
        return cg;
    }

    // lib.rs:122-175
    engine::AlignResult align() {
        using namespace engine;
        AlignResult out;
        const Cost h_0 = h0();
        Cost start_f = 0, start_inc = 1;  // band.rs:13-23
        if (params.start == DoublingStart::Gap) {
            start_f = start_inc = unit_gap_cost(0, 0, n, m);
        } else if (params.start == DoublingStart::H0) {
            start_f = h_0;
            start_inc = 1;
        }
        auto f = [&](Cost s) { return pass(s); };
        std::pair<Cost, std::optional<Cigar>> r;
        if (params.doubling == DoublingKind::LinearSearch) {
            const Cost delta = (Cost)params.delta;
            r = band_search(start_f, [delta](Cost s) { return s + delta; }, f, &stats.sanity_violations);
        } else {
            start_inc = std::max(start_inc, params.block_width);  // lib.rs:142
            const float factor = params.factor;
            const Cost offset = start_f;
            r = band_search(offset + start_inc,
                            [factor, offset](Cost s) { return std::max((Cost)std::ceil(factor * (float)(s - offset)), 1) + offset; }, f,
                            &stats.sanity_violations);
        }
        PA_ASSERT(h_0 <= r.first, "Heuristic at start > final cost");
        out.cost = r.first;
        out.has_cigar = r.second.has_value();
        if (r.second) out.cigar = std::move(*r.second);
        out.stats = stats;
        return out;
    }
};

}  // namespace sweep
}  // namespace pa
