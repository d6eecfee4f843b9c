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

// A Launcher provides device memory and the passes of ONE pair, numbered seq = 1, 2, ...:
//   begin_pair(n, m, nblk, sh table or nullptr, trace)
//   launch_pass(seq, prev_seq, f_max, sparse_h, init)   start pass seq (asynchronous); it reads pass prev_seq's records while that
//                                             runs, its merged records afterwards (prev_seq 0: no earlier pass)
//   wait_pass(seq) -> Status                  block until pass seq is over and merged (passes are waited for in order)
//   cancel_after(seq, wait = true)            give up every launched pass > seq and (wait) block until they are gone
//   cancel_without_waiting()                  policy: after the successful pass, start the traceback before the passes behind it are gone
//   read_merged(seq, k) -> BlockRec           block k's record after pass seq was merged (waited)
//   read_blocks(seq, blocks)                  the blocks of pass seq for Blocks::trace
//   pass_waves(f_max), wave_budget()          wavefronts a pass occupies / may be in flight together
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
    std::optional<engine::Cigar> cigar;

    // What the reference keeps between align_for_bounded_dist calls, as far as the first column and block 1 look at it.
    struct Carry {
        int32_t last_block_idx = 0;  // Blocks::last_block_idx as the previous pass left it (domain.rs:395-404 reads through it)
        int64_t blocks_len = 0;      // Blocks::blocks.len()
        BlockRec rec0;               // the first column as the previous pass left it
        bool have_rec0 = false;
        BlockRec old1;               // block 1 as the previous passes left it (js == kNone: none)
        int seq = 0;                 // the last device pass of the chain (0: none)
        Carry() { old1.js = kNone; }
    };
    Carry carry;
    int seq_counter = 0;  // device passes launched so far (given up or not)

    // A pass up to its launch: the first column and block 1 need no DP (domain.rs:395-413, blocks.rs:146-179).
    struct Prepared {
        Cost f_max = 0;
        bool early_none = false;  // no path for this bound before any block is computed
        bool rec0_set = false;    // the first column was (re)initialised before the pass gave up
        BlockRec r0;
        PassInit init;
        NextDecision nd;
        int seq = 0;
        double t0 = 0;
        Carry after_launch;       // the carry the NEXT pass is prepared from while this one still runs (see speculate())
    };

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
            engine::SeedHeuristicH sh(be.a(), n, be.b(), m, p.heuristic_k, (int)p.heuristic_p);
            sh_h.assign(sh.h_by_i.begin(), sh.h_by_i.end());
            hp.sh_h = sh_h.data();
        }
        stats.t_precomp = engine::now_s() - t0;
        dev.begin_pair(n, m, nblk, hp.kind == kHeurSH ? sh_h.data() : nullptr, trace);
    }

    Cost h0() const { return heur_h(hp, 0, 0); }

    // The host part of a pass before the launch, from the carry `c` (the real one, or the one assumed while the previous pass
    // still runs).  `stale` = the block after the previous pass's last one, as `blocks.next_block_j_range()` sees it before init().
    Prepared prepare(Cost f_max, const Carry& c, const BlockRec& stale) const {
        Prepared p;
        p.f_max = f_max;
        if (f_max < 0) engine::engine_panic("f_max >= 0");
        JRangeOut jr0;
        if (!next_j_range(hp, -1, 0, -1, -1, 0, f_max, params.sparse_h ? 1 : 0, stale.js, stale.je, &jr0) || jr0.ojs > 0) {
            p.early_none = true;
            return p;
        }
        BlockRec r0;
        r0.ojs = jr0.ojs;
        r0.oje = jr0.oje;
        r0.js = 0;
        r0.je = jr0.je;
        if (c.have_rec0 && c.rec0.je > r0.je) r0.je = c.rec0.je;  // initial_j_range.union(blocks[0].j_range), rounded
        r0.fs = jr0.ojs;
        r0.fe = jr0.oje;
        r0.top_val = 0;
        r0.bot_val = r0.je;
        p.r0 = r0;
        p.rec0_set = true;
        // ---- block 1 from the first column: index_0(j) = j ----
        BlockRec old1;
        old1.js = kNone;
        if (c.blocks_len > 1) old1 = c.old1;
        p.nd = decide_next(hp, f_max, params.sparse_h ? 1 : 0, 0, n < kBlockW ? n : kBlockW, r0.fs, r0.fe, r0.fe, old1, true);
        if (!p.nd.ok) {
            if (old1.js != kNone) engine::engine_panic("empty j_range with existing next block");
            p.early_none = true;
            return p;
        }
        p.init.js1 = p.nd.jr.js;
        p.init.je1 = p.nd.jr.je;
        p.init.ojs1 = p.nd.jr.ojs;
        p.init.oje1 = p.nd.jr.oje;
        p.init.flags1 = p.nd.flags;
        p.init.top1 = p.nd.jr.js + (n < kBlockW ? n : kBlockW);
        p.init.fs0 = r0.fs;
        const int32_t mrows = ((m + 63) / 64) * 64;
        p.init.last_strip = ((p.nd.jr.je < mrows ? p.nd.jr.je : mrows) - 1) / kStripRows;
        if (p.init.last_strip < 0) p.init.last_strip = 0;
        return p;
    }
    BlockRec stale_of(const Carry& c) {  // needs the previous pass merged
        BlockRec stale;
        stale.js = kNone;
        if ((int64_t)c.last_block_idx + 1 < c.blocks_len) stale = c.last_block_idx + 1 == 1 ? c.old1 : dev.read_merged(c.seq, c.last_block_idx + 1);
        return stale;
    }
    // The carry after `p` as far as it is known at launch time.
    void apply_prepared(Carry& c, const Prepared& p) const {
        if (p.rec0_set) {
            c.rec0 = p.r0;
            c.have_rec0 = true;
            if (c.blocks_len < 1) c.blocks_len = 1;
            c.last_block_idx = 0;
        }
    }
    void launch(Prepared& p, int prev_seq) {
        seq_counter += 1;
        p.seq = seq_counter;
        p.t0 = engine::now_s();
        dev.launch_pass(p.seq, prev_seq, p.f_max, params.sparse_h ? 1 : 0, p.init);
    }
    // The rest of the pass once its status is known (in pass order).  nullopt = no path for this bound (domain.rs returns None).
    std::optional<std::pair<Cost, std::optional<engine::Cigar>>> complete(const Prepared& p, const Status& st) {
        stats.block_stats.t_compute += engine::now_s() - p.t0;
        if (st.state == kStAbort || st.state == kStTimeout || st.state == kStRunning)
            throw SweepFallback(st.state == kStAbort ? "sweep pass aborted" : "sweep pass timed out", st.state == kStAbort ? st.value : -1);
        stats.block_stats.num_blocks += p.nd.d_num_blocks;
        stats.block_stats.unique_lanes += p.nd.d_unique_add - p.nd.d_unique_sub;
        stats.block_stats.computed_lanes += p.nd.d_computed;
        stats.block_stats.num_incremental_blocks += p.nd.d_incremental;
        stats.block_stats.num_blocks += st.stats.num_blocks;
        stats.block_stats.unique_lanes += st.stats.unique_lanes;
        stats.block_stats.computed_lanes += st.stats.computed_lanes;
        stats.block_stats.num_incremental_blocks += st.stats.num_incremental_blocks;
        carry.seq = p.seq;
        carry.last_block_idx = st.k_end;
        if ((int64_t)st.k_end + 1 > carry.blocks_len) carry.blocks_len = (int64_t)st.k_end + 1;
        if (st.k_end >= 1) {  // block 1's merged record, as the host needs it next pass
            carry.old1.js = p.init.js1;
            carry.old1.je = p.init.je1;
            carry.old1.ojs = p.init.ojs1;
            carry.old1.oje = p.init.oje1;
        }
        if (st.state == kStNoPath) return std::nullopt;
        const Cost dist = st.value;
        if (trace && dist <= p.f_max) {
            cigar = do_trace(p);
            return std::make_pair(dist, std::optional<engine::Cigar>(*cigar));
        }
        return std::make_pair(dist, std::optional<engine::Cigar>());
    }

    // Blocks::trace (blocks/trace.rs:21-135) over the blocks of the pass that just succeeded.
    engine::Cigar do_trace(const Prepared& p) {
        using namespace engine;
        Blocks<Backend> blocks(params.front, true, be);
        blocks.blocks.resize((size_t)nblk + 1);
        blocks.blocks[0] = Block::first_col(JRange{p.r0.ojs, p.r0.oje}, JRange{p.r0.js, p.r0.je});
        dev.read_blocks(p.seq, blocks.blocks);
        blocks.last_block_idx = (size_t)nblk;
        blocks.i_range = IRange{-1, n};
        auto [cg, ts] = blocks.trace(0, 0, n, m);
        stats.trace_stats = ts;
        return cg;
    }

    // The band search (band.rs:100-182, engine.hpp band_search) with the passes PIPELINED: while the pass for bound s runs, the
    // passes for next_s(s), next_s(next_s(s)), ... are prepared and launched under the assumptions that (a) the pass before
    // them finds no path or a cost above its bound that does not cap the next bound, and (b) it gets at least as far as every
    // pass before it, so that no stale block record reaches the first column (the quirk of domain.rs:395-404).  Both hold for
    // a band that grows with the bound; when the finished pass says otherwise the passes launched after it are given up and
    // the search goes on from the real state.  A pass that succeeds likewise gives up the ones behind it.
    struct InFlight {
        Prepared p;
        int waves;
    };
    std::pair<Cost, std::optional<engine::Cigar>> search(Cost first_s, const std::function<Cost(Cost)>& next_s) {
        static const bool no_pipe = std::getenv("PA_SWEEP_NO_PIPELINE") != nullptr;
        std::vector<InFlight> fl;  // launched, oldest first
        size_t head = 0;
        int waves_in_flight = 0;
        Carry spec;                // the carry after the newest launched pass, as assumed
        Cost last_s = -1, s = first_s, maxs = engine::COST_MAX;
        auto give_up = [&]() {  // every launched pass behind the head's predecessor
            if (head < fl.size()) dev.cancel_after(fl[head].p.seq - 1);
            fl.resize(head);
            waves_in_flight = 0;
        };
        for (;;) {
            std::optional<std::pair<Cost, std::optional<engine::Cigar>>> r;
            stats.f_max_tries += 1;
            if (head == fl.size()) {  // nothing launched for this bound: prepare it from the real state
                Prepared p = prepare(s, carry, stale_of(carry));
                apply_prepared(carry, p);
                if (!p.early_none) {
                    launch(p, carry.seq);
                    fl.push_back(InFlight{p, dev.pass_waves(s)});
                    waves_in_flight = fl.back().waves;
                    spec = carry;
                    spec.blocks_len = spec.blocks_len < 2 ? 2 : spec.blocks_len;  // the pass reaches block 1 at least
                    spec.old1.js = p.init.js1;
                    spec.old1.je = p.init.je1;
                    spec.old1.ojs = p.init.ojs1;
                    spec.old1.oje = p.init.oje1;
                    spec.seq = p.seq;
                }
            }
            if (head < fl.size()) {
                // look ahead while there is room on the chip
                if (!no_pipe) {
                    Cost sa = fl.back().p.f_max;
                    while ((int)(fl.size() - head) < dev.max_in_flight()) {
                        const Cost sn = next_s(sa);
                        if (sn <= sa) break;
                        const int w = dev.pass_waves(sn);
                        if (waves_in_flight + w > dev.wave_budget()) break;
                        BlockRec none;
                        none.js = kNone;
                        Prepared q = prepare(sn, spec, none);  // assumption (b): nothing stale
                        if (q.early_none) break;               // (decided again from the real state when its turn comes)
                        apply_prepared(spec, q);
                        launch(q, spec.seq);
                        fl.push_back(InFlight{q, w});
                        waves_in_flight += w;
                        spec.old1.js = q.init.js1;
                        spec.old1.je = q.init.je1;
                        spec.old1.ojs = q.init.ojs1;
                        spec.old1.oje = q.init.oje1;
                        spec.seq = q.seq;
                        sa = sn;
                    }
                }
                const InFlight cur = fl[head];
                head += 1;
                const int64_t len_before = carry.blocks_len;
                apply_prepared(carry, cur.p);
                Status st;
                try {
                    st = dev.wait_pass(cur.p.seq);
                    // found: the passes behind it are not needed.  (Waiting for them here costs less than letting them run into the
                    // traceback's kernels: 14.4 against 14.9 ms on C3.)
                    // Short pairs (round 6): the cancel words go out, the traceback does not wait for the passes to see them -- three
                    // speculative passes of a 10 kbp pair took 30-90 us to wind down plus their merges, a tenth of the call, and their
                    // handful of wavefronts does not hurt the traceback's two small kernels; give_up() below waits when they are long gone.
                    if (st.state == kStDone && st.value <= cur.p.f_max) {
                        if (dev.cancel_without_waiting() && head < fl.size()) dev.cancel_after(fl[head].p.seq - 1, false);
                        else give_up();
                    }
                    r = complete(cur.p, st);
                } catch (...) {
                    give_up();
                    throw;
                }
                waves_in_flight -= cur.waves;
                // assumption (b) for the passes launched behind this one
                if (head < fl.size() && (int64_t)st.k_end + 1 < (len_before > 2 ? len_before : 2)) give_up();
                // (tests: PA_SWEEP_TEST_GIVE_UP=k pretends that an assumption failed after every k-th pass, so that giving up,
                //  cancelling and launching again from the real state run all the time; read per pair)
                if (const char* e = std::getenv("PA_SWEEP_TEST_GIVE_UP")) {
                    const int k = std::atoi(e);
                    if (k > 0 && head < fl.size() && cur.p.seq % k == 0) give_up();
                }
            }
            // ---- band.rs:100-182 ----
            if (r) {
                const Cost cost = r->first;
                if (cost > maxs) stats.sanity_violations += 1;  // band.rs:118-121
                if (cost <= s) {
                    if (cost <= last_s) stats.sanity_violations += 1;  // band.rs:123-126
                    give_up();
                    return *r;
                }
                maxs = std::min(maxs, cost);
            } else if (maxs != engine::COST_MAX) {
                stats.sanity_violations += 1;  // band.rs:132-135
            }
            const Cost prev = s;
            last_s = s;
            s = std::min(next_s(s), maxs);
            if (s <= prev) s = next_s(prev);  // never stall (the reference's "potential infinite loop" TODO, band.rs:110)
            if (head < fl.size() && fl[head].p.f_max != s) give_up();  // assumption (a)
        }
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
        std::pair<Cost, std::optional<Cigar>> r;
        if (params.doubling == DoublingKind::LinearSearch) {
            const Cost delta = (Cost)params.delta;
            r = search(start_f, [delta](Cost s) { return s + delta; });
        } else {
            start_inc = std::max(start_inc, params.block_width);  // lib.rs:142
            const float factor = params.factor;
            const Cost offset = start_f;
            r = search(offset + start_inc, [factor, offset](Cost s) { return std::max((Cost)std::ceil(factor * (float)(s - offset)), 1) + offset; });
        }
        PA_ASSERT(h_0 <= r.first, "Heuristic at start > final cost");
        out.cost = r.first;
        out.has_cigar = r.second.has_value();
        if (r.second) out.cigar = std::move(*r.second);
        out.stats = stats;
        if (params.doubling != DoublingKind::BandDoubling) out.stats.block_stats = BlockStats{};  // lib.rs:132-140 against 158
        return out;
    }
};

}  // namespace sweep
}  // namespace pa
