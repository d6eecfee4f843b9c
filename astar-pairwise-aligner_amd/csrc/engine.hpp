// engine.hpp -- the A*PA2 block-DP engine (host orchestration), templated over a kernel backend.
//
// Restates, from reading the reference (paths relative to /root/reference/astarpa2/src/):
//   ranges.rs:10-124      IRange / JRange / rounding
//   block.rs:8-160        Block (right-edge column of vertical deltas, index/get/get_diff)
//   blocks.rs:31-831      BlockParams, Blocks::{init, compute_next_block (incl. incremental doubling),
//                         fill_with_blocks}, HMode, init_v_with_overlap[_preserve_fixed]
//   blocks/trace.rs:21-500 trace, parent, dt_trace_block, extend_left
//   domain.rs:77-541      j_range, fixed_j_range, align_for_bounded_dist
//   band.rs:13-182        DoublingStart, exponential_search, linear_search
//   lib.rs:122-175        cost_or_align;   params.rs:46-128 presets nw / simple / full
//
// The DP rectangles themselves are NOT computed here: every `compute` / `fill` goes to the Backend
// (the HIP strip kernels in the shipped library; see engine_hip.hip).  The backend owns the sequence
// profiles and the persistent horizontal-delta row `h` used by incremental doubling (blocks.rs:103-105).
#pragma once
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace pa {
namespace engine {

using I = int32_t;
using Cost = int32_t;
constexpr I WI = 64;  // lib.rs:35
constexpr Cost COST_MAX = INT32_MAX;

struct EnginePanic : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] inline void engine_panic(const char* msg) {
    // The reference panics (assert!/panic!) on internal inconsistencies.  We throw; the C boundary turns it
    // into an error code (pa_align) or, like a Rust panic crossing FFI, into abort() (astarpa2_* symbols).
    throw EnginePanic(msg);
}
#define PA_ASSERT(cond, msg) \
    do {                     \
        if (!(cond)) ::pa::engine::engine_panic(msg); \
    } while (0)

inline I next_multiple_of(I x, I r) {  // i32::next_multiple_of (rounds towards +inf)
    I m = x % r;
    if (m < 0) m += r;
    return m == 0 ? x : x + (r - m);
}
inline I div_ceil(I a, I b) { return (a + b - 1) / b; }  // only used with positive operands

// ---- encoding.rs:5-74 ---------------------------------------------------------------------------
struct V {
    uint64_t p = 0, m = 0;
    static V one() { return V{~0ull, 0}; }
    Cost value() const { return (Cost)__builtin_popcountll(p) - (Cost)__builtin_popcountll(m); }
    Cost value_of_prefix(I j) const {  // 0 <= j < 64
        const uint64_t mask = (1ull << j) - 1;
        return (Cost)__builtin_popcountll(p & mask) - (Cost)__builtin_popcountll(m & mask);
    }
    Cost value_of_suffix(I j) const {  // 0 < j <= 64
        const uint64_t mask = ~((1ull << (64 - j)) - 1);
        return (Cost)__builtin_popcountll(p & mask) - (Cost)__builtin_popcountll(m & mask);
    }
    bool operator==(const V& o) const { return p == o.p && m == o.m; }
};

// ---- ranges.rs ----------------------------------------------------------------------------------
struct IRange {  // left-exclusive (i0, i1]: characters a[i0..i1)
    I s = -1, e = 0;
    I len() const { return e - s; }
    bool operator==(const IRange& o) const { return s == o.s && e == o.e; }
};
struct JRange {  // inclusive rows
    I s = -WI, e = -WI;
    bool is_empty() const { return s > e; }
    I len() const { return e - s + 1; }
    I exclusive_len() const { return e - s; }
    bool contains(I j) const { return s <= j && j <= e; }
    bool contains_range(JRange o) const { return s <= o.s && o.e <= e; }
    JRange union_(JRange o) const { return JRange{std::min(s, o.s), std::max(e, o.e)}; }
    JRange intersection(JRange o) const { return JRange{std::max(s, o.s), std::min(e, o.e)}; }
    JRange round_out() const { return JRange{s / WI * WI, next_multiple_of(e, WI)}; }  // ranges.rs:71-73
    JRange round_in() const { return JRange{next_multiple_of(s, WI), e / WI * WI}; }   // ranges.rs:74-76
    bool operator==(const JRange& o) const { return s == o.s && e == o.e; }
    // v_range of an (already rounded) range: word indices [s/64, e/64)   ranges.rs:93-95
    size_t v_start() const { return (size_t)(s / WI); }
    size_t v_end() const { return (size_t)(e / WI); }
    size_t v_len() const { return v_end() - v_start(); }
};
struct VRange {
    size_t s = 0, e = 0;
    size_t len() const { return e - s; }
    bool empty() const { return s >= e; }
};
inline VRange v_range_of(JRange rounded) {
    PA_ASSERT(rounded.s % WI == 0 && rounded.e % WI == 0, "assert_rounded");
    return VRange{rounded.v_start(), rounded.v_end()};
}

// ---- CIGAR (pa-types, external: semantics inferred from call sites) -------------
enum class CigarOp : uint8_t { Match, Sub, Del, Ins };
struct CigarElem {
    CigarOp op;
    I cnt;
};
struct Cigar {
    std::vector<CigarElem> ops;
    void push_elem(CigarElem e) {  // merges with an equal trailing op (pa-affine-types/src/cigar.rs:137-146)
        if (!ops.empty() && ops.back().op == e.op) {
            ops.back().cnt += e.cnt;
            return;
        }
        ops.push_back(e);
    }
    void reverse() { std::reverse(ops.begin(), ops.end()); }
    // Format pinned by astarpa-c/example.cpp:16 ("=I4=X="): count omitted when 1; = X I D.
    std::string to_string() const {
        std::string s;
        for (const auto& e : ops) {
            if (e.cnt != 1) s += std::to_string(e.cnt);
            s += e.op == CigarOp::Match ? '=' : e.op == CigarOp::Sub ? 'X' : e.op == CigarOp::Ins ? 'I' : 'D';
        }
        return s;
    }
};

// ---- parameters (params.rs:8-42, blocks.rs:31-74, band.rs:5-63) -----------------------------------
enum class DomainKind : int32_t { Full = 0, GapStart = 1, GapGap = 2, Astar = 3 };
enum class HeuristicKind : int32_t { None = 0, Gap = 1, SH = 2, GCSH = 3 };  // NoCost (Dijkstra) / GapCost / SH / GCSH (exact matches)
enum class DoublingKind : int32_t { None = 0, BandDoubling = 1, LinearSearch = 2 };
enum class DoublingStart : int32_t { Zero = 0, Gap = 1, H0 = 2 };

struct BlockParams {
    bool sparse = true;
    bool simd = true;
    bool no_ilp = false;
    bool incremental_doubling = true;
    bool dt_trace = false;
    Cost max_g = 40;
    I fr_drop = 20;
};

struct AstarPa2Params {
    DomainKind domain = DomainKind::Astar;
    HeuristicKind heuristic = HeuristicKind::Gap;
    I heuristic_k = 15;  // HeuristicParams.k (pa-heuristic/src/cli.rs:47-114): seed length for SH / GCSH
    I heuristic_p = 0;   // HeuristicParams.p: local-pruning look-ahead of GCSH (0 = off)
    DoublingKind doubling = DoublingKind::BandDoubling;
    DoublingStart start = DoublingStart::H0;
    float factor = 2.0f;
    float delta = 1.0f;
    I block_width = 256;
    BlockParams front;
    bool sparse_h = false;
    bool prune = false;

    static AstarPa2Params nw() {  // params.rs:46-68
        AstarPa2Params p;
        p.domain = DomainKind::Full;
        p.heuristic = HeuristicKind::None;
        p.doubling = DoublingKind::None;
        p.block_width = 256;
        p.front = BlockParams{false, true, false, false, false, 40, 20};
        p.sparse_h = false;
        p.prune = false;
        return p;
    }
    static AstarPa2Params full() {  // params.rs:98-128: GCSH(k=12, r=1, p=14, Prune::Start) + incremental doubling + pruning
        AstarPa2Params p;
        p.domain = DomainKind::Astar;
        p.heuristic = HeuristicKind::GCSH;
        p.heuristic_k = 12;
        p.heuristic_p = 14;
        p.doubling = DoublingKind::BandDoubling;
        p.start = DoublingStart::H0;
        p.factor = 2.0f;
        p.block_width = 256;
        p.front = BlockParams{true, true, false, true, true, 40, 10};
        p.sparse_h = true;
        p.prune = true;
        return p;
    }
    static AstarPa2Params simple() {  // params.rs:70-96
        AstarPa2Params p;
        p.domain = DomainKind::Astar;
        p.heuristic = HeuristicKind::Gap;
        p.doubling = DoublingKind::BandDoubling;
        p.start = DoublingStart::H0;
        p.factor = 2.0f;
        p.block_width = 256;
        p.front = BlockParams{true, true, false, false, true, 40, 10};
        p.sparse_h = true;
        p.prune = false;
        return p;
    }
};

struct BlockStats {  // blocks.rs:76-84
    size_t num_blocks = 0, num_incremental_blocks = 0, computed_lanes = 0, unique_lanes = 0;
    double t_compute = 0;
};
struct TraceStats {  // blocks/trace.rs:3-14
    size_t dt_trace_tries = 0, dt_trace_success = 0, dt_trace_fallback = 0;
    size_t fill_tries = 0, fill_success = 0, fill_fallback = 0;
    double t_dt = 0, t_fill = 0;
};
struct AstarPa2Stats {  // domain.rs:31-43
    BlockStats block_stats;
    TraceStats trace_stats;
    size_t f_max_tries = 0;
    size_t sanity_violations = 0;  // see band_search
    double t_precomp = 0, t_j_range = 0, t_fixed_j_range = 0, t_pruning = 0, t_contours_update = 0;
};

inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- heuristics (pa-heuristic, host side) --------------------------------------------------------
struct Heuristic {
    virtual ~Heuristic() = default;
    virtual Cost h(I i, I j) const = 0;
    virtual void prune_block(I, I, I, I) {}   // heuristic.rs:150-157: default no-op
    virtual void update_contours() {}
};
struct NoCostH : Heuristic {  // distances.rs NoCost (Dijkstra domain)
    Cost h(I, I) const override { return 0; }
};
struct GapCostH : Heuristic {  // distances.rs:131-168
    I tn, tm;
    GapCostH(I n, I m) : tn(n), tm(m) {}
    Cost h(I i, I j) const override {
        const int64_t d = (int64_t)(tn - i) - (int64_t)(tm - j);
        return (Cost)(d < 0 ? -d : d);
    }
};

// SH (seed heuristic), pa-heuristic/src/heuristic/sh.rs:47-106 + contour/sh_contours.rs:16-75, for exact matches
// (r = 1, no local pruning, no transform filter -- the configuration astarpa2/src/tests.rs:68-79 runs).
// `a` is cut into disjoint k-mers at i = 0,k,2k,.. (matches/qgrams.rs:99-109); a seed has score 1 iff its k-mer
// occurs anywhere in b (matches/exact.rs:15-69; keys are 2-bit packed and truncated to u32 as there).
// h(i,j) = potential(i) - score(i) = number of seeds starting at >= i without a match.  SHI does not override
// prune_block / update_contours (heuristic.rs:150-157), so under A*PA2 it is static and column-only.
// HeuristicParams.p != 0 reaches SH as well (cli.rs:168-180: MatchConfig.local_pruning = p for every heuristic; sh.rs:48 passes it to
// find_matches with transform_filter = false): a seed then counts as matched only if one of its matches SURVIVES local pruning
// (MatchBuilder::push, matches.rs:205-247) -- sh_matched_with_local_pruning below, after gcsh.hpp.
struct SeedHeuristicH : Heuristic {
    std::vector<Cost> h_by_i;  // size n+1
    static std::vector<uint8_t> sh_matched_with_local_pruning(const uint8_t* a, I n, const uint8_t* b, I m, I k, int p);
    SeedHeuristicH(const uint8_t* a, I n, const uint8_t* b, I m, I k, int p = 0) {
        h_by_i.assign((size_t)n + 1, 0);
        if (k <= 0) k = 1;
        const I nseeds = n >= k ? (n - k) / k + 1 : 0;
        if (p != 0) {
            const std::vector<uint8_t> matched = sh_matched_with_local_pruning(a, n, b, m, k, p);
            fill_from(matched, n, k, nseeds);
            return;
        }
        auto bits = [](uint8_t c) -> uint64_t { return (uint64_t)((c >> 1) & 3); };  // qgrams.rs:30-33
        std::vector<std::pair<uint32_t, I>> keys;  // (key, seed index), sorted => multimap
        keys.reserve((size_t)nseeds);
        for (I sidx = 0; sidx < nseeds; ++sidx) {
            uint64_t q = 0;
            for (I t = 0; t < k; ++t) q = (q << 2) | bits(a[sidx * k + t]);
            keys.emplace_back((uint32_t)q, sidx);
        }
        std::sort(keys.begin(), keys.end());
        std::vector<uint8_t> matched((size_t)nseeds, 0);
        if (m >= k && nseeds > 0) {
            const uint64_t mask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1);
            uint64_t q = 0;
            for (I j = 0; j < m; ++j) {
                q = ((q << 2) | bits(b[j])) & mask;
                if (j + 1 < k) continue;
                const uint32_t key = (uint32_t)q;
                auto it = std::lower_bound(keys.begin(), keys.end(), std::make_pair(key, (I)0));
                for (; it != keys.end() && it->first == key; ++it) matched[(size_t)it->second] = 1;
            }
        }
        fill_from(matched, n, k, nseeds);
    }
    // potential[i] - score(i): walk seeds from the right (seeds.rs:47-66, sh_contours.rs:40-47,63-75)
    void fill_from(const std::vector<uint8_t>& matched, I n, I k, I nseeds) {
        Cost unmatched = 0;
        I next_seed = nseeds - 1;
        for (I i = n; i >= 0; --i) {
            if (next_seed >= 0 && i == next_seed * k) {
                if (!matched[(size_t)next_seed]) unmatched += 1;
                next_seed -= 1;
            }
            h_by_i[(size_t)i] = unmatched;
        }
    }
    Cost h(I i, I) const override { return h_by_i[(size_t)i]; }
};

}  // namespace engine
}  // namespace pa
#include "gcsh.hpp"  // GcshHeuristic (needs Heuristic / I / Cost from above)
namespace pa {
namespace engine {

inline std::vector<uint8_t> SeedHeuristicH::sh_matched_with_local_pruning(const uint8_t* a, I n, const uint8_t* b, I m, I k, int p) {
    // the matches MatchBuilder keeps for find_matches(a, b, {k, r = 1, local_pruning = p}, transform_filter = false): gcsh.hpp's push loop
    const GcshHeuristic g(a, n, b, m, k, p, /*prune=*/false, /*build_layers=*/false, /*transform_filter=*/false);
    std::vector<uint8_t> matched((size_t)g.nseeds, 0);
    for (const GcshHeuristic::Match& mt : g.by_start) matched[(size_t)(mt.i / g.k)] = 1;
    return matched;
}


// unit-cost AffineCost formulas (pa-affine-types/src/cost_model.rs:387-401,453-525 with sub=ins=del=1)
inline Cost unit_gap_cost(I si, I sj, I ti, I tj) {
    const int64_t d = (int64_t)(ti - si) - (int64_t)(tj - sj);
    return (Cost)(d < 0 ? -d : d);
}
inline Cost unit_extend_cost(I si, I sj, I ti, I tj) { return unit_gap_cost(si, sj, ti, tj); }

// ---- block.rs ---------------------------------------------------------------------------------------
struct Block {
    std::vector<V> v;
    IRange i_range{-1, 0};
    JRange original_j_range{-WI, -WI};
    JRange j_range{-WI, -WI};  // rounded out
    std::optional<JRange> fixed_j_range;
    I offset = 0;
    Cost top_val = COST_MAX;
    Cost bot_val = COST_MAX;
    std::optional<I> j_h;

    static Block first_col(JRange original, JRange rounded) {  // block.rs:51-66
        PA_ASSERT(rounded.s == 0, "first_col: j_range.0 == 0");
        Block b;
        b.v.assign((size_t)(rounded.exclusive_len() / WI), V::one());
        b.i_range = IRange{-1, 0};
        b.original_j_range = original;
        b.j_range = rounded;
        b.fixed_j_range = original;
        b.offset = 0;
        b.top_val = 0;
        b.bot_val = rounded.exclusive_len();
        b.j_h.reset();
        return b;
    }

    Cost index(I j) const {  // block.rs:69-122
        PA_ASSERT(j_range.s <= j, "Cannot index block below its range");
        PA_ASSERT(j_range.s - offset >= 0, "Offset too large");
        PA_ASSERT(j_range.e - offset <= (I)v.size() * WI, "v not long enough");
        if (j > j_range.e) return bot_val + (Cost)(j - j_range.e);
        if (j - j_range.s < j_range.e - j) {
            Cost val = top_val;
            I j0 = j_range.s;
            while (j0 + WI <= j) {
                val += v[(size_t)(j0 - offset) / 64].value();
                j0 += WI;
            }
            return val + v[(size_t)(j0 - offset) / 64].value_of_prefix(j - j0);
        } else {
            Cost val = bot_val;
            I j1 = j_range.e;
            while (j1 - WI > j) {
                val -= v[(size_t)(j1 - WI - offset) / 64].value();
                j1 -= WI;
            }
            if (j1 > j) val -= v[(size_t)(j1 - WI - offset) / 64].value_of_suffix(j1 - j);
            return val;
        }
    }
    std::optional<Cost> get(I j) const {  // block.rs:126-131
        if (j < j_range.s || j > j_range.e) return std::nullopt;
        return index(j);
    }
    std::optional<Cost> get_diff(I j) const {  // block.rs:134-145
        if (j < offset) return std::nullopt;
        const size_t idx = (size_t)(j - offset) / 64;
        if (idx >= v.size()) return std::nullopt;
        const unsigned bit = (unsigned)(j - offset) % 64;
        return (Cost)((v[idx].p >> bit) & 1) - (Cost)((v[idx].m >> bit) & 1);
    }
};

enum class HMode { None, Input, Update, Output };  // blocks.rs:665-671

// A backend MAY offer `compute_chain(i0, i1, segs, nseg, params)`: the ranges of one incrementally doubled block in one go
// (same results as calling `compute` for each in order; returns the last sum).  The CPU backend does not.
template <class B, class = void>
struct has_compute_chain : std::false_type {};
template <class B>
struct has_compute_chain<B, std::void_t<typename B::ChainSeg>> : std::true_type {};

// Backend concept (implemented by HipBackend in engine_hip.hip and by the test-only CpuBackend):
//   I n() const; I m() const;                       sequence lengths
//   const uint8_t* a() const; const uint8_t* b() const;   raw ASCII (trace uses them, trace.rs:443-500)
//   void enable_h_row();                            allocate the persistent h row (blocks.rs:119-123)
//   Cost compute(I i0, I i1, size_t w0, size_t w1, V* v, HMode mode, const BlockParams&);
//        rectangle columns [i0,i1) x words [w0,w1); v points at word w0; returns the bottom-row sum.
//        None: top = +1, bottom discarded.  Input: top = stored h (stored row unchanged).
//        Update: top = stored h, bottom stored.  Output: top = +1, bottom stored.   (blocks.rs:728-747)
//   void fill(I i0, I i1, size_t w0, size_t w1, V* v, V* values, int8_t* hbot, const BlockParams&);
//        top = +1; values[(i-i0)*(w1-w0) + (w-w0)] = V of word w after column i; hbot[i-i0] = bottom delta.

template <class Backend>
class Blocks {
   public:
    BlockParams params;
    bool trace_ = false;
    Backend& be;
    std::vector<Block> blocks;
    size_t last_block_idx = 0;
    IRange i_range{-1, 0};
    BlockStats stats;
    bool self_check = false;  // the reference's cfg!(test) recompute-and-compare (blocks.rs:471-543)

    Blocks(const BlockParams& p, bool trace, Backend& backend) : params(p), trace_(trace), be(backend) {  // blocks.rs:110-128
        if (params.incremental_doubling) be.enable_h_row();
    }

    void init(JRange initial_j_range) {  // blocks.rs:146-179
        PA_ASSERT(initial_j_range.s == 0, "init: j_range.0 == 0");
        last_block_idx = 0;
        i_range = IRange{-1, 0};
        const JRange fixed = initial_j_range;
        if (!blocks.empty()) initial_j_range = initial_j_range.union_(blocks[0].j_range);
        const JRange rounded = initial_j_range.round_out();
        Block block;
        if (trace_) {
            block = Block::first_col(fixed, rounded);
        } else {
            block.v.assign((size_t)((be.m() + 63) / 64), V::one());
            block.i_range = IRange{-1, 0};
            block.original_j_range = fixed;
            block.j_range = rounded;
            block.fixed_j_range = fixed;
            block.offset = 0;
            block.top_val = 0;
            block.bot_val = rounded.e;
            block.j_h.reset();
        }
        if (blocks.empty()) blocks.push_back(std::move(block));
        else blocks[0] = std::move(block);
    }

    void pop_last_block() {  // blocks.rs:182-185
        const IRange r = blocks[last_block_idx].i_range;
        PA_ASSERT(i_range.e == r.e, "Can not pop range");
        i_range.e = r.s;
        last_block_idx -= 1;
    }

    void reuse_next_block(IRange ir, JRange jr) {  // blocks.rs:190-197
        PA_ASSERT(i_range.e == ir.s, "IRange push");
        i_range.e = ir.e;
        last_block_idx += 1;
        PA_ASSERT(last_block_idx < blocks.size(), "reuse: block exists");
        PA_ASSERT(blocks[last_block_idx].i_range == ir, "reuse: same i_range");
        PA_ASSERT(blocks[last_block_idx].j_range == jr.round_out(), "reuse: same j_range");
    }

    const Block& last_block() const { return blocks[last_block_idx]; }
    std::optional<JRange> next_block_j_range() const {  // blocks.rs:551-553
        if (last_block_idx + 1 < blocks.size()) return blocks[last_block_idx + 1].j_range;
        return std::nullopt;
    }
    void set_last_block_fixed_j_range(std::optional<JRange> fixed) {  // blocks.rs:556-569
        auto& cur = blocks[last_block_idx].fixed_j_range;
        if (cur && fixed) cur = cur->union_(*fixed);
        else cur = fixed;
    }

    // free fn compute_block, blocks.rs:686-748
    Cost compute_block(IRange ir, VRange vr, V* v, HMode mode) {
        if (ir.len() > 1) {
            stats.computed_lanes += vr.len();
            stats.num_incremental_blocks += 1;
        }
        return be.compute(ir.s, ir.e, vr.s, vr.e, v, mode, params);
    }

    // The same accounting as one compute_block per segment (blocks.rs:704-712), then one backend call for all of them.
    template <class Seg>
    Cost compute_chain(IRange ir, const Seg* segs, int nseg) {
        for (int k = 0; k < nseg; ++k) {
            if (ir.len() > 1) {
                stats.computed_lanes += segs[k].w1 - segs[k].w0;
                stats.num_incremental_blocks += 1;
            }
        }
        return be.compute_chain(ir.s, ir.e, segs, nseg, params);
    }

    static void init_v_with_overlap(const Block& prev, Block& next) {  // blocks.rs:753-767
        PA_ASSERT(next.offset == next.j_range.s && prev.offset == prev.j_range.s, "offset == j_range.0");
        const VRange pv = v_range_of(prev.j_range), nv = v_range_of(next.j_range);
        next.v.clear();
        next.v.resize(nv.len(), V::one());
        const JRange ov_j = next.j_range.intersection(prev.j_range);
        const VRange ov = v_range_of(ov_j);
        if (ov.s < ov.e) {
            std::copy(prev.v.begin() + (ov.s - pv.s), prev.v.begin() + (ov.e - pv.s), next.v.begin() + (ov.s - nv.s));
        }
    }

    static void init_v_with_overlap_preserve_fixed(const Block& prev, const Block& old, Block& next) {  // blocks.rs:774-831
        auto& v = next.v;
        PA_ASSERT(prev.offset == prev.j_range.s && old.offset == old.j_range.s && next.offset == next.j_range.s, "offsets");
        PA_ASSERT(next.j_range.contains_range(old.j_range), "next contains old");
        const VRange pv = v_range_of(prev.j_range), ov = v_range_of(old.j_range), nv = v_range_of(next.j_range);
        PA_ASSERT(pv.s <= nv.s && nv.s <= ov.s, "range starts ordered");
        const VRange preserve = v_range_of(JRange{old.fixed_j_range->s - 1, *old.j_h}.round_in());
        PA_ASSERT(!preserve.empty(), "preserve non-empty");
        v.resize(nv.len(), V::one());
        if (nv.s != ov.s) {
            // copy_within(preserve - old_start .. , preserve.start - v_start): memmove semantics
            std::memmove(&v[preserve.s - nv.s], &v[preserve.s - ov.s], preserve.len() * sizeof(V));
        }
        std::copy(prev.v.begin() + (nv.s - pv.s), prev.v.begin() + (preserve.s - pv.s), v.begin());
        const size_t copy_end = std::min(nv.e, pv.e);
        if (copy_end > preserve.e)
            std::copy(prev.v.begin() + (preserve.e - pv.s), prev.v.begin() + (copy_end - pv.s), v.begin() + (preserve.e - nv.s));
        else
            PA_ASSERT(copy_end == preserve.e || copy_end >= preserve.e, "suffix copy range");
        for (size_t k = copy_end - nv.s; k < v.size(); ++k) v[k] = V::one();
    }

    // blocks.rs:205-545
    void compute_next_block(IRange ir, JRange j_range_in) {
        stats.num_blocks += 1;
        const double t0 = now_s();
        const JRange original_j_range = j_range_in;
        const JRange j_range = j_range_in.round_out();
        const VRange v_range = v_range_of(j_range);
        stats.unique_lanes += v_range.len();
        if (last_block_idx + 1 < blocks.size()) {
            const Block& nb = blocks[last_block_idx + 1];
            PA_ASSERT(j_range.contains_range(nb.j_range), "j_range must grow");
            stats.unique_lanes -= (size_t)(nb.j_range.exclusive_len() / WI);
        }

        if (trace_ && !params.sparse) {  // blocks.rs:232-241
            fill_with_blocks(ir, original_j_range);
            stats.t_compute += now_s() - t0;
            return;
        }

        PA_ASSERT(i_range.e == ir.s, "IRange push");
        i_range.e = ir.e;

        const Cost prev_top_val = last_block().index(j_range.s);
        const Cost prev_bot_val = last_block().index(j_range.e);

        if (!trace_ && !params.incremental_doubling) {  // blocks.rs:252-277
            Block& block = blocks[last_block_idx];
            const Cost top_val = prev_top_val + ir.len();
            const Cost bot_val = prev_bot_val + compute_block(ir, v_range, block.v.data() + v_range.s, HMode::None);
            block.i_range = ir;
            block.original_j_range = original_j_range;
            block.j_range = j_range;
            block.top_val = top_val;
            block.bot_val = bot_val;
            stats.t_compute += now_s() - t0;
            return;
        }

        PA_ASSERT(params.sparse, "sparse required");

        if (last_block_idx + 1 == blocks.size()) {
            blocks.emplace_back();
        } else {
            PA_ASSERT(blocks[last_block_idx + 1].i_range == ir, "next block i_range");
        }
        Block& prev_block = blocks[last_block_idx];
        Block& next_block = blocks[last_block_idx + 1];
        last_block_idx += 1;

        // Copy settings but not the vector (blocks.rs:297-301).
        Block old_block;
        old_block.i_range = next_block.i_range;
        old_block.original_j_range = next_block.original_j_range;
        old_block.j_range = next_block.j_range;
        old_block.fixed_j_range = next_block.fixed_j_range;
        old_block.offset = next_block.offset;
        old_block.top_val = next_block.top_val;
        old_block.bot_val = next_block.bot_val;
        old_block.j_h = next_block.j_h;

        next_block.i_range = ir;
        next_block.original_j_range = original_j_range;
        next_block.j_range = j_range;
        // fixed_j_range kept
        next_block.offset = j_range.s;
        next_block.top_val = prev_top_val + ir.len();
        next_block.bot_val = prev_bot_val;
        next_block.j_h.reset();

        if (!params.incremental_doubling || !prev_block.fixed_j_range) {  // blocks.rs:322-340
            init_v_with_overlap(prev_block, next_block);
            next_block.bot_val += compute_block(ir, v_range, next_block.v.data(), HMode::None);
            stats.t_compute += now_s() - t0;
            return;
        }

        // ---- incremental doubling, blocks.rs:342-469 ----
        const JRange prev_fixed = prev_block.fixed_j_range->round_in();
        const std::optional<JRange> old_fixed = old_block.fixed_j_range;
        next_block.j_h = prev_fixed.e;
        const I new_j_h = prev_fixed.e;
        const size_t offset = v_range.s;

        std::vector<int8_t> dbg_old_h;
        if (self_check) dbg_old_h = be.debug_read_h(ir.s, ir.e);

        bool three_range = false;
        if (old_block.j_h && old_fixed && next_multiple_of(old_fixed->s - 1, WI) < *old_block.j_h) {
            three_range = true;
            const I old_j_h = *old_block.j_h;
            init_v_with_overlap_preserve_fixed(prev_block, old_block, next_block);
            const VRange v0 = v_range_of(JRange{j_range.s, old_fixed->s - 1}.round_out());
            PA_ASSERT(v0.s <= v0.e, "v_range_0");
            PA_ASSERT(old_j_h <= new_j_h, "j_h may only increase");
            const VRange v1 = v_range_of(JRange{old_j_h, new_j_h});
            const VRange v2 = v_range_of(JRange{new_j_h, j_range.e});
            PA_ASSERT(v2.s <= v2.e, "v_range_2");
            if constexpr (has_compute_chain<Backend>::value) {
                typename Backend::ChainSeg segs[3];
                int ns = 0;
                segs[ns++] = {v0.s, v0.e, next_block.v.data() + (v0.s - offset), HMode::None};
                if (!v1.empty()) segs[ns++] = {v1.s, v1.e, next_block.v.data() + (v1.s - offset), HMode::Update};
                segs[ns++] = {v2.s, v2.e, next_block.v.data() + (v2.s - offset), HMode::Input};
                next_block.bot_val += compute_chain(ir, segs, ns);
            } else {
                compute_block(ir, v0, next_block.v.data() + (v0.s - offset), HMode::None);
                if (!v1.empty()) compute_block(ir, v1, next_block.v.data() + (v1.s - offset), HMode::Update);
                next_block.bot_val += compute_block(ir, v2, next_block.v.data() + (v2.s - offset), HMode::Input);
            }
        } else {
            init_v_with_overlap(prev_block, next_block);
            const VRange v01 = v_range_of(JRange{j_range.s, new_j_h});
            PA_ASSERT(v01.s <= v01.e, "v_range_01");
            const VRange v2 = v_range_of(JRange{new_j_h, j_range.e});
            PA_ASSERT(v2.s <= v2.e, "v_range_2");
            // NOTE: an empty output range must still run to set the stored row (blocks.rs:443-455)
            if constexpr (has_compute_chain<Backend>::value) {
                typename Backend::ChainSeg segs[2] = {{v01.s, v01.e, next_block.v.data() + (v01.s - offset), HMode::Output},
                                                      {v2.s, v2.e, next_block.v.data() + (v2.s - offset), HMode::Input}};
                next_block.bot_val += compute_chain(ir, segs, 2);
            } else {
                compute_block(ir, v01, next_block.v.data() + (v01.s - offset), HMode::Output);
                next_block.bot_val += compute_block(ir, v2, next_block.v.data() + (v2.s - offset), HMode::Input);
            }
        }

        if (self_check) {  // blocks.rs:471-543
            // The reference runs this first sub-check whenever an old j_h exists, against an `old_h` that is
            // only captured under DEBUG (blocks.rs:357-361,474-493), so it cannot pass there.  The stored row
            // is only claimed exact when the 3-range split trusted it, so that is when we check it.
            if (old_block.j_h && three_range) {
                Block nb2 = next_block;
                init_v_with_overlap(prev_block, nb2);
                const auto h2 = be.debug_read_h(ir.s, ir.e);
                const VRange vr = v_range_of(JRange{j_range.s, *old_block.j_h});
                be.compute(ir.s, ir.e, vr.s, vr.e, nb2.v.data() + (vr.s - offset), HMode::Output, params);
                PA_ASSERT(dbg_old_h == be.debug_read_h(ir.s, ir.e), "self-check: old fixed h");
                be.debug_write_h(ir.s, ir.e, h2);
            }
            {
                Block nb2 = next_block;
                init_v_with_overlap(prev_block, nb2);
                const auto h2 = be.debug_read_h(ir.s, ir.e);
                const VRange vr = v_range_of(JRange{j_range.s, new_j_h});
                be.compute(ir.s, ir.e, vr.s, vr.e, nb2.v.data() + (vr.s - offset), HMode::Output, params);
                PA_ASSERT(h2 == be.debug_read_h(ir.s, ir.e), "self-check: updated fixed h");
            }
            Block nb2 = next_block;
            init_v_with_overlap(prev_block, nb2);
            const Cost bot_diff = be.compute(ir.s, ir.e, v_range.s, v_range.e, nb2.v.data(), HMode::None, params);
            nb2.bot_val = prev_bot_val + bot_diff;
            PA_ASSERT(next_block.top_val == nb2.top_val, "self-check: top_val");
            PA_ASSERT(next_block.v == nb2.v, "self-check: v");
            PA_ASSERT(next_block.bot_val == nb2.bot_val, "self-check: bot_val");
        }
        stats.t_compute += now_s() - t0;
    }

    // blocks.rs:572-662
    void fill_with_blocks(IRange ir, JRange original_j_range) {
        const JRange j_range = original_j_range.round_out();
        PA_ASSERT(i_range.e == ir.s, "IRange push");
        i_range.e = ir.e;
        const VRange v_range = v_range_of(j_range);
        const size_t prev_idx = last_block_idx;
        PA_ASSERT(blocks[prev_idx].i_range.e == ir.s, "consecutive");

        Block next_block;
        next_block.i_range = IRange{ir.s, ir.s};
        next_block.original_j_range = original_j_range;
        next_block.j_range = j_range;
        next_block.offset = j_range.s;
        next_block.fixed_j_range.reset();
        next_block.top_val = blocks[prev_idx].index(j_range.s);
        next_block.bot_val = 0;
        next_block.j_h.reset();
        init_v_with_overlap(blocks[prev_idx], next_block);

        const size_t cols = (size_t)ir.len(), w = v_range.len();
        for (I i = ir.s; i < ir.e; ++i) {
            next_block.i_range = IRange{i, i + 1};
            next_block.top_val += 1;
            last_block_idx += 1;
            if (last_block_idx == blocks.size()) blocks.emplace_back();
            Block& dst = blocks[last_block_idx];
            // clone_from without the vector contents (they are overwritten below)
            dst.i_range = next_block.i_range;
            dst.original_j_range = next_block.original_j_range;
            dst.j_range = next_block.j_range;
            dst.fixed_j_range = next_block.fixed_j_range;
            dst.offset = next_block.offset;
            dst.top_val = next_block.top_val;
            dst.bot_val = next_block.bot_val;
            dst.j_h = next_block.j_h;
        }
        std::vector<V> values(cols * w);
        std::vector<int8_t> hbot(cols, 0);
        be.fill(ir.s, ir.e, v_range.s, v_range.e, next_block.v.data(), values.data(), hbot.data(), params);

        Cost bot_val = blocks[last_block_idx - cols].index(j_range.e);
        for (size_t c = 0; c < cols; ++c) {
            Block& blk = blocks[last_block_idx + 1 - cols + c];
            blk.v.assign(values.begin() + c * w, values.begin() + (c + 1) * w);
            bot_val += hbot[c];
            blk.bot_val = bot_val;
        }
    }

    // ---- blocks/trace.rs ---------------------------------------------------------------------------
    struct BlockElem {  // trace.rs:418-441
        I i = INT32_MAX;
        I ext = 0;
        I parent_d = 0;
    };

    static I extend_left(I& i, I i0, I& j, const uint8_t* a, const uint8_t* b) {  // trace.rs:443-500 (same result)
        I cnt = 0;
        while (i > i0 && j > 0 && a[i - 1] == b[j - 1]) {
            --i;
            --j;
            ++cnt;
        }
        return cnt;
    }

    // trace.rs:21-135
    std::pair<Cigar, TraceStats> trace(I from_i, I from_j, I to_i, I to_j) {
        PA_ASSERT(trace_, "trace requires trace mode");
        PA_ASSERT(blocks[last_block_idx].i_range.e == to_i, "last block ends at to.0");
        Cigar cigar;
        Cost g = blocks[last_block_idx].index(to_j);
        TraceStats st;
        std::vector<BlockElem> dt_cache((size_t)(params.max_g + 1) * (size_t)(params.max_g + 1));

        while (!(to_i == from_i && to_j == from_j)) {
            while (last_block_idx > 0 && blocks[last_block_idx].i_range.s >= to_i) pop_last_block();

            if (params.dt_trace && to_i > 0) {
                const Block& prev_block = blocks[last_block_idx - 1];
                if (prev_block.i_range.e < to_i - 1) {
                    st.dt_trace_tries += 1;
                    const double t0 = now_s();
                    I ni, nj;
                    const bool ok = dt_trace_block(to_i, to_j, g, prev_block, cigar, dt_cache, ni, nj);
                    st.t_dt += now_s() - t0;
                    if (ok) {
                        st.dt_trace_success += 1;
                        to_i = ni;
                        to_j = nj;
                        continue;
                    }
                    st.dt_trace_fallback += 1;
                }
            }

            if (params.sparse && to_i > 0) {
                const Block& block = blocks[last_block_idx];
                const Block& prev_block = blocks[last_block_idx - 1];
                PA_ASSERT(prev_block.i_range.e < to_i && to_i <= block.i_range.e, "trace block bracket");
                if (prev_block.i_range.e < to_i - 1 || block.i_range.e > to_i) {
                    const double t0 = now_s();
                    const JRange prev_j_range = prev_block.j_range;
                    const IRange ir{prev_block.i_range.e, to_i};
                    const JRange jr{block.j_range.s, to_j};
                    pop_last_block();
                    I height = std::min(jr.exclusive_len(), ir.len() * 5 / 4);
                    for (;;) {
                        const JRange j_range = JRange{std::max(jr.e - height, prev_j_range.s), jr.e}.round_out();
                        st.fill_tries += 1;
                        fill_with_blocks(ir, j_range);
                        if (blocks[last_block_idx].index(to_j) == g) {
                            st.fill_success += 1;
                            break;
                        }
                        st.fill_fallback += 1;
                        PA_ASSERT(j_range.s != 0, "No trace found through block");
                        for (I k = ir.s; k < ir.e; ++k) pop_last_block();
                        height *= 2;
                    }
                    st.t_fill += now_s() - t0;
                }
            }

            CigarElem elem;
            parent(to_i, to_j, g, elem);
            cigar.push_elem(elem);
        }
        PA_ASSERT(g == 0, "trace ends at distance 0");
        cigar.reverse();
        return {cigar, st};
    }

    // trace.rs:145-228.  Updates (si, sj) to the parent and g.
    void parent(I& si, I& sj, Cost& g, CigarElem& out) {
        const Block& block = blocks[last_block_idx];
        PA_ASSERT(block.i_range.e == si, "Parent of state: block.i mismatch");
        const uint8_t* a = be.a();
        const uint8_t* b = be.b();
        I cnt = 0;
        while (si > 0 && sj > 0 && a[si - 1] == b[sj - 1]) {  // BitProfile::is_match on real rows
            ++cnt;
            --si;
            --sj;
        }
        if (cnt > 0) {
            out = CigarElem{CigarOp::Match, cnt};
            return;
        }
        const auto vd = block.get_diff(sj - 1);
        if (vd && *vd == 1) {
            g -= 1;
            sj -= 1;
            out = CigarElem{CigarOp::Ins, 1};
            return;
        }
        const Block& prev_block = blocks[last_block_idx - 1];
        PA_ASSERT(prev_block.i_range.e == si - 1, "prev block is column st.0-1");
        const Cost hd = sj < prev_block.j_range.s ? 1 : g - prev_block.index(sj);
        if (hd == 1) {
            g -= 1;
            si -= 1;
            out = CigarElem{CigarOp::Del, 1};
            return;
        }
        Cost dd;
        if (sj > prev_block.j_range.e) {
            PA_ASSERT(sj == prev_block.j_range.e + 1, "diagonal edge case");
            dd = 1;
        } else {
            const auto d = prev_block.get_diff(sj - 1);
            PA_ASSERT(d.has_value(), "get_diff in range");
            dd = *d + hd;
        }
        if (dd == 1) {
            g -= 1;
            si -= 1;
            sj -= 1;
            out = CigarElem{CigarOp::Sub, 1};
            return;
        }
        engine_panic("PARENT NOT FOUND IN TRACEBACK");
    }

    // trace.rs:231-416
    bool dt_trace_block(I st_i, I st_j, Cost& g_st, const Block& prev_block, Cigar& cigar,
                        std::vector<BlockElem>& bl, I& out_i, I& out_j) {
        const uint8_t* a = be.a();
        const uint8_t* b = be.b();
        const I block_start = prev_block.i_range.e;
        auto index = [](Cost g, I d) { return (size_t)(g * g + g + d); };
        bl[0] = BlockElem{st_i, 0, 0};

        auto do_trace = [&](Cost g, I d) {  // inner fn trace, trace.rs:274-314
            out_i = block_start;
            out_j = st_j - (st_i - block_start) - d;
            g_st -= g;
            std::vector<CigarElem> ops;
            for (;;) {
                const BlockElem fr = bl[index(g, d)];
                if (fr.ext > 0) ops.push_back(CigarElem{CigarOp::Match, fr.ext});
                if (g == 0) break;
                g -= 1;
                d += fr.parent_d;
                const CigarOp op = fr.parent_d == -1 ? CigarOp::Ins : fr.parent_d == 0 ? CigarOp::Sub : CigarOp::Del;
                ops.push_back(CigarElem{op, 1});
            }
            for (size_t k = ops.size(); k-- > 0;) cigar.push_elem(ops[k]);
        };
        auto extend_and_check = [&](BlockElem& e, I j, Cost target_g) -> bool {  // trace.rs:319-336
            e.ext += extend_left(e.i, prev_block.i_range.e, j, a, b);
            if (e.i != prev_block.i_range.e) return false;
            const auto got = prev_block.get(j);
            return got.has_value() && *got == target_g;
        };

        Cost g = 0;
        if (extend_and_check(bl[0], st_j, g_st)) {
            do_trace(0, 0);
            return true;
        }
        I d_lo = 0, d_hi = 0;
        for (;;) {
            const Cost ng = g + 1;
            const size_t end_idx = index(ng, d_hi + 1);
            if (bl.size() <= end_idx) bl.resize(end_idx + 1);
            for (size_t k = index(ng, d_lo - 1); k <= end_idx; ++k) bl[k] = BlockElem{};
            for (I d = d_lo; d <= d_hi; ++d) {  // expand, trace.rs:351-364
                const BlockElem fr = bl[index(g, d)];
                auto update = [](BlockElem& x, I y, I pd) {
                    if (y < x.i) {
                        x.i = y;
                        x.parent_d = pd;
                    }
                };
                update(bl[index(ng, d - 1)], fr.i - 1, 1);
                update(bl[index(ng, d)], fr.i - 1, 0);
                update(bl[index(ng, d + 1)], fr.i, -1);
            }
            g += 1;
            d_lo -= 1;
            d_hi += 1;

            I min_fr = INT32_MAX, min_i = INT32_MAX;
            for (I d = d_lo; d <= d_hi; ++d) {  // extend, trace.rs:370-385
                BlockElem& fr = bl[index(g, d)];
                if (fr.i == INT32_MAX) continue;
                const I j = st_j - (st_i - fr.i) - d;
                if (extend_and_check(fr, j, g_st - g)) {
                    do_trace(g, d);
                    return true;
                }
                min_fr = std::min(min_fr, 2 * fr.i - d);
                min_i = std::min(min_i, fr.i);
            }
            if (g == params.max_g / 2 && min_i > (block_start + st_i) / 2) return false;
            if (g == params.max_g) return false;
            if (params.fr_drop > 0) {  // trace.rs:396-413
                auto bad = [&](I d) {
                    const I fi = bl[index(g, d)].i;
                    // 2*i - d in 64-bit: i may be I::MAX for unreachable diagonals
                    return fi <= block_start || (int64_t)2 * fi - d > (int64_t)min_fr + params.fr_drop;
                };
                while (d_lo < d_hi && bad(d_lo)) d_lo += 1;
                while (d_lo < d_hi && bad(d_hi)) d_hi -= 1;
                if (d_lo > d_hi) return false;
            }
        }
    }
};

// ---- domain.rs / lib.rs ------------------------------------------------------------------------------
struct AlignResult {
    Cost cost = 0;
    bool has_cigar = false;
    Cigar cigar;
    AstarPa2Stats stats;
};

template <class Backend>
class AstarPa2Instance {
   public:
    const AstarPa2Params& params;
    Backend& be;
    std::unique_ptr<Heuristic> heur;  // Some for Domain::Astar
    AstarPa2Stats stats;
    bool self_check = false;
    std::function<void(const Blocks<Backend>&, std::optional<Cost>, Cost)> pass_hook;  // test hook (never set by the library)

    AstarPa2Instance(const AstarPa2Params& p, Backend& backend) : params(p), be(backend) {  // lib.rs:87-120
        const double t0 = now_s();
        if (params.domain == DomainKind::Astar) {
            if (params.heuristic == HeuristicKind::Gap) heur = std::make_unique<GapCostH>(be.n(), be.m());
            else if (params.heuristic == HeuristicKind::SH)
                heur = std::make_unique<SeedHeuristicH>(be.a(), be.n(), be.b(), be.m(), params.heuristic_k, (int)params.heuristic_p);
            else if (params.heuristic == HeuristicKind::GCSH)  // Prune::Start iff params.prune (cli.rs:167-192)
                heur = std::make_unique<GcshHeuristic>(be.a(), be.n(), be.b(), be.m(), params.heuristic_k, (int)params.heuristic_p,
                                                       params.prune);
            else heur = std::make_unique<NoCostH>();
        }
        stats.t_precomp = now_s() - t0;
    }

    I blen() const { return be.m(); }
    I alen() const { return be.n(); }

    // domain.rs:77-246
    JRange j_range(IRange ir, std::optional<Cost> f_max_opt, const Block& prev, std::optional<JRange> old_range) {
        if (!f_max_opt) return JRange{0, blen()};
        const Cost f_max = *f_max_opt;
        const I is = ir.s, ie = ir.e;
        JRange range;
        switch (params.domain) {
            case DomainKind::Full:
                range = JRange{0, blen()};
                break;
            case DomainKind::GapStart:  // max_del_for_cost(s) = max_ins_for_cost(s) = s at unit cost
                range = JRange{is + 1 - f_max, ie + f_max};
                break;
            case DomainKind::GapGap: {
                const I d = blen() - alen();
                const Cost s = f_max - unit_gap_cost(0, 0, alen(), blen());
                // Rust `/` truncates toward zero; s may be negative here
                const I extra = s / 2;
                range = JRange{is + 1 + std::min(d, 0) - extra, ie + std::max(d, 0) + extra};
                break;
            }
            case DomainKind::Astar: {
                const double t0 = now_s();
                PA_ASSERT(prev.fixed_j_range.has_value(), "With A* Domain, fixed_j_range should always be set.");
                const I fixed_start = prev.fixed_j_range->s, fixed_end = prev.fixed_j_range->e;
                PA_ASSERT(fixed_start <= fixed_end, "Fixed range must not be empty");
                const I u0 = is, u1 = fixed_end;
                const Cost gu = is < 0 ? 0 : prev.index(fixed_end);
                I v0 = u0, v1 = u1;
                auto f = [&](I x, I y) -> Cost {
                    PA_ASSERT(y - u1 >= x - u0, "f: v below the diagonal of u");
                    return gu + unit_extend_cost(u0, u1, x, y) + heur->h(x, y);
                };
                if (!params.sparse_h) {  // domain.rs:171-181
                    while (v0 < ie) {
                        v0 += 1;
                        v1 += 1;
                        v1 += 1;
                        while (v1 <= blen() && f(v0, v1) <= f_max) v1 += 1;
                        v1 -= 1;
                    }
                } else {  // domain.rs:182-233
                    v0 += 1;
                    v1 += 1;
                    v1 += params.block_width;
                    v1 = std::min(v1, blen());
                    for (;;) {
                        if (v1 < v0 - u0 + u1) {
                            v1 = v0 - u0 + u1;
                            break;
                        }
                        const Cost fv = f(v0, v1);
                        if (fv <= f_max) {
                            if (v1 == blen()) break;
                            v1 += 8;
                            if (v1 >= blen()) v1 = blen();
                        } else {
                            v0 += div_ceil(fv - f_max, 2);
                            if (v0 > ie) {
                                v0 = ie;
                                break;
                            }
                        }
                    }
                    v0 = ie;
                    for (;;) {
                        if (v1 < v0 - u0 + u1) {
                            v1 = v0 - u0 + u1;
                            break;
                        }
                        const Cost fv = f(v0, v1);
                        if (fv <= f_max) break;
                        v1 -= div_ceil(fv - f_max, 2);
                    }
                }
                range = JRange{fixed_start, v1};
                stats.t_j_range += now_s() - t0;
                break;
            }
        }
        if (old_range) range = range.union_(*old_range);
        return range.intersection(JRange{0, blen()});
    }

    // domain.rs:251-350
    std::optional<JRange> fixed_j_range(I i, std::optional<Cost> f_max_opt, std::optional<JRange> prev_fixed,
                                        const Block& block) {
        if (params.domain != DomainKind::Astar) return std::nullopt;
        if (!f_max_opt) return std::nullopt;
        const Cost f_max = *f_max_opt;
        const double t0 = now_s();
        auto f = [&](I j) -> Cost { return block.index(j) + heur->h(i, j); };
        PA_ASSERT(prev_fixed.has_value(), "prev_fixed_j_range");
        PA_ASSERT(block.j_range.s <= prev_fixed->s, "block.j_range.0 <= prev_fixed.0");
        I start = prev_fixed->s;
        I end = std::min(block.original_j_range.e, blen());
        while (start <= end) {
            const Cost fv = f(start);
            if (fv <= f_max) break;
            start += params.sparse_h ? div_ceil(fv - f_max, 2) : 1;
        }
        while (end >= start) {
            const Cost fv = f(end);
            if (fv <= f_max) break;
            end -= params.sparse_h ? div_ceil(fv - f_max, 2) : 1;
        }
        JRange fixed{start, end};
        if (block.fixed_j_range) {
            if (fixed.is_empty()) fixed = *block.fixed_j_range;
            else fixed = fixed.union_(*block.fixed_j_range);
        }
        stats.t_fixed_j_range += now_s() - t0;
        return fixed;
    }

    // domain.rs:356-541.  Returns nullopt when no path was found for this bound.
    std::optional<std::pair<Cost, std::optional<Cigar>>> align_for_bounded_dist(std::optional<Cost> f_max, bool trace,
                                                                                Blocks<Backend>* blocks_in) {
        stats.f_max_tries += 1;
        if (params.prune && heur) {
            const double t0 = now_s();
            heur->update_contours();
            stats.t_contours_update += now_s() - t0;
        }
        std::unique_ptr<Blocks<Backend>> local;
        Blocks<Backend>* blocks = blocks_in;
        if (!blocks) {
            local = std::make_unique<Blocks<Backend>>(params.front, trace, be);
            local->self_check = self_check;
            blocks = local.get();
        }
        PA_ASSERT(f_max.value_or(0) >= 0, "f_max >= 0");

        Block dummy;
        dummy.fixed_j_range = JRange{-1, -1};
        const JRange initial_j_range = j_range(IRange{-1, 0}, f_max, dummy, blocks->next_block_j_range());
        if (initial_j_range.is_empty() || initial_j_range.s > 0) return std::nullopt;
        blocks->init(initial_j_range);
        blocks->set_last_block_fixed_j_range(initial_j_range);

        bool all_blocks_reused = true;
        const I n = alen();
        for (I i = 0; i < n; i += params.block_width) {
            const IRange ir{i, std::min(i + params.block_width, n)};
            const JRange jr = j_range(ir, f_max, blocks->last_block(), blocks->next_block_j_range());
            if (jr.is_empty()) {
                PA_ASSERT(!blocks->next_block_j_range().has_value(), "empty j_range with existing next block");
                return std::nullopt;
            }
            bool reuse = false;
            const auto nb = blocks->next_block_j_range();
            if (nb && *nb == jr && all_blocks_reused) reuse = true;
            all_blocks_reused = all_blocks_reused && reuse;

            const std::optional<JRange> prev_fixed = blocks->last_block().fixed_j_range;
            if (reuse) blocks->reuse_next_block(ir, jr);
            else blocks->compute_next_block(ir, jr);

            const std::optional<JRange> next_fixed = fixed_j_range(ir.e, f_max, prev_fixed, blocks->last_block());
            if (std::getenv("PA_ENGINE_DEBUG"))
                std::fprintf(stderr, "  i=(%d,%d] j_range=[%d,%d] reuse=%d fixed=[%d,%d] top=%d bot=%d\n", ir.s, ir.e, jr.s, jr.e, (int)reuse,
                             next_fixed ? next_fixed->s : -9, next_fixed ? next_fixed->e : -9, blocks->last_block().top_val, blocks->last_block().bot_val);
            if (next_fixed && next_fixed->is_empty()) return std::nullopt;
            blocks->set_last_block_fixed_j_range(next_fixed);

            if (params.prune && heur) {  // domain.rs:505-515
                const double t0 = now_s();
                const JRange inter = prev_fixed->intersection(*next_fixed);
                if (!inter.is_empty()) heur->prune_block(ir.s, ir.e, inter.s, inter.e);
                stats.t_pruning += now_s() - t0;
            }
        }

        const auto dist = blocks->last_block().get(blen());
        if (pass_hook) pass_hook(*blocks, f_max, dist.has_value() ? *dist : -1);  // tests: the blocks of a completed pass, before the trace pops them
        if (!dist) return std::nullopt;
        if (trace && *dist <= f_max.value_or(INT32_MAX)) {
            auto [cigar, tstats] = blocks->trace(0, 0, alen(), blen());
            stats.trace_stats = tstats;
            return std::make_pair(*dist, std::optional<Cigar>(std::move(cigar)));
        }
        return std::make_pair(*dist, std::optional<Cigar>());
    }
};

// band.rs:100-141 / 143-182 : shared search skeleton.
//
// Deviation (documented in DESIGN.md): the reference guards this loop with three sanity `assert!`s
// (band.rs:117-135).  They can fire on legal inputs -- a block reused across iterations keeps its old
// `original_j_range` (blocks.rs:190-197) which clips `fixed_j_range` (domain.rs:298), so with small block
// widths the band for f_max == d may miss the optimal path and a later, larger f_max finds
// cost <= last_s.  The reference aborts there; we keep every band decision identical, count the event in
// `sanity_violations`, and return the (exact, since cost <= s) answer instead of aborting.
template <class F>
std::pair<Cost, std::optional<Cigar>> band_search(Cost first_s, std::function<Cost(Cost)> next_s, F&& f,
                                                  size_t* sanity_violations) {
    Cost last_s = -1;
    Cost s = first_s;
    Cost maxs = COST_MAX;
    for (;;) {
        auto r = f(s);
        if (std::getenv("PA_ENGINE_DEBUG")) std::fprintf(stderr, "band_search s=%d -> %s %d\n", s, r ? "some" : "none", r ? r->first : -1);
        if (r) {
            const Cost cost = r->first;
            if (cost > maxs) *sanity_violations += 1;  // band.rs:118-121
            if (cost <= s) {
                if (cost <= last_s) *sanity_violations += 1;  // band.rs:123-126
                return *r;
            }
            maxs = std::min(maxs, cost);
        } else if (maxs != COST_MAX) {
            *sanity_violations += 1;  // band.rs:132-135
        }
        const Cost prev = s;
        last_s = s;
        s = std::min(next_s(s), maxs);
        if (s <= prev) s = next_s(prev);  // never stall (the reference's "potential infinite loop" TODO, band.rs:110)
    }
}

// lib.rs:122-175
template <class Backend>
AlignResult cost_or_align(const AstarPa2Params& params, Backend& be, bool trace, bool self_check = false,
                          std::function<void(const Blocks<Backend>&, std::optional<Cost>, Cost)> pass_hook = nullptr) {
    AstarPa2Instance<Backend> nw(params, be);
    nw.self_check = self_check;
    nw.pass_hook = std::move(pass_hook);
    const Cost h0 = nw.heur ? nw.heur->h(0, 0) : 0;
    AlignResult out;
    std::pair<Cost, std::optional<Cigar>> r;
    switch (params.doubling) {
        case DoublingKind::None: {
            PA_ASSERT(params.domain == DomainKind::Full, "DoublingType::None requires Domain::Full");
            auto x = nw.align_for_bounded_dist(std::nullopt, trace, nullptr);
            PA_ASSERT(x.has_value(), "unbounded alignment must succeed");
            r = *x;
            break;
        }
        case DoublingKind::LinearSearch:
        case DoublingKind::BandDoubling: {
            Cost start_f = 0, start_inc = 1;  // band.rs:13-23
            if (params.start == DoublingStart::Gap) {
                start_f = start_inc = unit_gap_cost(0, 0, be.n(), be.m());
            } else if (params.start == DoublingStart::H0) {
                start_f = h0;
                start_inc = 1;
            }
            Blocks<Backend> blocks(params.front, trace, be);
            blocks.self_check = self_check;
            auto f = [&](Cost s) { return nw.align_for_bounded_dist(s, trace, &blocks); };
            if (params.doubling == DoublingKind::LinearSearch) {
                const Cost delta = (Cost)params.delta;
                r = band_search(start_f, [delta](Cost s) { return s + delta; }, f, &nw.stats.sanity_violations);
            } else {
                start_inc = std::max(start_inc, params.block_width);  // lib.rs:142
                const float factor = params.factor;
                const Cost offset = start_f;
                r = band_search(offset + start_inc,
                                [factor, offset](Cost s) {  // band.rs:138
                                    return std::max((Cost)std::ceil(factor * (float)(s - offset)), 1) + offset;
                                },
                                f, &nw.stats.sanity_violations);
            }
            // Only the BandDoubling arm hands the Blocks' own counters to the caller (lib.rs:158); after a LinearSearch the
            // reference's block statistics stay at their defaults (lib.rs:132-140).  Found by oracle/astarpa2_restated.py.
            if (params.doubling == DoublingKind::BandDoubling) nw.stats.block_stats = blocks.stats;
            break;
        }
    }
    PA_ASSERT(h0 <= r.first, "Heuristic at start > final cost");
    out.cost = r.first;
    out.has_cigar = r.second.has_value();
    if (r.second) out.cigar = std::move(*r.second);
    out.stats = nw.stats;
    return out;
}

}  // namespace engine
}  // namespace pa
