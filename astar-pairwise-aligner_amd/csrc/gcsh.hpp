// gcsh.hpp -- host-side restatement of the gap-chaining seed heuristic (GCSH) that A*PA2-full uses:
//   HeuristicParams{GCSH, k = 12, r = 1, p = 14, Prune::Start}          astarpa2/src/params.rs:102-109
// Paths below are relative to /root/reference/pa-heuristic/src/.
//
//   seeds            disjoint k-mers of a, potential(i) = #seeds starting at >= i        seeds.rs:34-71, matches/qgrams.rs:99-109
//   exact matches    hash of a's seed k-mers, all k-mers of b looked up in DECREASING j   matches/exact.rs:15-69, qgrams.rs:81-97
//   push filters     gap-transform filter T(start) <= T(target), then local pruning p     matches.rs:205-247, matches/prepruning.rs:95-203
//   transform        T(i,j) = (i - j - P(i), j - i - P(i)), componentwise partial order    seeds.rs:140-143
//   contours         layer(start) = 1 + score(T(end)); score(q) = max layer holding a point >= q
//                                                                                          contour/hint_contours.rs:213-272
//   h                P(u) - score(T(u)), or max(gap, potential) distance when the score is 0   heuristic/csh.rs:341-376
//   pruning          prune_block marks every match starting in i_range x j_range (two-pointer windows per seed),
//                    update_contours re-derives the layers                                 prune.rs:245-292, csh.rs:472-554
//
// The reference keeps the contours incrementally (HintContours<RotateToFrontContour>::update_layers).  Called as it is
// from A*PA2 -- update_layers(lowest_modified, Layer::MAX, .., right_of = 0), domain.rs:365-371 / csh.rs:525-545 -- that
// update re-scores every point of every layer from the lowest modified one upwards with chain_score, i.e. it reaches the
// same state as building the contours from the still-active matches.  We therefore rebuild them in update_contours();
// between two updates pruned matches are only marked (as in the reference), so h is identical at every call.
// `Pos` ordering comes from the un-vendored pa-types crate; componentwise `<=` is inferred from its uses
// (LexPos exists for the lexicographic order; contour/rotate_to_front.rs:33-47).
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

// NOTE: included from the middle of engine.hpp (after `Heuristic`, `I`, `Cost`, PA_ASSERT are defined).

namespace pa {
namespace engine {

struct GcshHeuristic : Heuristic {
    struct TP {  // transformed position
        I x, y;
    };
    static bool le(TP a, TP b) { return a.x <= b.x && a.y <= b.y; }

    struct Match {
        I i, j;  // start; end = (i+k, j+k); match_cost 0, seed_potential 1
        bool active;
    };
    struct ActiveRange {  // prune.rs:96-102
        I col;
        size_t b0, b1;  // `before`
        bool has_after;
        size_t a0, a1;  // `after`
    };

    const uint8_t* a;
    const uint8_t* b;
    I n, m, k;
    int p;
    bool prune_enabled;
    I nseeds = 0;
    TP t_target{0, 0};
    std::vector<Match> by_start;  // sorted by (i, j)
    std::vector<ActiveRange> active_range;
    std::vector<std::vector<TP>> layers;  // layers[0] = sentinel
    bool dirty = false;
    size_t num_matches_pushed = 0, num_matches_kept = 0;

    // seeds start at 0, k, 2k, ...: the number of them at >= i, in closed form (a table of n + 1 entries cost 0.2 ms per 100 kbp)
    Cost P(I i) const {
        if (i < 0 || i > n) return 0;
        const I before = (i + k - 1) / k;  // seeds starting before i
        return before < nseeds ? nseeds - before : 0;
    }
    TP T(I i, I j) const { return TP{i - j - P(i), j - i - P(i)}; }

    // ---- matches/prepruning.rs:24-66: extend_right / extend_right_simd (same control flow, scalar) ----
    bool extend_right(I& i, I j, I end_i) const {
        while (i < end_i && j < m && a[i] == b[j]) {
            ++i;
            ++j;
        }
        return i >= end_i;
    }
    bool extend_right_simd(I& i, I j, I end_i) const {
        if (i < n && j < m && a[i] == b[j]) {
            ++i;
            ++j;
        } else {
            return i >= end_i;
        }
        while (i < n - 32 && j < m - 32) {
            I cnt = 0;
            while (cnt < 32 && a[i + cnt] == b[j + cnt]) ++cnt;
            i += cnt;
            j += cnt;
            if (cnt < 32) return i >= end_i;
            if (i >= end_i) return true;
        }
        return extend_right(i, j, end_i);
    }

    // CenteredVec<I>, matches.rs:96-127 (default I::MAX)
    struct CenteredVec {
        std::vector<I> vec;
        explicit CenteredVec(I center) : vec((size_t)(2 * (int64_t)(center < 0 ? -center : center) + 1), INT32_MAX) {}
        I index(I idx) const {
            const int64_t pos = (int64_t)idx + (int64_t)(vec.size() / 2);
            if (pos < 0 || pos >= (int64_t)vec.size()) return INT32_MAX;
            return vec[(size_t)pos];
        }
        I& index_mut(I idx) {
            const int64_t aidx = idx < 0 ? -(int64_t)idx : idx;
            if (aidx > (int64_t)(vec.size() / 2)) {
                const size_t old_mid = vec.size() / 2;
                const size_t new_mid = std::max<size_t>((size_t)aidx, vec.size());
                const size_t grow = new_mid - old_mid;
                vec.insert(vec.begin(), grow, INT32_MAX);
                vec.insert(vec.end(), grow, INT32_MAX);
            }
            return vec[(size_t)((int64_t)idx + (int64_t)(vec.size() / 2))];
        }
    };

    // matches/prepruning.rs:95-203
    bool preserve_for_local_pruning(I si, I sj, std::vector<I>& fr, std::vector<I>& next_fr, const CenteredVec& next_match_per_diag) const {
        if (p == 0) return true;
        const I ei = si + k, ej = sj + k;
        const Cost start_pot = P(si);
        const I seed_idx = si / k;
        const I last_seed = std::min<I>(seed_idx + (I)p - 1, nseeds - 1);
        const I end_i = last_seed * k + k;
        const Cost end_pot = P(end_i);
        const size_t pd = (size_t)(start_pot - end_pot);
        fr.assign(2 * pd + 1, INT32_MIN);
        next_fr.assign(2 * pd + 1, INT32_MIN);
        size_t d0 = pd, d1 = pd + 1;  // d_range
        fr[pd] = ei;
        if (extend_right_simd(fr[pd], ej, end_i)) return true;
        if (next_match_per_diag.index(ei - ej) <= fr[pd]) return true;
        for (Cost g = 1; g < (Cost)pd; ++g) {  // 1 + match_cost .. pd
            fr[d0 - 1] = INT32_MIN;
            fr[d1] = INT32_MIN;
            next_fr[d0 - 1] = INT32_MIN;
            next_fr[d1] = INT32_MIN;
            for (size_t d = d0; d < d1; ++d) {
                next_fr[d - 1] = std::max(next_fr[d - 1], fr[d]);
                next_fr[d] = std::max(next_fr[d], fr[d] + 1);
                next_fr[d + 1] = std::max(next_fr[d + 1], fr[d] + 1);
            }
            std::swap(fr, next_fr);
            d0 -= 1;
            d1 += 1;
            while (d0 < d1 && g + P(fr[d0]) >= start_pot) d0 += 1;
            while (d0 < d1 && g + P(fr[d1 - 1]) >= start_pot) d1 -= 1;
            if (d0 >= d1) return false;
            for (size_t d = d0; d < d1; ++d) {
                I& i = fr[d];
                const I dd = ei - ej + ((I)d - (I)pd);
                const I j = i - dd;
                const I old_i = i;
                if (extend_right_simd(i, j, end_i)) return true;
                const I nm = next_match_per_diag.index(dd);
                if (old_i <= nm && nm <= i) return true;
            }
        }
        return false;
    }

    // build_layers = false: the matches and the per-seed windows only (the batched GPU path derives the contours on the device)
    // transform_filter = false: find_matches(a, b, config, false) as SH calls it (heuristic/sh.rs:48) -- local pruning without the gap transform's filter
    GcshHeuristic(const uint8_t* a_, I n_, const uint8_t* b_, I m_, I k_, int p_, bool prune_, bool build_layers = true, bool transform_filter = true)
        : a(a_), b(b_), n(n_), m(m_), k(k_ < 1 ? 1 : k_), p(p_), prune_enabled(prune_) {
        // seeds + potentials (qgrams.rs:99-109, seeds.rs:34-71)
        nseeds = n >= k ? (n - k) / k + 1 : 0;
        t_target = T(n, m);

        // exact matches: hash a's seeds, look up b's k-mers in decreasing j (exact.rs:15-69)
        auto bits = [](uint8_t c) -> uint64_t { return (uint64_t)((c >> 1) & 3); };
        std::vector<uint32_t> keys((size_t)nseeds);  // the k-mer of seed sidx (a[sidx * k ..])
        for (I sidx = 0; sidx < nseeds; ++sidx) {
            uint64_t q = 0;
            for (I t = 0; t < k; ++t) q = (q << 2) | bits(a[sidx * k + t]);
            keys[(size_t)sidx] = (uint32_t)q;
        }
        // open-addressing table: key -> the FIRST seed with that k-mer; seeds sharing a k-mer are chained in increasing order
        // (the order the reference's per-key vectors have).  A lookup per position of b: a binary search over the sorted seeds was
        // most of the 10 ms this constructor once took on a 100 kbp pair; sorting them at all another 0.3 ms.
        size_t tbits = 4;
        while (((size_t)1 << tbits) < 2 * keys.size() + 1) ++tbits;
        const size_t tmask = ((size_t)1 << tbits) - 1;
        std::vector<int32_t> slot(tmask + 1, -1), next_same((size_t)nseeds, -1);
        auto hash = [tbits](uint32_t key) { return (size_t)((key * 0x9E3779B1u) >> (32 - tbits)); };
        for (I sidx = nseeds - 1; sidx >= 0; --sidx) {
            const uint32_t key = keys[(size_t)sidx];
            size_t h = hash(key);
            while (slot[h] >= 0 && keys[(size_t)slot[h]] != key) h = (h + 1) & tmask;
            next_same[(size_t)sidx] = slot[h];  // (-1 when the key is new)
            slot[h] = sidx;
        }
        auto first_of = [&](uint32_t key) -> int32_t {
            for (size_t h = hash(key);; h = (h + 1) & tmask) {
                if (slot[h] < 0) return -1;
                if (keys[(size_t)slot[h]] == key) return slot[h];
            }
        };
        // in front of the table: one bit per hashed seed key, 8 KB (stays in L1).  Most positions of b match no seed at all and
        // stop here instead of walking the 128 KB table (the scan below: 1.7 -> 1.0 ms for a 100 kbp pair)
        constexpr unsigned kSieveBits = 16;
        std::vector<uint64_t> sieve((size_t)1 << (kSieveBits - 6), 0);
        auto sieve_of = [](uint32_t key) { return (uint32_t)((key * 0x85EBCA6Bu) >> (32 - kSieveBits)); };
        for (const uint32_t key : keys) {
            const uint32_t hb = sieve_of(key);
            sieve[hb >> 6] |= (uint64_t)1 << (hb & 63);
        }
        const TP tt = t_target;
        CenteredVec next_match_per_diag(tt.x - tt.y);  // MatchBuilder::new, matches.rs:166-185
        std::vector<I> fr, next_fr;
        if (m >= k && nseeds > 0) {
            const unsigned leftshift = 2u * (unsigned)(k - 1);
            uint64_t q = 0;
            // b_qgrams_rev: q >>= 2; q |= bits << leftshift, skipping the first k-1
            for (I pos = m - 1; pos >= 0; --pos) {
                q >>= 2;
                q |= bits(b[pos]) << leftshift;
                if (m - 1 - pos < k - 1) continue;
                const I j = pos;
                const uint32_t key = (uint32_t)q;
                const uint32_t hb = sieve_of(key);
                if (!((sieve[hb >> 6] >> (hb & 63)) & 1)) continue;
                for (int32_t sidx = first_of(key); sidx >= 0; sidx = next_same[(size_t)sidx]) {
                    const I i = (I)sidx * k;
                    num_matches_pushed += 1;
                    // MatchBuilder::push, matches.rs:205-247
                    if (transform_filter && !le(T(i, j), tt)) continue;               // transform filter
                    if (!preserve_for_local_pruning(i, j, fr, next_fr, next_match_per_diag)) continue;  // local pruning
                    if (p != 0) {
                        I& old = next_match_per_diag.index_mut(i - j);
                        PA_ASSERT(old >= i, "Matches should be added in reverse order on each diagonal");
                        old = i;
                    }
                    by_start.push_back(Match{i, j, true});
                }
            }
        }
        // sort by (LexPos(start), LexPos(end), cost) + dedup (matches.rs:253-305); CSHI::new retains T(start) <= T(target)
        // again (csh.rs:218-220, a no-op after the push filter)
        std::sort(by_start.begin(), by_start.end(), [](const Match& x, const Match& y) { return x.i != y.i ? x.i < y.i : x.j < y.j; });
        by_start.erase(std::unique(by_start.begin(), by_start.end(), [](const Match& x, const Match& y) { return x.i == y.i && x.j == y.j; }),
                       by_start.end());
        num_matches_kept = by_start.size();

        // MatchPruner::new (Prune::Start), prune.rs:132-201: one active range per seed
        {
            size_t idx = 0;
            active_range.reserve((size_t)nseeds);
            for (I sidx = 0; sidx < nseeds; ++sidx) {
                ActiveRange ar{sidx * k, idx, idx, false, 0, 0};
                while (idx < by_start.size() && by_start[idx].i == sidx * k) {
                    idx += 1;
                    ar.b1 = idx;
                }
                active_range.push_back(ar);
            }
        }
        if (build_layers) rebuild_contours();
    }

    bool layer_contains(size_t v, TP q) const {
        for (const TP& s : layers[v])
            if (le(q, s)) return true;
        return false;
    }

    // HintContours::score with max_len = r = 1, hint_contours.rs:258-272
    Cost score(TP q) const {
        size_t low = 0, high = layers.size();
        while (high - low > 1) {
            const size_t mid = (low + high) / 2;
            if (layer_contains(mid, q)) low = mid;
            else high = mid;
        }
        return (Cost)low;
    }

    // HintContours::new_with_filter over the active arrows whose end is <= T(target), in reverse start order
    // (csh.rs:246-252, hint_contours.rs:213-255)
    void rebuild_contours() {
        layers.clear();
        layers.emplace_back();
        layers[0].push_back(TP{INT32_MAX, INT32_MAX});
        for (size_t idx = by_start.size(); idx-- > 0;) {
            const Match& mt = by_start[idx];
            if (!mt.active) continue;
            const TP end = T(mt.i + k, mt.j + k);
            if (!le(end, t_target)) continue;
            const size_t v = (size_t)score(end) + 1;
            if (layers.size() <= v) layers.resize(v + 1);
            layers[v].push_back(T(mt.i, mt.j));
        }
        dirty = false;
    }

    // csh.rs:341-350
    Cost h(I i, I j) const override {
        const Cost pot = P(i);
        const Cost val = score(T(i, j));
        if (val == 0) {
            // distance(pos, target) = max(gap, potential_distance): csh.rs:178-187, seeds.rs:84-89
            const int64_t d = (int64_t)(n - i) - (int64_t)(m - j);
            const Cost gap = (Cost)(d < 0 ? -d : d);
            return std::max(gap, pot - P(n));
        }
        return pot - val;
    }

    // MatchPruner::prune_block, prune.rs:245-292 (both ranges inclusive; called as i_range.0..i_range.1 and
    // intersection.0..intersection.1, domain.rs:505-515)
    void prune_block(I i_start, I i_end, I j_start, I j_end) override {
        if (!prune_enabled) return;
        PA_ASSERT(j_start <= j_end, "prune_block: j_range");
        size_t seed_idx = (size_t)(std::lower_bound(active_range.begin(), active_range.end(), i_start + 1,
                                                    [](const ActiveRange& ar, I col) { return ar.col < col; }) -
                                   active_range.begin());
        while (seed_idx < active_range.size() && active_range[seed_idx].col <= i_end) {
            ActiveRange& ar = active_range[seed_idx];
            if (!ar.has_after) {
                size_t a0 = ar.b1, a1 = ar.b1;
                while (a0 >= ar.b0 + 1 && by_start[a0 - 1].j > j_end) {
                    ar.b1 -= 1;
                    a0 -= 1;
                }
                ar.has_after = true;
                ar.a0 = a0;
                ar.a1 = a1;
            }
            while (ar.b1 > ar.b0 && by_start[ar.b1 - 1].j >= j_start) {
                by_start[ar.b1 - 1].active = false;
                dirty = true;
                ar.b1 -= 1;
            }
            while (ar.a0 < ar.a1 && by_start[ar.a0].j <= j_end) {
                by_start[ar.a0].active = false;
                dirty = true;
                ar.a0 += 1;
            }
            seed_idx += 1;
        }
    }

    // csh.rs:497-554 (see the header comment)
    void update_contours() override {
        if (dirty) rebuild_contours();
    }
};

}  // namespace engine
}  // namespace pa
