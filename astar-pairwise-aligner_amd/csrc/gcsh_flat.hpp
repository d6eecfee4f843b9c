// gcsh_flat.hpp -- the GCSH contour layers as flat arrays and h(i, j) as plain integer code for host AND device.
//
// STATUS: groundwork for a device-side `full`, next to apa2_full_logic.hpp; not included by the shipped library.
// gcsh.hpp keeps the layers as vectors of points and tests "does layer v hold a point >= q" by walking the layer.  Here every layer
// is reduced to its Pareto front -- the points no other point of the layer dominates -- sorted by x ascending, which makes y strictly
// descending; then "some point with x >= q.x and y >= q.y" is "the first point with x >= q.x has y >= q.y": one binary search.  The
// score is a binary search over the layers (hint_contours.rs:258-272), so h costs O(log layers * log front) reads of three arrays
// that fit in one CU's LDS for a 100 kbp pair (a few thousand points).  oracle/apa2_full_emu.cpp cross-checks every h call of whole
// alignments against gcsh.hpp (tests/test_apa2_full_emu.py).
#pragma once
#include <stdint.h>

#include "sweep_logic.hpp"  // PA_HD

#include <algorithm>
#include <vector>

namespace pa {
namespace apa2 {

struct GcshFlat {
    const int32_t* layer_off;  // [nlayers + 1]; layer 0 is the sentinel (holds everything), layers 1.. hold match starts
    const int32_t* px;         // transformed start positions of the fronts, x ascending within a layer
    const int32_t* py;         // ... y strictly descending within a layer
    int32_t nlayers;           // including layer 0
    int32_t n, m, k, nseeds;
};

PA_HD int32_t gcsh_potential(const GcshFlat& g, int32_t i) {  // seeds.rs:34-71: seeds at 0, k, 2k, ... starting at >= i
    if (i < 0 || i > g.n) return 0;
    const int32_t before = (i + g.k - 1) / g.k;
    return before < g.nseeds ? g.nseeds - before : 0;
}

PA_HD bool gcsh_layer_contains(const GcshFlat& g, int32_t v, int32_t qx, int32_t qy) {
    if (v == 0) return true;
    int32_t lo = g.layer_off[v], hi = g.layer_off[v + 1];  // first point with x >= qx
    while (lo < hi) {
        const int32_t mid = lo + (hi - lo) / 2;
        if (g.px[mid] < qx) lo = mid + 1;
        else hi = mid;
    }
    return lo < g.layer_off[v + 1] && g.py[lo] >= qy;
}

PA_HD int32_t gcsh_score(const GcshFlat& g, int32_t qx, int32_t qy) {  // hint_contours.rs:258-272
    int32_t low = 0, high = g.nlayers;
    while (high - low > 1) {
        const int32_t mid = (low + high) / 2;
        if (gcsh_layer_contains(g, mid, qx, qy)) low = mid;
        else high = mid;
    }
    return low;
}

PA_HD int32_t gcsh_h(const GcshFlat& g, int32_t i, int32_t j) {  // csh.rs:341-350
    const int32_t pot = gcsh_potential(g, i);
    const int32_t val = gcsh_score(g, i - j - pot, j - i - pot);
    if (val == 0) {
        const int32_t d = (g.n - i) - (g.m - j);
        const int32_t gap = d < 0 ? -d : d;
        const int32_t pd = pot - gcsh_potential(g, g.n);
        return gap > pd ? gap : pd;
    }
    return pot - val;
}

// ---- prune_block (prune.rs:245-292) on flat arrays: the matches sorted by (start column, start row), one window record per seed ----
struct GcshSeedWindow {
    int32_t b0, b1;  // `before`: matches [b0, b1) of this seed not examined yet, lowest rows first
    int32_t a0, a1;  // `after` (a0 < 0: not split yet): matches above the rows of the seed's first visit
};
// Marks (active[t] = 0) the matches that start in columns i0 + 1 ..= i1 (seed starts) and rows j0 ..= j1; returns how many.
PA_HD int32_t gcsh_prune_block(const int32_t* mj, uint8_t* active, GcshSeedWindow* win, int32_t nseeds, int32_t k, int32_t i0, int32_t i1,
                               int32_t j0, int32_t j1) {
    int32_t pruned = 0;
    int32_t s = (i0 + 1 + k - 1) / k;  // first seed whose start s * k >= i0 + 1
    if (s < 0) s = 0;
    for (; s < nseeds && s * k <= i1; ++s) {
        GcshSeedWindow& w = win[s];
        if (w.a0 < 0) {
            int32_t a0 = w.b1;
            const int32_t a1 = w.b1;
            while (a0 >= w.b0 + 1 && mj[a0 - 1] > j1) {
                w.b1 -= 1;
                a0 -= 1;
            }
            w.a0 = a0;
            w.a1 = a1;
        }
        while (w.b1 > w.b0 && mj[w.b1 - 1] >= j0) {
            active[w.b1 - 1] = 0;
            w.b1 -= 1;
            pruned += 1;
        }
        while (w.a0 < w.a1 && mj[w.a0] <= j1) {
            active[w.a0] = 0;
            w.a0 += 1;
            pruned += 1;
        }
    }
    return pruned;
}

// Host: the flat arrays of a set of layers given as point lists (layer 0 = the sentinel, ignored).
struct GcshFlatStorage {
    std::vector<int32_t> layer_off, px, py;
    template <class Layers>  // Layers: indexable, each element iterable over points with .x / .y
    void build(const Layers& layers) {
        layer_off.assign(1, 0);
        px.clear();
        py.clear();
        layer_off.push_back(0);  // layer 0: no points stored
        std::vector<std::pair<int32_t, int32_t>> pts;
        for (size_t v = 1; v < layers.size(); ++v) {
            pts.clear();
            for (const auto& p : layers[v]) pts.emplace_back(p.x, p.y);
            // Pareto front: by x descending (y descending among equal x), keep a point iff its y beats every y seen so far
            std::sort(pts.begin(), pts.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second > b.second; });
            std::vector<std::pair<int32_t, int32_t>> front;
            int64_t best = INT64_MIN;
            for (const auto& p : pts)
                if ((int64_t)p.second > best) {
                    front.push_back(p);
                    best = p.second;
                }
            for (size_t t = front.size(); t-- > 0;) {  // x ascending, y descending
                px.push_back(front[t].first);
                py.push_back(front[t].second);
            }
            layer_off.push_back((int32_t)px.size());
        }
    }
    GcshFlat view(int32_t n, int32_t m, int32_t k, int32_t nseeds) const {
        GcshFlat g;
        g.layer_off = layer_off.data();
        g.px = px.data();
        g.py = py.data();
        g.nlayers = (int32_t)layer_off.size() - 1;
        g.n = n;
        g.m = m;
        g.k = k;
        g.nseeds = nseeds;
        return g;
    }
};

}  // namespace apa2
}  // namespace pa
