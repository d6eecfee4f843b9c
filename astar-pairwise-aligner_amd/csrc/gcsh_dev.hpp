// gcsh_dev.hpp -- the gap-chaining seed heuristic (GCSH) of A*PA2-full in the form the GPU keeps it: plain arrays per pair, integer
// code for host AND device.  apa2_full_kernel.hpp runs it wave-parallel (64 contour layers per probe round, one lane per seed when a
// block's matches are pruned); oracle/apa2_full_emu.cpp runs the same functions on the CPU next to gcsh.hpp and counts disagreements
// at every h call, every prune_block and every re-derivation of the contours (tests/test_apa2_full_emu.py).
// Paths below are relative to /root/reference/pa-heuristic/src/.
//
//   transform   T(i, j) = (i - j - P(i), j - i - P(i)), P(i) = #seeds starting at >= i        seeds.rs:34-71,140-143
//   contours    layer(start) = 1 + score(T(end)), matches taken from the last start to the first; score(q) = the highest layer
//               that holds a point >= q componentwise                                          contour/hint_contours.rs:213-272
//   h           P(u) - score(T(u)), or max(gap, potential) distance when the score is 0        heuristic/csh.rs:341-376
//   pruning     prune_block marks the matches starting in i_range x j_range (two windows per seed); the contours are re-derived
//               at the start of the next pass                                                  prune.rs:245-292, csh.rs:472-554
//
// Layout.  The matches of a pair sorted by (start column, start row): mi[t], mj[t], active[t].  A contour layer is a linked list of
// the transformed starts it holds, newest first: the newest element sits INLINE in the layer's record (lrec[v] = {x, y, next}), older
// ones in cells of `cell` (one per match, the match's own index) -- almost every layer of a real alignment holds ONE point (the
// chain along the alignment), so "does layer v hold a point >= q" is one 16-byte load.  Layers are nested (a point >= q in layer v
// implies one in layer v - 1: the next match of its chain), so the score is a search for the boundary and any search order finds it.
#pragma once
#include <stdint.h>

#include "sweep_logic.hpp"  // PA_HD

namespace pa {
namespace apa2 {

struct GcshCell {
    int32_t x, y;   // transformed start
    int32_t next;   // index into `cell` of the next older point of the same layer, or -1
    int32_t pad;
};
static_assert(sizeof(GcshCell) == 16, "GcshCell layout");

struct GcshSeedWindow {  // prune.rs:96-102, as indices into the matches
    int32_t b0, b1;  // `before`: matches [b0, b1) of this seed not examined yet, lowest rows first
    int32_t a0, a1;  // `after` (a0 < 0: not split yet): matches above the rows of the seed's first visit
};
static_assert(sizeof(GcshSeedWindow) == 16, "GcshSeedWindow layout");

struct GcshDev {
    const int32_t* mi;     // [nmatch] start columns (multiples of k)
    const int32_t* mj;     // [nmatch] start rows
    uint8_t* active;       // [nmatch] 1 = not pruned
    GcshSeedWindow* win;   // [nseeds]
    GcshCell* lrec;        // [nmatch + 2] layer records; layer 0 is the sentinel (holds everything)
    GcshCell* cell;        // [nmatch]
    int32_t nmatch;
    int32_t nlayers;       // including layer 0 (state: set by the contour build)
    int32_t n, m, k, nseeds;
    int32_t prune;         // prune_block is live (AstarPa2Params::prune)
    int32_t pad;
};

PA_HD int32_t gd_potential(int32_t n, int32_t k, int32_t nseeds, int32_t i) {  // seeds at 0, k, 2k, ...: those starting at >= i
    if (i < 0 || i > n) return 0;
    const int32_t before = (i + k - 1) / k;
    return before < nseeds ? nseeds - before : 0;
}
PA_HD int32_t gd_potential(const GcshDev& g, int32_t i) { return gd_potential(g.n, g.k, g.nseeds, i); }
PA_HD int32_t gd_tx(const GcshDev& g, int32_t i, int32_t j) { return i - j - gd_potential(g, i); }
PA_HD int32_t gd_ty(const GcshDev& g, int32_t i, int32_t j) { return j - i - gd_potential(g, i); }

// "layer v holds a point >= (qx, qy)" (rotate_to_front.rs:33-47; the order of the points does not matter)
PA_HD bool gd_contains(const GcshDev& g, int32_t v, int32_t qx, int32_t qy) {
    if (v == 0) return true;
    GcshCell c = g.lrec[v];
    for (;;) {
        if (c.x >= qx && c.y >= qy) return true;
        if (c.next < 0) return false;
        c = g.cell[c.next];
    }
}

// hint_contours.rs:258-272 (the scalar form: host, emulation; the device searches 64 layers per round)
PA_HD int32_t gd_score_scalar(const GcshDev& g, int32_t qx, int32_t qy) {
    int32_t low = 0, high = g.nlayers;
    while (high - low > 1) {
        const int32_t mid = (low + high) / 2;
        if (gd_contains(g, mid, qx, qy)) low = mid;
        else high = mid;
    }
    return low;
}

// csh.rs:341-350 given the score of T(i, j)
PA_HD int32_t gd_h_from_score(const GcshDev& g, int32_t i, int32_t j, int32_t val) {
    const int32_t pot = gd_potential(g, i);
    if (val == 0) {  // distance(pos, target) = max(gap, potential distance): csh.rs:178-187, seeds.rs:84-89
        const int32_t d = (g.n - i) - (g.m - j);
        const int32_t gap = d < 0 ? -d : d;
        const int32_t pd = pot - gd_potential(g, g.n);
        return gap > pd ? gap : pd;
    }
    return pot - val;
}

// One more point for layer v (v == nlayers: a new layer).  `t` = the match's index = the cell that takes the layer's previous newest.
PA_HD void gd_insert(GcshDev& g, int32_t v, int32_t t, int32_t x, int32_t y) {
    GcshCell rec;
    rec.x = x;
    rec.y = y;
    rec.next = -1;
    rec.pad = 0;
    if (v < g.nlayers) {
        g.cell[t] = g.lrec[v];
        rec.next = t;
    } else {
        g.nlayers = v + 1;
    }
    g.lrec[v] = rec;
}

// HintContours::new_with_filter over the active arrows whose end is <= T(target), from the last start to the first
// (csh.rs:246-252, hint_contours.rs:213-255).  Score: (g, qx, qy) -> the highest layer that holds a point >= q.
template <class Score>
PA_HD void gd_build_contours(GcshDev& g, Score&& score) {
    g.nlayers = 1;
    const int32_t ttx = gd_tx(g, g.n, g.m), tty = gd_ty(g, g.n, g.m);
    for (int32_t t = g.nmatch - 1; t >= 0; --t) {
        if (!g.active[t]) continue;
        const int32_t i = g.mi[t], j = g.mj[t];
        const int32_t ex = gd_tx(g, i + g.k, j + g.k), ey = gd_ty(g, i + g.k, j + g.k);
        if (!(ex <= ttx && ey <= tty)) continue;
        const int32_t v = score(g, ex, ey) + 1;
        gd_insert(g, v, t, gd_tx(g, i, j), gd_ty(g, i, j));
    }
}

// MatchPruner::prune_block for ONE seed (prune.rs:245-292; both ranges inclusive): marks the matches of seed s that start in rows
// j0 ..= j1.  Returns how many it marked.  (The device runs one lane per seed of the block.)
PA_HD int32_t gd_prune_seed(const GcshDev& g, int32_t s, int32_t j0, int32_t j1) {
    GcshSeedWindow w = g.win[s];
    int32_t pruned = 0;
    if (w.a0 < 0) {
        int32_t a0 = w.b1;
        const int32_t a1 = w.b1;
        while (a0 >= w.b0 + 1 && g.mj[a0 - 1] > j1) {
            w.b1 -= 1;
            a0 -= 1;
        }
        w.a0 = a0;
        w.a1 = a1;
    }
    while (w.b1 > w.b0 && g.mj[w.b1 - 1] >= j0) {
        g.active[w.b1 - 1] = 0;
        w.b1 -= 1;
        pruned += 1;
    }
    while (w.a0 < w.a1 && g.mj[w.a0] <= j1) {
        g.active[w.a0] = 0;
        w.a0 += 1;
        pruned += 1;
    }
    g.win[s] = w;
    return pruned;
}
// The seeds whose start column lies in i0 + 1 ..= i1: [first, end).
PA_HD void gd_block_seeds(const GcshDev& g, int32_t i0, int32_t i1, int32_t* first, int32_t* end) {
    int32_t s = (i0 + 1 + g.k - 1) / g.k;
    if (s < 0) s = 0;
    int32_t e = i1 / g.k + 1;  // seeds with s * k <= i1
    if (e > g.nseeds) e = g.nseeds;
    *first = s;
    *end = e > s ? e : s;
}
PA_HD int32_t gd_prune_block(const GcshDev& g, int32_t i0, int32_t i1, int32_t j0, int32_t j1) {
    int32_t s0, s1, pruned = 0;
    gd_block_seeds(g, i0, i1, &s0, &s1);
    for (int32_t s = s0; s < s1; ++s) pruned += gd_prune_seed(g, s, j0, j1);
    return pruned;
}

}  // namespace apa2
}  // namespace pa
