// The full-family per-pair program (csrc/apa2_full_logic.hpp) with the flat GCSH heuristic (csrc/gcsh_flat.hpp) running ON THE DEVICE --
// functionally: one thread per pair does everything, the rectangles by a scalar Myers word loop.  Not fast and not meant to be: it shows
// that the program, the heuristic probe and prune_block produce on the GPU what they produce on the host, before a wave-parallel
// backend replaces the scalar one.  The same backend code runs on the host with --host (no GPU needed), which is how it was written.
//   input : lines "<k> <p> <prune> <incremental> <sparse_h> <heuristic 0 none|1 gap|3 gcsh> <a> <b>"
//   output: one line per pair "status cost f_max_tries num_blocks num_incremental computed_lanes unique_lanes passes_on_device"
// Per pair and pass: a kernel launch runs ONE pass (PairProgFull::step); between two launches the host re-derives the contours from the
// match flags the device pruned (gcsh.hpp rebuild) and uploads the flat arrays again -- the orchestration of DESIGN.md 9 item 3.
// hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include -o tools/apa2_full_device_check tools/apa2_full_device_check.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../astar-pairwise-aligner_amd/csrc/engine.hpp"
#include "../astar-pairwise-aligner_amd/csrc/apa2_full_logic.hpp"
#include "../astar-pairwise-aligner_amd/csrc/gcsh_flat.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

using namespace pa::apa2;

// Everything a pair's program touches, as plain pointers (host memory with --host, device memory otherwise).
struct PairMem {
    int32_t n, m, wtot, nblk, heur;  // heur: 0 none, 1 gap, 3 gcsh
    const uint8_t* a_code;           // 0..3 per column
    const uint64_t* peq;             // [4][wtot]: rows of b equal to each letter
    FullRec* rec;                    // [nblk + 2]
    uint64_t* colp;                  // [nblk + 2][wtot] V.p
    uint64_t* colm;                  // [nblk + 2][wtot] V.m
    uint8_t* hrow;                   // [n] bit0 = +1, bit1 = -1 (incremental doubling)
    GcshFlat g;
    const int32_t* mj;               // matches (by start): rows
    uint8_t* active;
    GcshSeedWindow* win;
    int32_t nwin, prune_enabled;
};

struct NaiveBackend {
    PairMem& pm;
    PA_HD explicit NaiveBackend(PairMem& p) : pm(p) {}
    PA_HD FullRec load_rec(int32_t k) const { return pm.rec[k]; }
    PA_HD void store_rec(int32_t k, const FullRec& r) { pm.rec[k] = r; }
    PA_HD static int32_t pc(uint64_t x) { return (int32_t)__builtin_popcountll(x); }
    PA_HD int32_t index(int32_t k, const FullRec& r, int32_t j) const {
        if (k == 0) return j;
        if (j > r.je) return r.bot_val + (j - r.je);
        const uint64_t *p = pm.colp + (size_t)k * pm.wtot, *mm = pm.colm + (size_t)k * pm.wtot;
        int32_t v = r.top_val, j0 = r.js;
        while (j0 + 64 <= j) {
            v += pc(p[j0 / 64]) - pc(mm[j0 / 64]);
            j0 += 64;
        }
        if (j > j0) {
            const uint64_t mask = (1ull << (j - j0)) - 1;
            v += pc(p[j0 / 64] & mask) - pc(mm[j0 / 64] & mask);
        }
        return v;
    }
    PA_HD void set_word(int32_t k, int32_t w, uint64_t p, uint64_t m_) {
        pm.colp[(size_t)k * pm.wtot + w] = p;
        pm.colm[(size_t)k * pm.wtot + w] = m_;
    }
    PA_HD void copy_word(int32_t k, int32_t w) { set_word(k, w, pm.colp[(size_t)(k - 1) * pm.wtot + w], pm.colm[(size_t)(k - 1) * pm.wtot + w]); }
    PA_HD void init_plain(int32_t k, const FullRec& prev, const FullRec& cur) {
        for (int32_t w = cur.js / 64; w < cur.je / 64; ++w) {
            const bool in_prev = k > 1 && w * 64 >= prev.js && w * 64 < prev.je;
            if (in_prev) copy_word(k, w);
            else set_word(k, w, ~0ull, 0ull);
        }
    }
    PA_HD void init_preserve(int32_t k, const FullRec&, const FullRec& cur, int32_t p0, int32_t p1, int32_t prev_w1) {
        const int32_t w0 = cur.js / 64, w1 = cur.je / 64;
        const int32_t copy_end = w1 < prev_w1 ? w1 : prev_w1;
        for (int32_t w = w0; w < p0; ++w) copy_word(k, w);
        for (int32_t w = p1; w < copy_end; ++w) copy_word(k, w);
        for (int32_t w = copy_end > p1 ? copy_end : p1; w < w1; ++w) set_word(k, w, ~0ull, 0ull);
    }
    // myers.rs:27-55, word by word down a column, column by column
    PA_HD int32_t compute(int32_t k, int32_t i0, int32_t i1, int32_t w0, int32_t w1, int32_t mode) {
        uint64_t *vp = pm.colp + (size_t)k * pm.wtot, *vm = pm.colm + (size_t)k * pm.wtot;
        int32_t sum = 0;
        for (int32_t i = i0; i < i1; ++i) {
            uint64_t hp = 1, hm = 0;
            if (mode == kHInput || mode == kHUpdate) {
                hp = pm.hrow[i] & 1;
                hm = (pm.hrow[i] >> 1) & 1;
            }
            const uint64_t* eqrow = pm.peq + (size_t)pm.a_code[i] * pm.wtot;
            for (int32_t w = w0; w < w1; ++w) {
                uint64_t eq = eqrow[w];
                const uint64_t p = vp[w], mm = vm[w];
                const uint64_t vx = eq | mm;
                eq |= hm;
                const uint64_t hx = (((eq & p) + p) ^ p) | eq;
                uint64_t php = mm | ~(hx | p);
                uint64_t phm = p & hx;
                const uint64_t hpw = php >> 63, hmw = phm >> 63;
                php = (php << 1) | hp;
                phm = (phm << 1) | hm;
                hp = hpw;
                hm = hmw;
                vp[w] = phm | ~(vx | php);
                vm[w] = php & vx;
            }
            sum += (int32_t)hp - (int32_t)hm;
            if (mode == kHUpdate || mode == kHOutput) pm.hrow[i] = (uint8_t)(hp | (hm << 1));
        }
        return sum;
    }
    PA_HD int32_t h(int32_t i, int32_t j) {
        if (pm.heur == 1) {
            const int32_t d = (pm.n - i) - (pm.m - j);
            return d < 0 ? -d : d;
        }
        if (pm.heur == 3) return gcsh_h(pm.g, i, j);
        return 0;
    }
    PA_HD void prune_block(int32_t i0, int32_t i1, int32_t j0, int32_t j1) {
        if (pm.heur == 3 && pm.prune_enabled) gcsh_prune_block(pm.mj, pm.active, pm.win, pm.nwin, pm.g.k, i0, i1, j0, j1);
    }
    PA_HD void update_contours() {}  // (the host does it between two launches)
};

using Prog = PairProgFull<NaiveBackend>;

__global__ void begin_kernel(PairMem* pms, FullParams* sps, Prog::SearchState* sts, int npairs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npairs) return;
    NaiveBackend be(pms[t]);
    Prog prog(be, sps[t], pms[t].n, pms[t].m);
    prog.external_update = true;
    prog.begin(&sts[t]);
}
__global__ void step_kernel(PairMem* pms, FullParams* sps, Prog::SearchState* sts, int npairs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npairs || sts[t].done != 0) return;
    NaiveBackend be(pms[t]);
    Prog prog(be, sps[t], pms[t].n, pms[t].m);
    prog.external_update = true;
    prog.step(&sts[t]);
}

struct HostPair {
    std::string a, b;
    int k, p, prune, incremental, sparse_h, heur;
    std::unique_ptr<pa::engine::GcshHeuristic> gh;
    GcshFlatStorage flat;
    std::vector<uint8_t> a_code, active, hrow;
    std::vector<uint64_t> peq, colp, colm;
    std::vector<FullRec> rec;
    std::vector<int32_t> mj;
    std::vector<GcshSeedWindow> win;
    int wtot = 0, nblk = 0;
};

template <class T>
static T* to_device(const std::vector<T>& v) {
    T* d = nullptr;
    CK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    const bool host = argc > 2 && std::string(argv[2]) == "--host";
    if (argc < 2) {
        printf("usage: %s pairs.txt [--host]\n", argv[0]);
        return 2;
    }
    std::ifstream in(argv[1]);
    std::vector<HostPair> hp;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::istringstream ss(line);
        HostPair x;
        ss >> x.k >> x.p >> x.prune >> x.incremental >> x.sparse_h >> x.heur >> x.a >> x.b;
        hp.push_back(std::move(x));
    }
    const int P = (int)hp.size();
    std::vector<PairMem> pms(P);
    std::vector<FullParams> sps(P);
    std::vector<Prog::SearchState> sts(P);
    FullRec none;
    std::memset(&none, 0, sizeof none);
    none.js = none.je = none.ojs = none.oje = none.fs = none.fe = none.j_h = pa::sweep::kNone;
    auto code = [](char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; };
    for (int t = 0; t < P; ++t) {
        HostPair& x = hp[t];
        const int n = (int)x.a.size(), m = (int)x.b.size();
        x.wtot = (m + 63) / 64 + 1;
        x.nblk = (n + 255) / 256;
        x.a_code.resize(n);
        for (int i = 0; i < n; ++i) x.a_code[i] = (uint8_t)code(x.a[i]);
        x.peq.assign((size_t)4 * x.wtot, 0);
        for (int j = 0; j < m; ++j) x.peq[(size_t)code(x.b[j]) * x.wtot + j / 64] |= 1ull << (j % 64);
        x.rec.assign((size_t)x.nblk + 2, none);
        x.colp.assign((size_t)(x.nblk + 2) * x.wtot, ~0ull);
        x.colm.assign((size_t)(x.nblk + 2) * x.wtot, 0ull);
        x.hrow.assign((size_t)n + 1, 0);
        if (x.heur == 3) {
            x.gh = std::make_unique<pa::engine::GcshHeuristic>((const uint8_t*)x.a.data(), n, (const uint8_t*)x.b.data(), m, x.k, x.p, x.prune != 0);
            x.flat.build(x.gh->layers);
            for (const auto& mt : x.gh->by_start) x.mj.push_back(mt.j);
            x.active.assign(x.mj.size(), 1);
            for (const auto& ar : x.gh->active_range) x.win.push_back(GcshSeedWindow{(int32_t)ar.b0, (int32_t)ar.b1, -1, 0});
        }
        FullParams& sp = sps[t];
        sp.sparse_h = x.sparse_h;
        sp.prune = x.prune;
        sp.incremental = x.incremental;
        sp.doubling = 1;
        sp.start = 2;
        sp.factor = 2.0f;
        sp.delta = 1;
    }
    // memory image of every pair (host pointers or device copies)
    struct DevPtrs {
        uint8_t *a_code, *active, *hrow;
        uint64_t *peq, *colp, *colm;
        FullRec* rec;
        int32_t *mj, *off, *px, *py;
        GcshSeedWindow* win;
    };
    std::vector<DevPtrs> dp(P);
    auto fill_pm = [&](int t) {
        HostPair& x = hp[t];
        PairMem& pm = pms[t];
        pm.n = (int)x.a.size();
        pm.m = (int)x.b.size();
        pm.wtot = x.wtot;
        pm.nblk = x.nblk;
        pm.heur = x.heur;
        pm.nwin = (int)x.win.size();
        pm.prune_enabled = x.prune;
        if (host) {
            pm.a_code = x.a_code.data(); pm.peq = x.peq.data(); pm.rec = x.rec.data(); pm.colp = x.colp.data(); pm.colm = x.colm.data();
            pm.hrow = x.hrow.data(); pm.mj = x.mj.data(); pm.active = x.active.data(); pm.win = x.win.data();
            pm.g = x.flat.view(pm.n, pm.m, x.k, x.gh ? x.gh->nseeds : 0);
        } else {
            pm.a_code = dp[t].a_code; pm.peq = dp[t].peq; pm.rec = dp[t].rec; pm.colp = dp[t].colp; pm.colm = dp[t].colm;
            pm.hrow = dp[t].hrow; pm.mj = dp[t].mj; pm.active = dp[t].active; pm.win = dp[t].win;
            pm.g = x.flat.view(pm.n, pm.m, x.k, x.gh ? x.gh->nseeds : 0);
            pm.g.layer_off = dp[t].off; pm.g.px = dp[t].px; pm.g.py = dp[t].py;
        }
    };
    if (!host)
        for (int t = 0; t < P; ++t) {
            HostPair& x = hp[t];
            dp[t] = DevPtrs{to_device(x.a_code), to_device(x.active), to_device(x.hrow), to_device(x.peq), to_device(x.colp), to_device(x.colm),
                            to_device(x.rec),    to_device(x.mj),     to_device(x.flat.layer_off), to_device(x.flat.px), to_device(x.flat.py), to_device(x.win)};
        }
    for (int t = 0; t < P; ++t) fill_pm(t);
    PairMem* d_pms = nullptr;
    FullParams* d_sps = nullptr;
    Prog::SearchState* d_sts = nullptr;
    int launches = 0;
    std::vector<int> passes(P, 0);
    if (!host) {
        d_pms = to_device(pms);
        d_sps = to_device(sps);
        d_sts = to_device(sts);
        hipLaunchKernelGGL(begin_kernel, dim3((P + 63) / 64), dim3(64), 0, 0, d_pms, d_sps, d_sts, P);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(sts.data(), d_sts, P * sizeof(Prog::SearchState), hipMemcpyDeviceToHost));
    } else {
        for (int t = 0; t < P; ++t) {
            NaiveBackend be(pms[t]);
            Prog prog(be, sps[t], pms[t].n, pms[t].m);
            prog.external_update = true;
            prog.begin(&sts[t]);
        }
    }
    for (;;) {
        bool any = false;
        for (int t = 0; t < P; ++t) any = any || sts[t].done == 0;
        if (!any) break;
        for (int t = 0; t < P; ++t) passes[t] += sts[t].done == 0;
        if (!host) {
            hipLaunchKernelGGL(step_kernel, dim3((P + 63) / 64), dim3(64), 0, 0, d_pms, d_sps, d_sts, P);
            CK(hipDeviceSynchronize());
            launches += 1;
            CK(hipMemcpy(sts.data(), d_sts, P * sizeof(Prog::SearchState), hipMemcpyDeviceToHost));
        } else {
            for (int t = 0; t < P; ++t) {
                if (sts[t].done != 0) continue;
                NaiveBackend be(pms[t]);
                Prog prog(be, sps[t], pms[t].n, pms[t].m);
                prog.external_update = true;
                prog.step(&sts[t]);
            }
        }
        // between two launches: the contours of the pairs that go on, from the match flags the pass pruned
        for (int t = 0; t < P; ++t) {
            HostPair& x = hp[t];
            if (sts[t].done != 0 || x.heur != 3 || !x.prune) continue;
            if (!host) CK(hipMemcpy(x.active.data(), dp[t].active, x.active.size(), hipMemcpyDeviceToHost));
            bool changed = false;
            for (size_t q = 0; q < x.active.size(); ++q)
                if (x.gh->by_start[q].active != (x.active[q] != 0)) {
                    x.gh->by_start[q].active = x.active[q] != 0;
                    changed = true;
                }
            if (!changed) continue;
            x.gh->rebuild_contours();
            x.flat.build(x.gh->layers);
            if (!host) {
                CK(hipFree(dp[t].off)); CK(hipFree(dp[t].px)); CK(hipFree(dp[t].py));
                dp[t].off = to_device(x.flat.layer_off); dp[t].px = to_device(x.flat.px); dp[t].py = to_device(x.flat.py);
            }
            fill_pm(t);
            if (!host) CK(hipMemcpy(d_pms + t, &pms[t], sizeof(PairMem), hipMemcpyHostToDevice));
        }
    }
    for (int t = 0; t < P; ++t) {
        FullResult res;
        PairMem pm_host = pms[t];
        NaiveBackend be(pm_host);
        Prog prog(be, sps[t], pms[t].n, pms[t].m);
        prog.finish(sts[t], &res);  // (finish only reads the saved state)
        printf("%d %d %u %u %u %llu %llu %d\n", res.status, res.cost, res.f_max_tries, res.num_blocks, res.num_incremental_blocks,
               (unsigned long long)res.computed_lanes, (unsigned long long)res.unique_lanes, passes[t]);
    }
    fprintf(stderr, "%d pairs, %s, %d step launches\n", P, host ? "on the host" : "on the device", launches);
    return 0;
}
