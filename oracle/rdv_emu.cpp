/*
 * oracle/rdv_emu.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The rendezvous of half-wave blocks (astar-pairwise-aligner_amd/csrc/rdv_logic.hpp: two wavefronts of one workgroup run their two
 * blocks as ONE strip) WITHOUT a GPU: the protocol -- the very template the kernels instantiate over LDS atomics -- runs here over
 * std::atomic on host threads, one thread per wavefront, four threads per "workgroup", every thread running the product's per-pair
 * band-search program (csrc/apa2_logic.hpp) over the oracle's CPU kernels and pulling pairs from one ticket.  A block of at most 16
 * words goes through the rendezvous exactly as in apa2_kernel.hpp: its description into the wavefront's mail slot, arrive(); the taker
 * computes BOTH blocks (here: two calls of the CPU kernel, one of them on the partner's column and profile), finish().
 * What it checks: (a) with tests/test_rdv_emu.py, that cost and statistics of every pair equal the program run alone, whatever the
 * timing; (b) under ThreadSanitizer (`make -C oracle tsan_rdv`), that everything the two wavefronts share -- the partner's column, its
 * sum, the mail -- is ordered by the protocol's release / acquire pairs on the one shared word, which is what the device relies on.
 */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../astar-pairwise-aligner_amd/csrc/apa2_logic.hpp"
#include "../astar-pairwise-aligner_amd/csrc/rdv_logic.hpp"
#include "cpu_backend.hpp"

using namespace pa::engine;
using namespace pa::apa2;
using pa_oracle_cpu::CpuBackend;

namespace {

struct MailJob {  // what strip2_kernel.hpp's DualJob carries, in host terms
    CpuBackend* cb;
    V* col;  // words [w0, w1) of the block's column, absolute indexing
    int32_t i0, i1, w0, w1;
    int32_t* sum_out;
};

struct Workgroup {  // RdvShared of strip2_kernel.hpp
    std::atomic<uint32_t> st{0};
    std::atomic<uint32_t> live{0};
    MailJob mail[pa::rdv::kMaxWaves];
};

struct HostPolicy {  // the policy of rdv_logic.hpp over std::atomic (device: RdvLds)
    Workgroup* wg;
    uint32_t load() { return wg->st.load(std::memory_order_acquire); }
    bool cas(uint32_t expect, uint32_t desired) { return wg->st.compare_exchange_strong(expect, desired, std::memory_order_acq_rel, std::memory_order_acquire); }
    void add(uint32_t delta) { wg->st.fetch_add(delta, std::memory_order_acq_rel); }
    uint32_t live() { return wg->live.load(std::memory_order_acquire); }
    uint64_t now() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void nap() { std::this_thread::yield(); }
};

struct RdvBackend {  // oracle/apa2_emu.cpp's EmuBackend + the rendezvous in compute()
    CpuBackend& cb;
    pa::sweep::HeurParams hp;
    int sparse_h;
    size_t wtot;
    std::vector<BlockRec> rec;
    std::vector<std::vector<V>> col;
    BlockParams bp;
    Workgroup* wg;
    int wave;
    uint64_t patience_ns;
    pa::rdv::Counters* cnt;
    int32_t sum_slot = 0;
    bool stuck = false;

    RdvBackend(CpuBackend& c, const pa::sweep::HeurParams& h, int sh, int nblk, Workgroup* g, int w, uint64_t pat, pa::rdv::Counters* k)
        : cb(c), hp(h), sparse_h(sh), wg(g), wave(w), patience_ns(pat), cnt(k) {
        wtot = (size_t)(c.m() + 63) / 64;
        rec.resize((size_t)nblk + 2);
        col.assign((size_t)nblk + 2, std::vector<V>(wtot, V::one()));
        bp.simd = true;
        bp.no_ilp = false;
    }
    bool failed() const { return stuck; }
    void mark(int, uint32_t) const {}
    int32_t uniform(int32_t x) const { return x; }
    uint64_t strip_instructions() const { return 0; }
    BlockRec load_rec(int32_t k) const { return rec[(size_t)k]; }
    void store_rec(int32_t k, const BlockRec& r) { rec[(size_t)k] = r; }
    int32_t index(int32_t k, const BlockRec& r, int32_t j) const {
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
    static void run_job(const MailJob& j, const BlockParams& bp) {
        *j.sum_out = j.cb->compute(j.i0, j.i1, (size_t)j.w0, (size_t)j.w1, j.col + j.w0, HMode::None, bp);
    }
    int32_t compute(int32_t k, const BlockRec& prev, const BlockRec& cur, int32_t i0, int32_t i1) {
        const size_t w0 = (size_t)cur.js / 64, w1 = (size_t)cur.je / 64;
        for (size_t w = w0; w < w1; ++w) {
            const bool in_prev = k > 1 && (int32_t)(w * 64) >= prev.js && (int32_t)(w * 64) < prev.je;
            col[(size_t)k][w] = in_prev ? col[(size_t)k - 1][w] : V::one();
        }
        if (w1 == w0) return i1 - i0;
        MailJob mine{&cb, col[(size_t)k].data(), i0, i1, (int32_t)w0, (int32_t)w1, &sum_slot};
        if (wg && w1 - w0 <= 16) {  // half a wave: meet another wavefront's block (apa2_kernel.hpp compute / strip2_kernel.hpp rdv_strip)
            wg->mail[wave] = mine;
            HostPolicy pol{wg};
            int partner = -1;
            const int32_t r = pa::rdv::arrive(pol, wave, pa::rdv::kMaxWaves, patience_ns, 20ull * 1000000000ull, &partner, cnt);
            if (r == pa::rdv::kTook) {
                const MailJob theirs = wg->mail[partner];
                run_job(theirs, bp);  // (on the device: one strip, their block in lanes 0..31, ours in lanes 32..63)
                run_job(mine, bp);
                pa::rdv::finish(pol, partner);
                return sum_slot;
            }
            if (r == pa::rdv::kServed) return sum_slot;
            if (r == pa::rdv::kStuck) {
                stuck = true;
                return 0;
            }
        }
        run_job(mine, bp);
        return sum_slot;
    }
    int32_t f(int32_t k, const BlockRec& r, int32_t i, int32_t j) const { return index(k, r, j) + pa::sweep::heur_h(hp, i, j); }
    bool scan_first(int32_t k, const BlockRec& r, int32_t i, int32_t f_max, int32_t lo, int32_t hi, int32_t* out) {
        int32_t start = lo;
        while (start <= hi) {
            const int32_t fv = f(k, r, i, start);
            if (fv <= f_max) break;
            start += sparse_h ? pa::sweep::div_ceil_pos(fv - f_max, 2) : 1;
        }
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
        *out = end;
        return end >= lo;
    }
};

}  // namespace

// Runs the `simple` band search (GapCost, band doubling from h0) of `npairs` pairs on `groups` x 4 host threads.  patience_us < 0: no
// rendezvous at all (every block alone).  out[8 * i ..]: status, cost, f_max_tries, num_blocks, computed_lanes, unique_lanes, 0, 0.
// counters[4]: fused (took), served, alone, withdrawn.  Returns 0, or -1 on bad input.
extern "C" int pa_rdv_emu_run(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len, size_t npairs, int groups,
                              double patience_us, int64_t* out, uint64_t* counters) {
    if (groups < 1 || !out) return -1;
    std::vector<std::unique_ptr<Workgroup>> wgs;
    for (int g = 0; g < groups; ++g) {
        wgs.emplace_back(new Workgroup);
        wgs.back()->live.store(pa::rdv::kMaxWaves);
    }
    std::atomic<size_t> ticket{0};
    std::vector<pa::rdv::Counters> cnts((size_t)groups * pa::rdv::kMaxWaves);
    auto wave_main = [&](int g, int w) {
        Workgroup* wg = patience_us < 0 ? nullptr : wgs[(size_t)g].get();
        pa::rdv::Counters& cnt = cnts[(size_t)g * pa::rdv::kMaxWaves + (size_t)w];
        for (;;) {
            const size_t i = ticket.fetch_add(1);
            if (i >= npairs) break;
            int64_t* o = out + 8 * i;
            std::memset(o, 0, 8 * sizeof(int64_t));
            if (a_len[i] == 0 || b_len[i] == 0) {
                o[0] = kErrDegenerate;
                continue;
            }
            CpuBackend cb(a[i], a_len[i], b[i], b_len[i]);
            pa::sweep::HeurParams hp;
            hp.kind = pa::sweep::kHeurGap;
            hp.n = (int32_t)a_len[i];
            hp.m = (int32_t)b_len[i];
            hp.sh_h = nullptr;
            SearchParams sp;
            sp.heur = hp.kind;
            sp.sparse_h = 1;
            sp.doubling = kDoublingBand;
            sp.start = kStartH0;
            sp.factor = 2.0f;
            sp.delta = 1;
            RdvBackend be(cb, hp, 1, ((int)a_len[i] + 255) / 256, wg, w, (uint64_t)(patience_us * 1000.0), &cnt);
            PairProg<RdvBackend> prog(be, hp, sp);
            PairResult res;
            prog.run(&res);
            o[0] = res.status;
            o[1] = res.cost;
            o[2] = (int64_t)res.f_max_tries;
            o[3] = (int64_t)res.num_blocks;
            o[4] = (int64_t)res.computed_lanes;
            o[5] = (int64_t)res.unique_lanes;
        }
        wgs[(size_t)g]->live.fetch_sub(1, std::memory_order_acq_rel);  // (RdvLds::leave: a waiting block knows one candidate less)
    };
    std::vector<std::thread> th;
    for (int g = 0; g < groups; ++g)
        for (int w = 0; w < pa::rdv::kMaxWaves; ++w) th.emplace_back(wave_main, g, w);
    for (auto& t : th) t.join();
    if (counters) {
        counters[0] = counters[1] = counters[2] = counters[3] = 0;
        for (const auto& c : cnts) {
            counters[0] += c.took;
            counters[1] += c.served;
            counters[2] += c.alone;
            counters[3] += c.withdrawn;
        }
    }
    return 0;
}
