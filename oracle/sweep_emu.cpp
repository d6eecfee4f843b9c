/*
 * oracle/sweep_emu.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Runs the product's device-side A*PA2 sweep (astar-pairwise-aligner_amd/csrc/sweep_wave.hpp + sweep_host.hpp) WITHOUT a
 * GPU: the wave program is instantiated over the array-emulated wavefront of host_wave.hpp and every wavefront is a host
 * thread, so the band logic, the block-boundary bookkeeping and the hand-off protocol between strips are exercised by the
 * CPU test-suite.  Expected values come from the host-driven engine over the oracle kernels (engine_cpu.cpp).
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../astar-pairwise-aligner_amd/csrc/sweep_host.hpp"
#include "cpu_backend.hpp"
#include "host_wave.hpp"

using namespace pa::engine;
using namespace pa::sweep;
using pa_host_wave::HostWave;
using pa_oracle_cpu::CpuBackend;

namespace {

struct HostLauncher {
    static constexpr int kSlots = 8;  // (two more than passes in flight, like the device launcher)
    const uint8_t *a, *b;
    int32_t n = 0, m = 0, nblk = 0;
    const int32_t* sh_h = nullptr;
    bool trace = false;
    int nwaves = 4;
    int in_flight = 3;  // passes launched together at most (PA_SWEEP_EMU_DEPTH)
    uint32_t pass_id = 0;
    uint32_t spin_limit = 1u << 24;
    int32_t heur_kind = kHeurGap;
    std::vector<uint32_t> codes, prof;
    std::vector<BlockRec> merged0;  // "no block exists yet" 
    // One pass in flight: its own records and buffers, its wavefront threads, and a closer thread that merges its records
    // into the older ones and publishes the done word (what sweep_merge_kernel does behind the launch).
    struct Slot {
        int seq = 0, prev_seq = 0;
        std::vector<BlockRec> merged;  // the earlier passes' records with this pass's on top
        uint32_t pass = 0;
        std::vector<BRec> brec;
        std::vector<TRec> trec;
        std::vector<uint64_t> strip_start, pring, gran, col;
        uint64_t bprog = 0;
        struct {
            uint64_t cancel;  // directly before the status block (the wave program fetches both with one load)
            Status status;
        } ctl;
        uint32_t ticket = 0;
        uint64_t done = 0;
        PassGeometry geo;
        Ctx c;
        std::vector<std::thread> waves;
        std::thread closer;
        bool live = false;
    };
    Slot slots[kSlots];
    Status last_status;

    ~HostLauncher() { cancel_after(0); }
    Slot& slot_of(int seq) { return slots[seq % kSlots]; }
    int max_in_flight() const { return in_flight; }
    // (the product's policy for short pairs; PA_SWEEP_EMU_CANCEL_NOWAIT=1 runs the emulation that way, so that the hosts's bookkeeping of
    //  passes that are cancelled but not yet waited for is exercised under the thread sanitizer too)
    bool cancel_without_waiting() const {
        static const bool on = std::getenv("PA_SWEEP_EMU_CANCEL_NOWAIT") != nullptr;
        return on;
    }
    int pass_waves(int32_t) const { return 1; }
    int wave_budget() const { return 1 << 20; }

    void begin_pair(int32_t n_, int32_t m_, int32_t nblk_, const int32_t* sh, bool tr) {
        n = n_;
        m = m_;
        nblk = nblk_;
        sh_h = sh;
        trace = tr;
        if (const char* e = std::getenv("PA_SWEEP_EMU_DEPTH")) in_flight = std::atoi(e) < 1 ? 1 : (std::atoi(e) > kSlots - 2 ? kSlots - 2 : std::atoi(e));
        codes.assign((size_t)(n + 15) / 16 + 16, 0);
        for (int32_t i = 0; i < n; ++i) {
            const uint8_t ch = a[i];
            const uint32_t r = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
            codes[(size_t)i / 16] |= r << (2 * (i % 16));
        }
        const size_t wt = (size_t)(m + 63) / 64;
        std::vector<pa_bits_t> pa_(n ? n : 1), pb_(wt ? wt : 1);
        pa_or_bitprofile_build(a, (size_t)n, b, (size_t)m, pa_.data(), pb_.data());
        prof.assign(wt * 4 + 4, 0);
        std::memcpy(prof.data(), pb_.data(), wt * 16);
        BlockRec none;
        none.js = none.je = none.ojs = none.oje = none.fs = none.fe = kNone;
        none.top_val = none.bot_val = 0;
        merged0.assign((size_t)nblk + 2, none);
        for (Slot& sl : slots) {
            sl.merged.assign((size_t)nblk + 2, none);
            sl.ctl.cancel = 0;
            sl.done = 0;
            sl.brec.assign((size_t)nblk + 2, BRec{});
            sl.trec.assign((size_t)nblk + 2, TRec{});
        }
    }
    BlockRec read_merged(int seq, int32_t k) {
        if (seq != 0 && slot_of(seq).seq != seq) std::abort();
        return seq == 0 ? merged0[(size_t)k] : slot_of(seq).merged[(size_t)k];
    }

    void launch_pass(int seq, int prev_seq, int32_t f_max, int32_t sparse_h, const PassInit& init) {
        Slot& sl = slot_of(seq);
        if (sl.live) std::abort();  // (the aligner never has more than kSlots - 2 passes in flight)
        Slot* pv = prev_seq ? &slot_of(prev_seq) : nullptr;
        if (pv && (pv == &sl || pv->seq != prev_seq)) std::abort();
        const bool pv_running = pv && pv->live;
        const std::vector<BlockRec>* merged_in = pv ? &pv->merged : &merged0;
        pass_id += 1;
        sl.seq = seq;
        sl.prev_seq = prev_seq;
        sl.pass = pass_id;
        sl.geo = pass_geometry(n, m, f_max);
        const PassGeometry& geo = sl.geo;
        sl.strip_start.resize((size_t)geo.nstrips, 0);
        sl.pring.resize((size_t)geo.nstrips * geo.pr_stride, 0);
        sl.gran.assign((size_t)geo.nstrips * geo.gran_stride, 0);
        const size_t nslots = trace ? (size_t)nblk + 1 : (size_t)geo.col_ring;
        sl.col.assign(nslots * (size_t)geo.col_stride * 2, 0);
        std::memset(&sl.ctl.status, 0, sizeof(Status));
        sl.ticket = 0;
        const uint32_t t1 = blk_tag(sl.pass, 1);
        BRec& b1 = sl.brec[1];
        b1.js = tw_make(t1, init.js1);
        b1.je = tw_make(t1, init.je1);
        b1.ojs = tw_make(t1, init.ojs1);
        b1.oje = tw_make(t1, init.oje1);
        b1.flags = tw_make(t1, init.flags1);
        b1.smax = tw_make(t1, init.last_strip);
        b1.specmax = tw_make(t1, 0);
        TRec& tr1 = sl.trec[1];
        tr1.js = tw_make(t1, init.js1);
        tr1.top_val = tw_make(t1, init.top1);
        tr1.fs_prev = tw_make(t1, init.fs0);
        tr1.lim = tw_make(t1, 0);
        tr1.found = tw_make(t1, 0);
        tr1.state = tw_make(t1, kTDesc);
        for (int32_t r = 0; r <= init.last_strip && r < geo.nstrips; ++r) sl.strip_start[(size_t)r] = tw_make(sl.pass, 1);
        sl.bprog = tw_make(t1, init.oje1);

        Ctx& c = sl.c;
        c.a_codes = codes.data();
        c.b_prof = prof.data();
        c.n = n;
        c.m = m;
        c.nblk = nblk;
        c.wtot = geo.wtot;
        c.f_max = f_max;
        c.pass = sl.pass;
        c.heur = sh_h ? kHeurSH : heur_kind;
        c.sparse_h = sparse_h;
        c.sh_h = sh_h;
        c.store_cols = trace ? 1 : 0;
        c.d_old = merged_in->data();
        c.prev_brec = pv_running ? pv->brec.data() : nullptr;
        c.prev_pass = pv_running ? pv->pass : 0;
        c.prev_done = pv_running ? &pv->done : &sl.done;
        c.cancel = &sl.ctl.cancel;
        c.brec = sl.brec.data();
        c.trec = sl.trec.data();
        c.bprog = &sl.bprog;
        c.strip_start = sl.strip_start.data();
        c.pring = sl.pring.data();
        c.pr_stride = geo.pr_stride;
        c.gran = sl.gran.data();
        c.gran_stride = geo.gran_stride;
        c.win = geo.win;
        c.col = sl.col.data();
        c.col_stride = geo.col_stride;
        c.col_ring = geo.col_ring;
        c.status = &sl.ctl.status;
        c.ticket = &sl.ticket;
        c.nstrips = geo.nstrips;
        c.nwaves = nwaves < geo.nstrips ? nwaves : geo.nstrips;
        c.spin_limit = spin_limit;
        c.timing = nullptr;
        sl.live = true;
        for (int w = 0; w < c.nwaves; ++w) sl.waves.emplace_back([&c]() { wave_main<HostWave>(c); });
        // behind the pass: merge (after the previous pass's merge, like the stream-ordered kernels on the device), then the done word
        const uint64_t prev_pass_id = pv ? pv->pass : 0;
        sl.closer = std::thread([this, &sl, pv, pv_running, prev_pass_id, merged_in]() {
            for (auto& t : sl.waves) t.join();
            if (pv_running)
                while (__atomic_load_n(&pv->done, __ATOMIC_ACQUIRE) != prev_pass_id) std::this_thread::yield();
            const Status& st = sl.ctl.status;
            const std::vector<BlockRec>& mo = *merged_in;
            std::vector<BlockRec>& mn = sl.merged;
            const bool ended = st.state == kStDone || st.state == kStNoPath;
            for (int32_t k = 0; k <= nblk + 1; ++k) {
                BlockRec d = mo[(size_t)k];
                if (ended && k >= 1 && k <= nblk && k <= st.k_end) {
                    const BRec& s = sl.brec[(size_t)k];
                    d.js = tw_val(s.js);
                    d.je = tw_val(s.je);
                    d.ojs = tw_val(s.ojs);
                    d.oje = tw_val(s.oje);
                    if (k <= st.k_fixed) {
                        d.fs = tw_val(s.fs);
                        d.fe = tw_val(s.fe);
                        d.top_val = tw_val(s.top_val);
                        d.bot_val = tw_val(s.bot_val);
                    }
                }
                mn[(size_t)k] = d;
            }
            __atomic_store_n(&sl.done, (uint64_t)sl.pass, __ATOMIC_RELEASE);
        });
    }
    Status wait_pass(int seq) {
        Slot& sl = slot_of(seq);
        if (sl.closer.joinable()) sl.closer.join();
        sl.waves.clear();
        sl.live = false;
        last_status = sl.ctl.status;
        if (std::getenv("PA_SWEEP_DEBUG")) {
            const Status& st = sl.ctl.status;
            std::fprintf(stderr, "pass %u (seq %d after %d) state=%u value=%d k_end=%d k_fixed=%d\n", sl.pass, seq, sl.prev_seq, st.state, st.value, st.k_end, st.k_fixed);
            for (int32_t k = 1; k <= st.k_end && k <= nblk; ++k) {
                const BRec& s = sl.brec[(size_t)k];
                std::fprintf(stderr, "  i=(%d,%d] j_range=[%d,%d] fixed=[%d,%d] top=%d bot=%d\n", (k - 1) * kBlockW, k * kBlockW < n ? k * kBlockW : n,
                             tw_val(s.ojs), tw_val(s.oje), k <= st.k_fixed ? tw_val(s.fs) : -9, k <= st.k_fixed ? tw_val(s.fe) : -9,
                             k <= st.k_fixed ? tw_val(s.top_val) : -9, k <= st.k_fixed ? tw_val(s.bot_val) : -9);
            }
        }
        return sl.ctl.status;
    }
    void cancel_after(int seq, bool wait = true) {  // in launch order, so that every closer finds its predecessor's done word
        for (int pass = 0; pass < (wait ? 2 : 1); ++pass)
            for (int q = seq + 1; q <= seq + kSlots; ++q) {
                Slot& sl = slot_of(q);
                if (!sl.live || sl.seq <= seq) continue;
                if (pass == 0) __atomic_store_n(&sl.ctl.cancel, (uint64_t)sl.pass, __ATOMIC_RELEASE);
                else {
                    if (sl.closer.joinable()) sl.closer.join();
                    sl.waves.clear();
                    sl.live = false;
                }
            }
    }
    void read_blocks(int seq, std::vector<Block>& blocks) {
        Slot& sl = slot_of(seq);
        for (int32_t k = 1; k <= nblk; ++k) {
            Block& bl = blocks[(size_t)k];
            const BRec& s = sl.brec[(size_t)k];
            const int32_t js = tw_val(s.js), je = tw_val(s.je);
            bl.i_range = IRange{(k - 1) * kBlockW, k * kBlockW < n ? k * kBlockW : n};
            bl.original_j_range = JRange{tw_val(s.ojs), tw_val(s.oje)};
            bl.j_range = JRange{js, je};
            bl.fixed_j_range = JRange{tw_val(s.fs), tw_val(s.fe)};
            bl.offset = js;
            bl.top_val = tw_val(s.top_val);
            bl.bot_val = tw_val(s.bot_val);
            bl.j_h.reset();
            bl.v.resize((size_t)(je - js) / 64);
            Ctx c;
            c.win = sl.geo.win;
            c.store_cols = 1;
            c.col_ring = sl.geo.col_ring;
            c.n = n;
            const uint64_t* colk = sl.col.data() + ((int64_t)k * sl.geo.col_stride - col_base_word(c, k)) * 2;
            for (size_t w = 0; w < bl.v.size(); ++w) {
                bl.v[w].p = colk[2 * ((size_t)js / 64 + w)];
                bl.v[w].m = colk[2 * ((size_t)js / 64 + w) + 1];
            }
        }
    }
};

}  // namespace

// Same contract as pa_cpu_align (engine_cpu.cpp); info[0] = 1 if the sweep ran, 0 if the parameters are not supported,
// negative = the sweep asked for the fallback (abort reason), info[1..] = last pass's status words.
extern "C" int pa_sweep_emu_align(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, const pa_astarpa2_params* params,
                                  int trace, int nwaves, int32_t* cost_out, char** cigar_out, pa_astarpa2_stats* stats_out,
                                  int32_t* info) {
    if (!params || !params_valid(*params)) return -4;
    const AstarPa2Params p = params_from_c(*params);
    if (info) info[0] = 0;
    if (!sweep_supported(p, a_len, b_len)) return 1;
    CpuBackend be(a, a_len, b, b_len);
    if (!be.ok) return -1;
    HostLauncher dev;
    dev.a = a;
    dev.b = b;
    dev.nwaves = nwaves > 0 ? nwaves : 4;
    dev.heur_kind = p.heuristic == HeuristicKind::Gap ? kHeurGap : p.heuristic == HeuristicKind::SH ? kHeurSH : kHeurNone;
    AlignResult r;
    try {
        SweepAligner<CpuBackend, HostLauncher> al(p, be, dev, trace != 0);
        r = al.align();
    } catch (const SweepFallback& e) {
        if (info) {
            info[0] = -1000 - e.reason;
            info[1] = (int32_t)dev.last_status.state;
            info[2] = dev.last_status.value;
            info[3] = dev.last_status.k_end;
        }
        return 2;
    } catch (const EnginePanic& e) {
        std::fprintf(stderr, "sweep emu: engine panic: %s\n", e.what());
        return -5;
    }
    if (info) info[0] = 1;
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

// Test hooks for sweep_logic.hpp: the fast-forwarded j_range end against the literal loops of engine.hpp.
extern "C" int32_t pa_sweep_jr_end(int32_t kind, int32_t n, int32_t m, const int32_t* sh_h, int32_t is, int32_t ie, int32_t fixed_end,
                                   int32_t gu, int32_t f_max, int32_t sparse_h) {
    HeurParams hp{kind, n, m, sh_h};
    return jr_end_astar(hp, is, ie, fixed_end, gu, f_max, sparse_h);
}
