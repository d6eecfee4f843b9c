// engine_hip.hip -- the A*PA2 block engine (engine.hpp) driven by the HIP strip kernels.
//
// HipBackend keeps the pair's profile (packed codes of a, BitProfile words of b) and the persistent
// horizontal-delta row (one byte per column, blocks.rs:103-105) resident on the GPU; every
// compute / fill rectangle of the engine is one chained-strip launch of strip_kernel.  Block right-edge
// columns (`Block::v`) live in host memory because the band logic reads them (Block::index).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <thread>
#include <vector>

#include "combine_logic.hpp"
#include "engine_capi.hpp"
#include "pa_hip_internal.hpp"
#include "sweep_host.hpp"
#include "sweep_kernel.hpp"

namespace pa {

using engine::BlockParams;
using engine::Cost;
using engine::HMode;
using engine::I;
using engine::V;

struct HipBackend {
    std::vector<uint8_t> a_, b_;
    DeviceBuf d_a, d_b, d_codes, d_prof, d_h, d_htmp, d_gran, d_call, d_misc, d_values;
    size_t w_total = 0;
    hipStream_t s = nullptr;
    bool ok = false;
    int err = 0;
    bool has_h = false;
    // pinned staging
    void* h_stage = nullptr;
    size_t h_stage_size = 0;
    // host-mapped mailbox of the per-block fast path: [done, err, sum, pad.. | v words]
    uint8_t* mbox = nullptr;      // host address
    uint8_t* mbox_dev = nullptr;  // the same memory as the GPU sees it
    size_t mbox_size = 0;
    uint32_t seq = 0;
    DeviceBuf d_counter;
    size_t gran_zeroed = 0;  // granules of d_gran known to be zero (the strips hand every granule back zeroed)

    int device = -1;  // the device the pooled buffers live on

    HipBackend() = default;
    HipBackend(const uint8_t* a, size_t n, const uint8_t* b, size_t m) { bind(a, n, b, m); }

    // (Re)bind the backend to a pair.  Buffers, the stream and the mailbox are kept from call to call (a thread-local pool,
    // see pooled_backend()): a relinked astarpa-c user calls astarpa2_simple in a loop, and six hipMallocs + a stream per
    // call cost more than a short alignment.
    void bind(const uint8_t* a, size_t n, const uint8_t* b, size_t m) {
        ok = false;
        err = 0;
        has_h = false;
        a_.assign(a, a + n);
        b_.assign(b, b + m);
        if (!ensure_device()) { err = PA_E_HIP; return; }
        (void)hipGetDevice(&device);
        w_total = (m + 63) / 64;
        const size_t cw = (n + 15) / 16 + 16;  // the sweep kernel reads up to 8 words past the last column's
        if (!d_a.reserve(n) || !d_b.reserve(m) || !d_codes.reserve(cw * 4) || !d_prof.reserve(w_total * 16 + 16) ||
            !d_misc.reserve(16) || !d_htmp.reserve(n + 64)) { err = PA_E_HIP; return; }
        if (!s && !hip_ok(hipStreamCreate(&s), "hipStreamCreate")) { err = PA_E_HIP; return; }
        // The per-call set-up in four stream operations (round 6; there were eight, two of them copies from pageable memory): the sequences go
        // through the pinned staging buffer, ONE kernel packs a (zero padding included) and builds b's profile, the "character outside ACGT"
        // flag is a word of the host-mapped mailbox.
        uint32_t misc[4] = {0, 0, 0, 0};
        bool good = true;
        try {
            ensure_mailbox(0);
            uint8_t* st = static_cast<uint8_t*>(stage(n + m + 64));
            const size_t off_b = (n + 63) & ~size_t(63);
            if (n) std::memcpy(st, a, n);
            if (m) std::memcpy(st + off_b, b, m);
            volatile uint32_t* mb = reinterpret_cast<volatile uint32_t*>(mbox);
            mb[3] = 0;
            __atomic_thread_fence(__ATOMIC_SEQ_CST);
            good = (n == 0 || hip_ok(hipMemcpyAsync(d_a.ptr, st, n, hipMemcpyHostToDevice, s), "H2D a")) &&
                   (m == 0 || hip_ok(hipMemcpyAsync(d_b.ptr, st + off_b, m, hipMemcpyHostToDevice, s), "H2D b")) &&
                   encode_pair_device(d_a.as<uint8_t>(), (int)n, d_codes.as<uint32_t>(), (int)cw, d_b.as<uint8_t>(), (int)m, d_prof.as<uint64_t>(),
                                      reinterpret_cast<uint32_t*>(mbox_dev) + 3, s) &&
                   hip_ok(hipStreamSynchronize(s), "sync");
            misc[3] = mb[3];
        } catch (const engine::EnginePanic&) {
            good = false;
        }
        if (!good) { err = PA_E_HIP; return; }
        if (misc[3]) {
            set_error("sequence contains a base outside ACGT");
            err = PA_E_INVALID_BASE;
            return;
        }
        ok = true;
    }
    ~HipBackend() {
        if (mbox) (void)hipHostFree(mbox);
        if (h_stage) (void)hipHostFree(h_stage);
        if (s) (void)hipStreamDestroy(s);
    }

    I n() const { return (I)a_.size(); }
    I m() const { return (I)b_.size(); }
    const uint8_t* a() const { return a_.data(); }
    const uint8_t* b() const { return b_.data(); }

    void fail(int code) {
        err = code;
        throw engine::EnginePanic(std::string("HIP backend failure: ") + pa_last_error());
    }

    void enable_h_row() {  // blocks.rs:119-123: vec![(0,0); a.len()]
        if (has_h) return;
        if (!d_h.reserve(a_.size() + 64) || !hip_ok(hipMemsetAsync(d_h.ptr, 0, a_.size() + 64, s), "memset h")) fail(PA_E_HIP);
        has_h = true;
    }

    void* stage(size_t bytes) {
        if (bytes > h_stage_size) {
            if (h_stage) (void)hipHostFree(h_stage);
            h_stage = nullptr;
            h_stage_size = 0;
            const size_t want = std::max<size_t>(bytes * 2, 1 << 16);
            if (!hip_ok(hipHostMalloc(&h_stage, want, hipHostMallocDefault), "hipHostMalloc")) fail(PA_E_HIP);
            h_stage_size = want;
        }
        return h_stage;
    }

    static constexpr size_t kMboxV = 64;  // offset of the v words inside the mailbox

    void ensure_mailbox(size_t words, size_t extra_bytes = 0) {
        const size_t need = kMboxV + words * 16 + extra_bytes;
        if (need <= mbox_size) return;
        if (mbox) (void)hipHostFree(mbox);
        mbox = nullptr;
        const size_t want = std::max<size_t>(need * 2, 1 << 16);
        void* hp = nullptr;
        void* dp = nullptr;
        if (!hip_ok(hipHostMalloc(&hp, want, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc(mailbox)") ||
            !hip_ok(hipHostGetDevicePointer(&dp, hp, 0), "hipHostGetDevicePointer"))
            fail(PA_E_HIP);
        mbox = (uint8_t*)hp;
        mbox_dev = (uint8_t*)dp;
        mbox_size = want;
        std::memset(mbox, 0, kMboxV);
        if (!d_counter.ptr && (!d_counter.alloc(64) || !hip_ok(hipMemsetAsync(d_counter.ptr, 0, 64, s), "memset counter"))) fail(PA_E_HIP);
    }

    void ensure_granules(size_t ngran) {
        if (ngran <= gran_zeroed) return;
        const size_t want = std::max<size_t>(ngran * 2, 512);
        if (!d_gran.alloc(want * 8) || !hip_ok(hipMemsetAsync(d_gran.ptr, 0, want * 8, s), "memset gran")) fail(PA_E_HIP);
        gran_zeroed = want;
    }

    // Fast path of one cost-only rectangle: the strips are described by kernel arguments, `v` / sum / err / done live in
    // the host-mapped mailbox, the host spins on `done`.  One API call (the launch) per block.
    // values_host / hbot_host (round 6): the traceback's re-fill through the same mailbox -- every column's V and the bottom row's deltas are
    // written by the kernel straight into host-mapped memory behind the v words (no copy command, no stream synchronisation: 81 -> ~40 us per
    // re-filled block of the loop over the drop-in symbol).
    Cost launch_rect_fast(I i0, I i1, size_t w0, size_t w1, V* v, const uint8_t* hin, uint8_t* hout, bool exact, V* values_host = nullptr,
                          int8_t* hbot_host = nullptr) {
        const int n = i1 - i0;
        const size_t w = w1 - w0;
        const size_t S = (w + kWordsPerStrip - 1) / kWordsPerStrip;
        const size_t G = (size_t)(n + 31) / 32;
        const bool fill = values_host != nullptr;
        const size_t off_values = kMboxV + ((w * 16 + 63) & ~size_t(63)), values_bytes = fill ? (size_t)n * w * 16 : 0;
        const size_t off_hbot = off_values + ((values_bytes + 63) & ~size_t(63));
        ensure_mailbox(w, fill ? (off_hbot - kMboxV - w * 16) + (size_t)n + 64 : 0);
        ensure_granules(S > 1 ? (S - 1) * G : 0);
        volatile uint32_t* mb = reinterpret_cast<volatile uint32_t*>(mbox);
        std::memcpy(mbox + kMboxV, v, w * 16);
        mb[1] = 0;  // err
        mb[2] = 0;  // sum
        ++seq;
        RectArgs r;
        r.a_codes = d_codes.as<uint32_t>();
        r.b_prof = d_prof.as<uint32_t>();
        r.v = reinterpret_cast<uint32_t*>(mbox_dev + kMboxV) - w0 * 4;
        r.hin_arr = hin;
        r.hout_arr = fill ? mbox_dev + off_hbot - i0 : hout;  // (indexed by absolute column)
        r.gran = d_gran.as<uint64_t>();
        r.gran_stride = G;
        r.sum_out = reinterpret_cast<int32_t*>(mbox_dev) + 2;
        r.err = reinterpret_cast<uint32_t*>(mbox_dev) + 1;
        r.done = reinterpret_cast<uint32_t*>(mbox_dev);
        r.counter = d_counter.as<uint32_t>();
        r.n = n;
        r.col0 = i0;
        r.w0 = (int)w0;
        r.w1 = (int)w1;
        r.exact_end = exact ? 1 : 0;
        r.seq = seq;
        r.values = fill ? reinterpret_cast<uint32_t*>(mbox_dev + off_values) : nullptr;
        r.fill_stride = (int)w;
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        if (fill) hipLaunchKernelGGL((rect_kernel<1, true>), dim3((unsigned)S), dim3(64), 0, s, r);
        else hipLaunchKernelGGL((rect_kernel<1>), dim3((unsigned)S), dim3(64), 0, s, r);
        if (!hip_ok(hipGetLastError(), "rect_kernel launch")) fail(PA_E_HIP);
        // spin on the completion word; the kernel's own spins are bounded, so this ends
        uint64_t spins = 0;
        while (__atomic_load_n(reinterpret_cast<uint32_t*>(mbox), __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 0xFFFFF) == 0 && hipStreamQuery(s) != hipErrorNotReady) {
                // the stream drained (or failed) without the flag: take the slow, certain route
                if (!hip_ok(hipStreamSynchronize(s), "sync")) fail(PA_E_HIP);
                if (__atomic_load_n(reinterpret_cast<uint32_t*>(mbox), __ATOMIC_ACQUIRE) != seq) {
                    set_error("rect_kernel finished without signalling completion");
                    fail(PA_E_INTERNAL);
                }
            }
        }
        if (mb[1] != PA_ERR_NONE) {
            set_error("device spin timeout (err=%u)", (unsigned)mb[1]);
            gran_zeroed = 0;  // the hand-off buffer may be dirty
            fail(PA_E_TIMEOUT);
        }
        std::memcpy(v, mbox + kMboxV, w * 16);
        if (fill) {
            std::memcpy(values_host, mbox + off_values, values_bytes);
            std::memcpy(hbot_host, mbox + off_hbot, (size_t)n);
        }
        return (Cost)(int32_t)mb[2];
    }

    // The ranges of one block of the incremental doubling (blocks.rs:370-469) in ONE launch.  Equivalent to calling
    // compute() for every segment in order; the bottom-row sum of the last segment is returned.
    struct ChainSeg {
        size_t w0, w1;
        V* v;
        HMode mode;
    };
    Cost compute_chain(I i0, I i1, const ChainSeg* segs, int nseg, const BlockParams& bp) {
        const I n = i1 - i0;
        static const bool no_fast = getenv("PA_ENGINE_NO_FAST_PATH") != nullptr || getenv("PA_ENGINE_NO_CHAIN") != nullptr;
        bool fuse = !no_fast && n > 0 && nseg >= 2 && nseg <= 3 && has_h;
        size_t strips = 0, lo = SIZE_MAX, hi = 0;
        for (int k = 0; k < nseg && fuse; ++k) {
            if (segs[k].w0 >= segs[k].w1) fuse = false;  // empty ranges have side effects of their own (see compute())
            if (k > 0 && segs[k].w0 < segs[k - 1].w1) fuse = false;
            strips += (segs[k].w1 - segs[k].w0 + kWordsPerStrip - 1) / kWordsPerStrip;
            lo = std::min(lo, segs[k].w0);
            hi = std::max(hi, segs[k].w1);
        }
        if (fuse && strips > 1024) fuse = false;
        if (!fuse) {
            Cost last = 0;
            for (int k = 0; k < nseg; ++k) last = compute(i0, i1, segs[k].w0, segs[k].w1, segs[k].v, segs[k].mode, bp);
            return last;
        }
        const size_t G = (size_t)(n + 31) / 32;
        ensure_mailbox(hi - lo);
        ensure_granules(strips * G);
        volatile uint32_t* mb = reinterpret_cast<volatile uint32_t*>(mbox);
        for (int k = 0; k < nseg; ++k) std::memcpy(mbox + kMboxV + (segs[k].w0 - lo) * 16, segs[k].v, (segs[k].w1 - segs[k].w0) * 16);
        mb[1] = 0;
        mb[2] = 0;
        ++seq;
        ChainArgs r;
        r.a_codes = d_codes.as<uint32_t>();
        r.b_prof = d_prof.as<uint32_t>();
        r.v = reinterpret_cast<uint32_t*>(mbox_dev + kMboxV) - lo * 4;
        r.h_arr = d_h.as<uint8_t>();
        r.gran = d_gran.as<uint64_t>();
        r.gran_stride = G;
        r.sum_out = reinterpret_cast<int32_t*>(mbox_dev) + 2;
        r.err = reinterpret_cast<uint32_t*>(mbox_dev) + 1;
        r.done = reinterpret_cast<uint32_t*>(mbox_dev);
        r.counter = d_counter.as<uint32_t>();
        r.n = n;
        r.col0 = i0;
        r.nseg = nseg;
        r.seq = seq;
        bool prev_stores = false;
        for (int k = 0; k < 3; ++k) {
            r.w0[k] = r.w1[k] = r.top[k] = r.store[k] = 0;
            if (k >= nseg) continue;
            r.w0[k] = (int32_t)segs[k].w0;
            r.w1[k] = (int32_t)segs[k].w1;
            const HMode m = segs[k].mode;
            r.store[k] = (m == HMode::Update || m == HMode::Output) ? 1 : 0;
            if (m == HMode::None || m == HMode::Output) r.top[k] = kTopOne;
            else if (prev_stores && k > 0 && segs[k - 1].w1 == segs[k].w0) r.top[k] = kTopChain;  // the row the segment above stores
            else r.top[k] = kTopStored;
            prev_stores = r.store[k] != 0;
        }
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        hipLaunchKernelGGL((rect_chain_kernel<1>), dim3((unsigned)strips), dim3(64), 0, s, r);
        if (!hip_ok(hipGetLastError(), "rect_chain_kernel launch")) fail(PA_E_HIP);
        uint64_t spins = 0;
        while (__atomic_load_n(reinterpret_cast<uint32_t*>(mbox), __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 0xFFFFF) == 0 && hipStreamQuery(s) != hipErrorNotReady) {
                if (!hip_ok(hipStreamSynchronize(s), "sync")) fail(PA_E_HIP);
                if (__atomic_load_n(reinterpret_cast<uint32_t*>(mbox), __ATOMIC_ACQUIRE) != seq) {
                    set_error("rect_chain_kernel finished without signalling completion");
                    fail(PA_E_INTERNAL);
                }
            }
        }
        if (mb[1] != PA_ERR_NONE) {
            set_error("device spin timeout (err=%u)", (unsigned)mb[1]);
            gran_zeroed = 0;
            fail(PA_E_TIMEOUT);
        }
        for (int k = 0; k < nseg; ++k) std::memcpy(segs[k].v, mbox + kMboxV + (segs[k].w0 - lo) * 16, (segs[k].w1 - segs[k].w0) * 16);
        return (Cost)(int32_t)mb[2];
    }

    // One rectangle launch.  hin/hout are device byte rows indexed by absolute column (or nullptr).
    // Per call: ONE H2D of a pinned staging image [ticket,err,sum,pad | v words | jobs] into `d_call`, an optional
    // granule clear (only when the rectangle spans several strips), the launch, ONE D2H of [misc | v], one sync.
    Cost launch_rect(I i0, I i1, size_t w0, size_t w1, V* v, const uint8_t* hin, uint8_t* hout, bool exact,
                     V* values_host, int8_t* hbot_host) {
        const int n = i1 - i0;
        const size_t w = w1 - w0;
        const bool fill = values_host != nullptr;
        static const bool no_fast = getenv("PA_ENGINE_NO_FAST_PATH") != nullptr;
        if (!fill && !no_fast && (w + kWordsPerStrip - 1) / kWordsPerStrip <= 1024) return launch_rect_fast(i0, i1, w0, w1, v, hin, hout, exact);
        static const bool no_fast_fill = no_fast || getenv("PA_ENGINE_NO_FAST_FILL") != nullptr;
        if (fill && !no_fast_fill && hin == nullptr && (size_t)n * w * 16 <= (size_t(1) << 20) && (w + kWordsPerStrip - 1) / kWordsPerStrip <= 64)
            return launch_rect_fast(i0, i1, w0, w1, v, nullptr, nullptr, exact, values_host, hbot_host);
        const size_t ngran = rect_granules(n, (int)w);
        const size_t G = (size_t)(n + 31) / 32;
        ensure_granules(ngran);
        if (fill && d_values.size < (size_t)n * w * 16 && !d_values.alloc((size_t)n * w * 16 * 2)) fail(PA_E_HIP);
        const size_t S = (w + kWordsPerStrip - 1) / kWordsPerStrip;
        const size_t off_v = 64, off_jobs = off_v + ((w * 16 + 63) & ~size_t(63));
        const size_t total = off_jobs + S * sizeof(StripJob);
        if (d_call.size < total && !d_call.alloc(total * 2)) fail(PA_E_HIP);
        uint8_t* dev = d_call.as<uint8_t>();

        std::vector<StripJob> jobs;
        RectPlan r;
        r.a_codes = d_codes.as<uint32_t>();
        r.col0 = i0;
        r.b_prof = d_prof.as<uint32_t>();
        // the strip indexes v by absolute word: bias the pointer so that word w0 lands at dev + off_v
        r.v = reinterpret_cast<uint32_t*>(dev + off_v) - w0 * 4;
        r.n = n;
        r.w0 = (int)w0;
        r.w1 = (int)w1;
        r.hin_arr = hin;
        r.hout_arr = hout;
        r.gran = d_gran.as<uint64_t>();
        r.gran_stride = G;
        r.sum_out = reinterpret_cast<int32_t*>(dev) + 2;
        r.exact_end = exact;
        r.values = fill ? d_values.as<uint32_t>() : nullptr;
        r.fill_stride = (int)w;
        r.fill_word0 = 0;
        plan_rect(jobs, r);
        if (fill)
            for (auto& j : jobs) j.fill_word0 = j.word0 - (int)w0;

        uint8_t* st = (uint8_t*)stage(total);
        std::memset(st, 0, off_v);
        std::memcpy(st + off_v, v, w * 16);
        std::memcpy(st + off_jobs, jobs.data(), jobs.size() * sizeof(StripJob));
        bool good = hip_ok(hipMemcpyAsync(dev, st, total, hipMemcpyHostToDevice, s), "H2D call image") &&
                    launch_strips(reinterpret_cast<const StripJob*>(dev + off_jobs), (int)jobs.size(), fill,
                                  reinterpret_cast<uint32_t*>(dev), s, /*zero_ticket=*/false) &&
                    hip_ok(hipMemcpyAsync(st, dev, off_v + w * 16, hipMemcpyDeviceToHost, s), "D2H misc+v");
        if (good && fill) {
            good = hip_ok(hipMemcpyAsync(values_host, d_values.ptr, (size_t)n * w * 16, hipMemcpyDeviceToHost, s), "D2H values") &&
                   hip_ok(hipMemcpyAsync(hbot_host, hout + i0, (size_t)n, hipMemcpyDeviceToHost, s), "D2H hbot");
        }
        good = good && hip_ok(hipStreamSynchronize(s), "sync");
        if (!good) fail(PA_E_HIP);
        const uint32_t* misc = reinterpret_cast<const uint32_t*>(st);
        if (misc[1] != PA_ERR_NONE) {
            set_error("device spin timeout (err=%u)", misc[1]);
            fail(PA_E_TIMEOUT);
        }
        std::memcpy(v, st + off_v, w * 16);
        return (Cost)(int32_t)misc[2];
    }

    Cost sum_h_row(I i0, I i1) {  // empty word range: bottom row == top row
        std::vector<uint8_t> h((size_t)(i1 - i0));
        if (!hip_ok(hipMemcpyAsync(h.data(), d_h.as<uint8_t>() + i0, h.size(), hipMemcpyDeviceToHost, s), "D2H h") ||
            !hip_ok(hipStreamSynchronize(s), "sync"))
            fail(PA_E_HIP);
        Cost c = 0;
        for (uint8_t x : h) c += (Cost)(x & 1) - (Cost)((x >> 1) & 1);
        return c;
    }

    // blocks.rs:686-748 (the `simd` / `no_ilp` switches select CPU schedules in the reference; results are
    // schedule independent, the GPU always runs its strip schedule).
    Cost compute(I i0, I i1, size_t w0, size_t w1, V* v, HMode mode, const BlockParams&) {
        const I n = i1 - i0;
        if (n <= 0) return 0;
        if (w0 >= w1) {
            switch (mode) {
                case HMode::None: return n;
                case HMode::Output:
                    if (!hip_ok(hipMemsetAsync(d_h.as<uint8_t>() + i0, 1, (size_t)n, s), "memset h")) fail(PA_E_HIP);
                    return n;
                default: return sum_h_row(i0, i1);
            }
        }
        switch (mode) {
            case HMode::None: return launch_rect(i0, i1, w0, w1, v, nullptr, nullptr, false, nullptr, nullptr);
            case HMode::Input: return launch_rect(i0, i1, w0, w1, v, d_h.as<uint8_t>(), nullptr, false, nullptr, nullptr);
            case HMode::Update: return launch_rect(i0, i1, w0, w1, v, d_h.as<uint8_t>(), d_h.as<uint8_t>(), true, nullptr, nullptr);
            case HMode::Output: return launch_rect(i0, i1, w0, w1, v, nullptr, d_h.as<uint8_t>(), true, nullptr, nullptr);
        }
        return 0;
    }

    // blocks.rs:627-648
    void fill(I i0, I i1, size_t w0, size_t w1, V* v, V* values, int8_t* hbot, const BlockParams&) {
        const I n = i1 - i0;
        if (n <= 0) return;
        if (w0 >= w1) {
            for (I i = 0; i < n; ++i) hbot[i] = 1;
            return;
        }
        std::vector<int8_t> raw((size_t)n);
        launch_rect(i0, i1, w0, w1, v, nullptr, d_htmp.as<uint8_t>(), true, values, raw.data());
        for (I i = 0; i < n; ++i) hbot[i] = (int8_t)((raw[i] & 1) - ((raw[i] >> 1) & 1));
    }

    std::vector<int8_t> debug_read_h(I i0, I i1) {
        std::vector<uint8_t> h((size_t)(i1 - i0));
        if (!hip_ok(hipMemcpyAsync(h.data(), d_h.as<uint8_t>() + i0, h.size(), hipMemcpyDeviceToHost, s), "D2H h") ||
            !hip_ok(hipStreamSynchronize(s), "sync"))
            fail(PA_E_HIP);
        std::vector<int8_t> r;
        for (uint8_t x : h) r.push_back((int8_t)((x & 1) - ((x >> 1) & 1)));
        return r;
    }
    void debug_write_h(I i0, I i1, const std::vector<int8_t>& x) {
        std::vector<uint8_t> h((size_t)(i1 - i0));
        for (size_t k = 0; k < h.size(); ++k) h[k] = (uint8_t)((x[k] > 0 ? 1 : 0) | (x[k] < 0 ? 2 : 0));
        if (!hip_ok(hipMemcpyAsync(d_h.as<uint8_t>() + i0, h.data(), h.size(), hipMemcpyHostToDevice, s), "H2D h") ||
            !hip_ok(hipStreamSynchronize(s), "sync"))
            fail(PA_E_HIP);
    }
};

// One backend per host thread and device, reused from call to call.
static std::unique_ptr<HipBackend>& pooled_backend_slot() {
    static thread_local std::unique_ptr<HipBackend> tl;
    return tl;
}
static HipBackend& pooled_backend() {
    std::unique_ptr<HipBackend>& tl = pooled_backend_slot();
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!tl || (tl->device >= 0 && tl->device != dev)) tl = std::make_unique<HipBackend>();
    return *tl;
}

// ---- the device-side sweep: one launch per align_for_bounded_dist pass (sweep_wave.hpp) ---------------------------------
// Launcher of sweep::SweepAligner over HIP.  Its buffers are pooled per host thread like the backend's.
// Passes of one band search run pipelined (sweep_host.hpp search()): every pass in flight has a SLOT with its own records,
// buffers and stream; the merged block records of the completed passes alternate between two arrays.
struct SweepSlot {
    DeviceBuf d_brec, d_trec, d_misc, d_start, d_pring, d_gran, d_col;
    DeviceBuf d_merged;  // the block records of every earlier pass with this pass's on top (what the next pass reads once this one is over)
    hipStream_t s = nullptr;
    hipEvent_t merged_ev = nullptr;  // recorded behind the merge of the pass that ran here
    sweep::Status* h_status = nullptr;  // pinned
    // the pass that runs (or ran last) here
    int seq = 0;
    uint32_t pass = 0;
    bool live = false;
    int32_t f_max = 0, waves = 0;
    sweep::PassGeometry geo{};
    double t_launch = 0;
    // bprog @0, ticket @16, done @24, merge counter @32, cancel @56 (directly before the status block), status @64, phase clocks @512
    uint64_t* bprog() { return d_misc.as<uint64_t>(); }
    uint32_t* ticket() { return d_misc.as<uint32_t>() + 4; }
    uint64_t* done() { return d_misc.as<uint64_t>() + 3; }
    uint32_t* merge_count() { return d_misc.as<uint32_t>() + 8; }
    uint64_t* cancel() { return d_misc.as<uint64_t>() + 7; }
    sweep::Status* status() { return reinterpret_cast<sweep::Status*>(d_misc.as<uint8_t>() + 64); }
};
struct SweepPool {
    // a pass's records must outlive its successor, and a pass that is taken again (see sweep_host.hpp) follows a pass that was
    // completed five launches earlier: two slots more than passes in flight
    static constexpr int kSlots = 7;
    static constexpr int kMaxInFlight = 5;
    SweepSlot slots[kSlots];
    DeviceBuf d_merged0, d_sh, d_recs, d_offs, d_pack;  // d_merged0: "no block exists yet" (what the first pass of a pair reads)
    // hipFree waits for the whole device -- with passes in flight that serialises them (C5 cold: 2.7 s instead of 1.x).  A slot
    // buffer that has to grow while other passes run is therefore replaced, and the old allocation freed when nothing is in flight.
    std::vector<void*> graveyard;
    void bury(DeviceBuf& b) {
        if (b.ptr) graveyard.push_back(b.ptr);
        b.ptr = nullptr;
        b.size = 0;
    }
    void free_graveyard() {
        for (void* q : graveyard) (void)hipFree(q);
        graveyard.clear();
    }
    hipStream_t ctl = nullptr;  // cancel words go out here, past the running passes
    void* h_pin = nullptr;
    size_t h_pin_size = 0;
    uint32_t pass_id = 0;
    int device = -1;
    bool ok = false;
    SweepPool() {
        ok = hip_ok(hipStreamCreateWithFlags(&ctl, hipStreamNonBlocking), "hipStreamCreate");
        for (SweepSlot& sl : slots)
            ok = ok && hip_ok(hipStreamCreateWithFlags(&sl.s, hipStreamNonBlocking), "hipStreamCreate") &&
                 hip_ok(hipEventCreateWithFlags(&sl.merged_ev, hipEventDisableTiming), "hipEventCreate") &&
                 hip_ok(hipHostMalloc((void**)&sl.h_status, sizeof(sweep::Status), hipHostMallocDefault), "hipHostMalloc");
    }
    ~SweepPool() {
        free_graveyard();
        if (h_pin) (void)hipHostFree(h_pin);
        for (SweepSlot& sl : slots) {
            if (sl.h_status) (void)hipHostFree(sl.h_status);
            if (sl.merged_ev) (void)hipEventDestroy(sl.merged_ev);
            if (sl.s) (void)hipStreamDestroy(sl.s);
        }
        if (ctl) (void)hipStreamDestroy(ctl);
    }
    void* pinned(size_t bytes) {
        if (bytes > h_pin_size) {
            if (h_pin) (void)hipHostFree(h_pin);
            h_pin = nullptr;
            h_pin_size = 0;
            const size_t want = std::max<size_t>(bytes * 2, 1 << 16);
            if (!hip_ok(hipHostMalloc(&h_pin, want, hipHostMallocDefault), "hipHostMalloc")) return nullptr;
            h_pin_size = want;
        }
        return h_pin;
    }
};
static std::unique_ptr<SweepPool>& sweep_pool_slot() {
    static thread_local std::unique_ptr<SweepPool> tl;
    return tl;
}
static SweepPool& sweep_pool() {
    std::unique_ptr<SweepPool>& tl = sweep_pool_slot();
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!tl || tl->device != dev) {
        tl = std::make_unique<SweepPool>();
        tl->device = dev;
    }
    return *tl;
}

struct HipSweepLauncher {
    HipBackend& be;
    SweepPool& pool;
    int32_t n = 0, m = 0, nblk = 0;
    bool trace = false;
    bool has_sh = false;
    int32_t heur_kind = sweep::kHeurGap;

    HipSweepLauncher(HipBackend& backend, SweepPool& p) : be(backend), pool(p) { active_callers().fetch_add(1, std::memory_order_relaxed); }
    ~HipSweepLauncher() {
        cancel_after(0);
        active_callers().fetch_sub(1, std::memory_order_relaxed);
    }
    HipSweepLauncher(const HipSweepLauncher&) = delete;
    HipSweepLauncher& operator=(const HipSweepLauncher&) = delete;

    void hip_fail(const char* what) { throw sweep::SweepFallback(what, -2); }
    static bool timing_on() {
        static const bool on = std::getenv("PA_SWEEP_TIMING") != nullptr;
        return on;
    }
    SweepSlot& slot_of(int seq) { return pool.slots[seq % SweepPool::kSlots]; }
    // (a pass's records must outlive its successor, which reads them: one slot more than passes in flight)
    // Passes in flight run on separate streams, and streams only run side by side on separate hardware queues: the ROCm runtime
    // multiplexes all streams of a process over GPU_MAX_HW_QUEUES of them (default 4; pa_runtime_hints() below asks for 16
    // when the library is loaded before the runtime starts).  A pass queued BEHIND a later one would only cost time, never
    // correctness: passes are submitted in order and wait for their predecessors only.
    // Every pass in flight is a RUNNING kernel (it polls its predecessor) on a stream of its own, and the GPU serves only so many
    // queues side by side: with five passes each, four host threads got 716 pairs/s out of the drop-in loop (10 kbp pairs), with
    // two each 1042; eight threads 698 -> 1489 (profiles/r02_runs/dropin_threads.log).  Short pairs rarely need more than three
    // tries, passes beyond that are launches and cancellations for nothing.
    int max_in_flight() const {
        static const int forced = [] {
            const char* e = std::getenv("PA_SWEEP_DEPTH");
            return e ? std::min(std::max(std::atoi(e), 1), SweepPool::kMaxInFlight) : 0;
        }();
        if (forced) return forced;
        static const int queues = [] {
            const char* q = std::getenv("GPU_MAX_HW_QUEUES");
            return q ? std::max(std::atoi(q), 1) : 4;
        }();
        const int base = queues >= 8 ? (nblk > kShortPairBlocks ? SweepPool::kMaxInFlight : 3) : 3;
        const int callers = recent_callers();
        return callers == 1 ? base : (callers == 2 ? std::min(base, 3) : 2);
    }
    // The callers inside an alignment now, or the most seen during the last 20 ms: a thread between two calls of a loop still counts.
    static int recent_callers() {
        static std::atomic<int> peak{0};
        static std::atomic<int64_t> peak_ns{0};
        const int now_callers = std::max(active_callers().load(std::memory_order_relaxed), 1);
        const int64_t now = (int64_t)(engine::now_s() * 1e9);
        if (now_callers >= peak.load(std::memory_order_relaxed) || now - peak_ns.load(std::memory_order_relaxed) > 20'000'000) {
            peak.store(now_callers, std::memory_order_relaxed);  // (racing updates can only misjudge the depth for a moment)
            peak_ns.store(now, std::memory_order_relaxed);
        }
        return std::max(now_callers, peak.load(std::memory_order_relaxed));
    }
    static constexpr int32_t kShortPairBlocks = 128;  // 32 kbp
    static std::atomic<int>& active_callers() {  // host threads inside a sweep alignment right now
        static std::atomic<int> n{0};
        return n;
    }
    // wavefronts: one per strip the band can cover at a time (+ slack), one workgroup each
    int pass_waves(int32_t f_max) const {
        const sweep::PassGeometry g = sweep::pass_geometry(n, m, f_max);
        int64_t waves = (2ll * g.win) / sweep::kStripRows + 6;
        if (waves > g.nstrips) waves = g.nstrips;
        if (waves > 1024) waves = 1024;
        return (int)waves;
    }
    // Wavefronts of all passes in flight.  Most of a pass's wavefronts idle (a pass reserves one per strip its window can hold,
    // the band covers a fraction of them at a time), so somewhat more than one per SIMD is fine; far more would only slow
    // the passes that matter.
    int wave_budget() const {
        static const int budget = std::getenv("PA_SWEEP_WAVE_BUDGET") ? std::atoi(std::getenv("PA_SWEEP_WAVE_BUDGET")) : 1600;  // (C5: 1024 -> 1.9 s, 1600 -> 1.5 s, 4096 -> 4.9 s)
        return budget;
    }

    // bytes of a zero-initialised tagged buffer: cleared only when it is new (tags of older passes never match)
    void reserve_tagged(DeviceBuf& b, size_t bytes, hipStream_t st) {
        if (reserve_no_sync(b, bytes) && !hip_ok(hipMemsetAsync(b.ptr, 0, b.size, st), "memset")) hip_fail("memset");
    }
    // grow-only, never a hipFree (see SweepPool::graveyard); true when a new, uninitialised buffer was allocated
    bool reserve_no_sync(DeviceBuf& b, size_t bytes) {
        if (b.ptr && b.size >= bytes) return false;
        pool.bury(b);
        if (!b.alloc(bytes + bytes / 4 + 256)) hip_fail("hipMalloc");
        return true;
    }

    void begin_pair(int32_t n_, int32_t m_, int32_t nblk_, const int32_t* sh, bool tr) {
        using namespace sweep;
        if (!pool.ok) hip_fail("sweep pool");
        pool.free_graveyard();  // (nothing is in flight between pairs)
        n = n_;
        m = m_;
        nblk = nblk_;
        trace = tr;
        has_sh = sh != nullptr;
        const size_t recs = (size_t)nblk + 2;
        if (pool.pass_id >= 3000) {  // tags wrap at 4095: start over with clean tagged buffers (nothing is in flight between pairs)
            for (SweepSlot& sl : pool.slots)
                for (DeviceBuf* b : {&sl.d_brec, &sl.d_trec, &sl.d_start, &sl.d_pring, &sl.d_misc})
                    if (b->ptr && !hip_ok(hipMemsetAsync(b->ptr, 0, b->size, be.s), "memset")) hip_fail("memset");
            pool.pass_id = 0;
        }
        if (!pool.d_merged0.reserve(recs * sizeof(BlockRec)) ||
            !hip_ok(hipMemsetD32Async((hipDeviceptr_t)pool.d_merged0.ptr, (int)kNone, recs * sizeof(BlockRec) / 4, be.s), "memset merged"))
            hip_fail("merged records");
        for (SweepSlot& sl : pool.slots) {
            if (!sl.d_merged.reserve(recs * sizeof(BlockRec))) hip_fail("merged records");
            reserve_tagged(sl.d_brec, recs * sizeof(BRec), be.s);
            reserve_tagged(sl.d_trec, recs * sizeof(TRec), be.s);
            reserve_tagged(sl.d_misc, 1024, be.s);
            sl.live = false;
            sl.seq = 0;
        }
        if (has_sh) {
            if (!pool.d_sh.reserve(((size_t)n + 1) * 4) ||
                !hip_ok(hipMemcpyAsync(pool.d_sh.ptr, sh, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, be.s), "H2D sh"))
                hip_fail("sh table");
        }
        // the passes run on the slots' streams: everything set up on the pair's stream (profiles, codes, the above) is done first
        if (!hip_ok(hipStreamSynchronize(be.s), "sync")) hip_fail("begin_pair");
    }

    sweep::BlockRec read_merged(int seq, int32_t k) {  // after wait_pass(seq)
        sweep::BlockRec r;
        const DeviceBuf& mb = seq == 0 ? pool.d_merged0 : slot_of(seq).d_merged;
        if (seq != 0 && slot_of(seq).seq != seq) hip_fail("merged records of a pass whose slot was reused");
        if (!hip_ok(hipMemcpyAsync(&r, mb.as<sweep::BlockRec>() + k, sizeof(r), hipMemcpyDeviceToHost, be.s), "D2H rec") ||
            !hip_ok(hipStreamSynchronize(be.s), "sync"))
            hip_fail("read_merged");
        return r;
    }

    void launch_pass(int seq, int prev_seq, int32_t f_max, int32_t sparse_h, const sweep::PassInit& init) {
        using namespace sweep;
        SweepSlot& sl = slot_of(seq);
        if (sl.live) hip_fail("sweep slot busy");
        SweepSlot* pv = prev_seq ? &slot_of(prev_seq) : nullptr;
        if (pv && (pv == &sl || pv->seq != prev_seq)) hip_fail("sweep slot of the previous pass was reused");
        const bool pv_running = pv && pv->live;  // still in flight: read its records as they appear; else only its merged array
        const BlockRec* merged_in = pv ? pv->d_merged.as<BlockRec>() : pool.d_merged0.as<BlockRec>();
        // tags carry 12 bits of pass id (sweep_logic.hpp blk_tag) and the tagged buffers are cleared between pairs only: a pair that
        // needs more passes than that (LinearSearch with a small delta) goes to the host-driven engine instead of aliasing tags
        if (pool.pass_id >= 4000) throw SweepFallback("pass ids exhausted within one pair", -4);
        pool.pass_id += 1;
        sl.seq = seq;
        sl.pass = pool.pass_id;
        sl.f_max = f_max;
        sl.geo = pass_geometry(n, m, f_max);
        const PassGeometry& geo = sl.geo;
        const size_t nslots = trace ? (size_t)nblk + 1 : (size_t)geo.col_ring;
        const size_t gran_bytes = (size_t)geo.nstrips * (size_t)geo.gran_stride * 8;
        const size_t col_bytes = nslots * (size_t)geo.col_stride * 16;
        const size_t pr_bytes = (size_t)geo.nstrips * (size_t)geo.pr_stride * 8;
        if (gran_bytes + col_bytes + pr_bytes > (size_t)40 << 30) throw SweepFallback("sweep buffers too large", -3);
        reserve_tagged(sl.d_start, (size_t)geo.nstrips * 8, sl.s);
        reserve_tagged(sl.d_pring, pr_bytes, sl.s);
        (void)reserve_no_sync(sl.d_gran, gran_bytes);
        (void)reserve_no_sync(sl.d_col, col_bytes);

        InitArgs ia;
        ia.brec = sl.d_brec.as<BRec>();
        ia.trec = sl.d_trec.as<TRec>();
        ia.bprog = sl.bprog();
        ia.strip_start = sl.d_start.as<uint64_t>();
        ia.status = sl.status();
        ia.ticket = sl.ticket();
        ia.pass = sl.pass;
        ia.js1 = init.js1;
        ia.je1 = init.je1;
        ia.ojs1 = init.ojs1;
        ia.oje1 = init.oje1;
        ia.flags1 = init.flags1;
        ia.top1 = init.top1;
        ia.fs0 = init.fs0;
        ia.last_strip = init.last_strip;
        ia.nstrips = geo.nstrips;
        // workgroup 0 sets the pass up, all of them clear the hand-off granules (16 words per thread and round); a large granule
        // buffer (long pairs: a pass takes milliseconds, a launch more does not matter) is left to the runtime's fill kernel
        ia.gran = sl.d_gran.as<uint64_t>();
        ia.gran_words = gran_bytes / 8;
        if (gran_bytes > ((size_t)4 << 20)) {
            ia.gran_words = 0;
            if (!hip_ok(hipMemsetAsync(sl.d_gran.ptr, 0, gran_bytes, sl.s), "memset granules")) hip_fail("memset");
        }
        const uint64_t init_groups = std::min<uint64_t>(std::max<uint64_t>(ia.gran_words / ((uint64_t)kInitThreads * 16), 1), 2048);
        hipLaunchKernelGGL(sweep_init_kernel, dim3((unsigned)init_groups), dim3(kInitThreads), 0, sl.s, ia);

        Ctx c;
        c.a_codes = be.d_codes.as<uint32_t>();
        c.b_prof = be.d_prof.as<uint32_t>();
        c.n = n;
        c.m = m;
        c.nblk = nblk;
        c.wtot = geo.wtot;
        c.f_max = f_max;
        c.pass = sl.pass;
        c.heur = heur_kind;
        c.sparse_h = sparse_h;
        c.sh_h = has_sh ? pool.d_sh.as<int32_t>() : nullptr;
        c.store_cols = trace ? 1 : 0;
        c.d_old = merged_in;
        c.prev_brec = pv_running ? pv->d_brec.as<BRec>() : nullptr;
        c.prev_pass = pv_running ? pv->pass : 0;
        c.prev_done = pv_running ? pv->done() : sl.done();
        c.cancel = sl.cancel();
        c.brec = sl.d_brec.as<BRec>();
        c.trec = sl.d_trec.as<TRec>();
        c.bprog = sl.bprog();
        c.strip_start = sl.d_start.as<uint64_t>();
        c.pring = sl.d_pring.as<uint64_t>();
        c.pr_stride = geo.pr_stride;
        c.gran = sl.d_gran.as<uint64_t>();
        c.gran_stride = geo.gran_stride;
        c.win = geo.win;
        c.col = sl.d_col.as<uint64_t>();
        c.col_stride = geo.col_stride;
        c.col_ring = geo.col_ring;
        c.status = sl.status();
        c.ticket = sl.ticket();
        c.nstrips = geo.nstrips;
        c.nwaves = pass_waves(f_max);
        sl.waves = c.nwaves;
        c.spin_limit = 1u << 19;  // ~2 s of backed-off polls
        c.timing = nullptr;
        if (timing_on()) {
            c.timing = sl.d_misc.as<uint64_t>() + 64;  // bytes 512..575 of d_misc
            (void)hipMemsetAsync(c.timing, 0, 64, sl.s);
            sl.t_launch = engine::now_s();
        }
        hipLaunchKernelGGL(sweep_kernel, dim3((unsigned)c.nwaves), dim3(64), 0, sl.s, c);
        // behind the pass: merge its records into the older ones (after the previous pass's merge); the same launch then publishes
        // the done word and writes the status into the pinned copy that wait_pass reads
        if (pv_running && !hip_ok(hipStreamWaitEvent(sl.s, pv->merged_ev, 0), "hipStreamWaitEvent")) hip_fail("event");
        hipLaunchKernelGGL(sweep_merge_kernel, dim3((unsigned)((nblk + 2 + 255) / 256)), dim3(256), 0, sl.s, sl.d_brec.as<BRec>(), merged_in,
                           sl.d_merged.as<BlockRec>(), sl.status(), nblk, sl.merge_count(), sl.done(), sl.pass, sl.h_status);
        if (!hip_ok(hipEventRecord(sl.merged_ev, sl.s), "hipEventRecord") || !hip_ok(hipGetLastError(), "sweep launch")) hip_fail("sweep pass");
        sl.live = true;
    }

    sweep::Status wait_pass(int seq) {
        SweepSlot& sl = slot_of(seq);
        if (sl.seq != seq) hip_fail("sweep slot lost");
        if (!hip_ok(hipStreamSynchronize(sl.s), "sync")) hip_fail("sweep pass");
        sl.live = false;
        const sweep::Status st = *sl.h_status;
        if (timing_on()) {
            uint64_t tm[8] = {0};
            (void)hipMemcpy(tm, sl.d_misc.as<uint64_t>() + 64, 64, hipMemcpyDeviceToHost);
            std::fprintf(stderr, "sweep pass %u (seq %d): f_max=%d waves=%d state=%u value=%d k_end=%d  %.3f ms after its launch | strip-us: begin %.0f slow %.0f cross %.0f (probes %.0f) end %.0f bottom %.0f plain %.0f gran %.0f flush %.0f\n",
                         sl.pass, seq, sl.f_max, sl.waves, st.state, st.value, st.k_end, (engine::now_s() - sl.t_launch) * 1e3, tm[0] * 0.01, tm[1] * 0.01,
                         tm[6] * 0.01, tm[7] * 0.01, tm[2] * 0.01, tm[3] * 0.01, tm[4] * 0.01, (double)(tm[5] & 0xFFFFFFFFull) * 0.01, (double)(tm[5] >> 32) * 0.01);
        }
        return st;
    }

    // (short pairs: the speculative passes' wavefronts are few and the traceback's kernels small; on C3 waiting first measured better)
    bool cancel_without_waiting() const {
        static const bool off = std::getenv("PA_SWEEP_CANCEL_WAIT") != nullptr;
        return !off && nblk <= kShortPairBlocks;
    }
    // Give up every launched pass behind `seq` and wait until they (and their merges) are gone.
    void cancel_after(int seq, bool wait = true) {
        bool any = false;
        for (SweepSlot& sl : pool.slots)
            if (sl.live && sl.seq > seq) {
                any = hip_ok(hipMemsetD32Async((hipDeviceptr_t)sl.cancel(), (int)sl.pass, 1, pool.ctl), "cancel") || any;
            }
        if (!any || !wait) return;
        (void)hipStreamSynchronize(pool.ctl);
        for (int q = seq + 1; q <= seq + SweepPool::kSlots; ++q) {  // in launch order
            SweepSlot& sl = slot_of(q);
            if (!sl.live || sl.seq <= seq) continue;
            (void)hipStreamSynchronize(sl.s);
            sl.live = false;
        }
    }

    // The blocks of the pass that just succeeded, for Blocks::trace.
    void read_blocks(int seq, std::vector<engine::Block>& blocks) {
        using namespace sweep;
        SweepSlot& sl = slot_of(seq);
        const size_t recs = (size_t)nblk + 1;
        if (!pool.d_recs.reserve(recs * sizeof(BlockOut)) || !pool.d_offs.reserve(recs * 8)) hip_fail("hipMalloc");
        // One synchronisation (round 6): a block's column holds at most col_stride words, so the packed columns fit a pinned buffer of
        // nblk * col_stride words that the gather kernel writes directly; the records come with them.  (Beyond 16 MB -- Mbp pairs --
        // the two-step route below, which sizes the buffer by what the records say.)
        static const bool two_step = std::getenv("PA_SWEEP_READ_TWO_STEP") != nullptr;
        const size_t bound_words = (size_t)nblk * (size_t)sl.geo.col_stride;
        if (!two_step && bound_words * 16 <= (size_t(16) << 20)) {
            const size_t off_cols = (recs * sizeof(BlockOut) + 63) & ~size_t(63);
            uint8_t* hb = static_cast<uint8_t*>(pool.pinned(off_cols + bound_words * 16 + 64));
            if (!hb) hip_fail("pinned");
            BlockOut* hrp = reinterpret_cast<BlockOut*>(hb);
            uint64_t* hp = reinterpret_cast<uint64_t*>(hb + off_cols);
            hipLaunchKernelGGL(sweep_records_offsets_kernel, dim3(1), dim3(1024), 0, be.s, sl.d_brec.as<BRec>(), pool.d_recs.as<BlockOut>(), hrp,
                               pool.d_offs.as<int64_t>(), nblk);
            hipLaunchKernelGGL(sweep_gather_kernel, dim3((unsigned)nblk), dim3(256), 0, be.s, sl.d_col.as<uint64_t>(), sl.geo.col_stride, sl.geo.win,
                               pool.d_recs.as<BlockOut>(), pool.d_offs.as<int64_t>(), hp, nblk);
            if (!hip_ok(hipGetLastError(), "sweep gather launch") || !hip_ok(hipStreamSynchronize(be.s), "sync")) hip_fail("columns");
            int64_t at = 0;
            for (int32_t k = 1; k <= nblk; ++k) {
                engine::Block& bl = blocks[(size_t)k];
                const BlockOut o = hrp[(size_t)k];
                bl.i_range = engine::IRange{(k - 1) * kBlockW, k * kBlockW < n ? k * kBlockW : n};
                bl.original_j_range = engine::JRange{o.ojs, o.oje};
                bl.j_range = engine::JRange{o.js, o.je};
                bl.fixed_j_range = engine::JRange{o.fs, o.fe};
                bl.offset = o.js;
                bl.top_val = o.top_val;
                bl.bot_val = o.bot_val;
                bl.j_h.reset();
                const size_t w = (size_t)(o.je - o.js) / 64;
                if ((size_t)at + w > bound_words) hip_fail("sweep columns beyond their bound");
                bl.v.resize(w);
                std::memcpy(bl.v.data(), hp + at * 2, w * 16);
                at += (int64_t)w;
            }
            return;
        }
        hipLaunchKernelGGL(sweep_records_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, be.s, sl.d_brec.as<BRec>(),
                           pool.d_recs.as<BlockOut>(), nblk);
        std::vector<BlockOut> hr(recs);
        if (!hip_ok(hipMemcpyAsync(hr.data(), pool.d_recs.ptr, recs * sizeof(BlockOut), hipMemcpyDeviceToHost, be.s), "D2H records") ||
            !hip_ok(hipStreamSynchronize(be.s), "sync"))
            hip_fail("records");
        std::vector<int64_t> offs(recs, 0);
        int64_t total = 0;
        for (int32_t k = 1; k <= nblk; ++k) {
            offs[(size_t)k] = total;
            total += (hr[(size_t)k].je - hr[(size_t)k].js) / 64;
        }
        if (!pool.d_pack.reserve((size_t)total * 16 + 16)) hip_fail("hipMalloc");
        uint64_t* hp = static_cast<uint64_t*>(pool.pinned((size_t)total * 16 + 16));
        if (!hp) hip_fail("pinned");
        if (!hip_ok(hipMemcpyAsync(pool.d_offs.ptr, offs.data(), recs * 8, hipMemcpyHostToDevice, be.s), "H2D offsets")) hip_fail("offsets");
        hipLaunchKernelGGL(sweep_gather_kernel, dim3((unsigned)nblk), dim3(256), 0, be.s, sl.d_col.as<uint64_t>(), sl.geo.col_stride, sl.geo.win,
                           pool.d_recs.as<BlockOut>(), pool.d_offs.as<int64_t>(), pool.d_pack.as<uint64_t>(), nblk);
        if (!hip_ok(hipMemcpyAsync(hp, pool.d_pack.ptr, (size_t)total * 16, hipMemcpyDeviceToHost, be.s), "D2H columns") ||
            !hip_ok(hipStreamSynchronize(be.s), "sync"))
            hip_fail("columns");
        for (int32_t k = 1; k <= nblk; ++k) {
            engine::Block& bl = blocks[(size_t)k];
            const BlockOut& o = hr[(size_t)k];
            bl.i_range = engine::IRange{(k - 1) * kBlockW, k * kBlockW < n ? k * kBlockW : n};
            bl.original_j_range = engine::JRange{o.ojs, o.oje};
            bl.j_range = engine::JRange{o.js, o.je};
            bl.fixed_j_range = engine::JRange{o.fs, o.fe};
            bl.offset = o.js;
            bl.top_val = o.top_val;
            bl.bot_val = o.bot_val;
            bl.j_h.reset();
            const size_t w = (size_t)(o.je - o.js) / 64;
            bl.v.resize(w);
            std::memcpy(bl.v.data(), hp + offs[(size_t)k] * 2, w * 16);
        }
    }
};

// More hardware queues for the pipelined passes.  NOT done behind the application's back when the library is loaded (a library that
// edits the process environment at dlopen surprises every other HIP user in the process and races with their getenv): the
// application calls this once, before anything starts the HIP runtime, or exports the variable itself.  Returns 1 if it set the
// variable, 0 if it was set already (left alone).
extern "C" int pa_runtime_hints(void) {
    if (std::getenv("GPU_MAX_HW_QUEUES")) return 0;
    return setenv("GPU_MAX_HW_QUEUES", "16", 0) == 0 ? 1 : 0;
}

// pa_align and the drop-in symbols keep their device buffers, pinned staging and streams in per-thread pools that only grow (one
// 10 Mbp call leaves gigabytes behind): this returns the calling thread's pools to the driver.  The next call builds them again.
extern "C" void pa_release_pools(void) {
    (void)hipDeviceSynchronize();
    sweep_pool_slot().reset();
    pooled_backend_slot().reset();
    release_alloc_cache();  // (after the pools: their buffers land in the cache first)
}

// The engine's bookkeeping without any kernel work (used where the numbers come from a fused GPU pass).
struct StatsOnlyBackend {
    const uint8_t* a_;
    size_t n_;
    const uint8_t* b_;
    size_t m_;
    I n() const { return (I)n_; }
    I m() const { return (I)m_; }
    const uint8_t* a() const { return a_; }
    const uint8_t* b() const { return b_; }
    void enable_h_row() {}
    Cost compute(I, I, size_t, size_t, V*, HMode, const BlockParams&) { return 0; }
    void fill(I, I, size_t, size_t, V*, V*, int8_t*, const BlockParams&) {}
    std::vector<int8_t> debug_read_h(I, I) { return {}; }
    void debug_write_h(I, I, const std::vector<int8_t>&) {}
};

// Shared by pa_align and the astarpa-c symbols.  Returns 0 or a PA_E_* code.
// ---- call combining behind pa_align and the astarpa-c symbols (round 5) ---------------------------------------------------------------
// The reference's entry points are stateless and re-entrant (astarpa-c/src/lib.rs:8-46): a multi-threaded caller aligns one pair per
// thread at a time.  On the GPU one pair at a time is latency bound (a 10 kbp pair: 2 ms through the sweep, whatever else the chip could
// do), while the batch kernels run thousands side by side and return per pair EXACTLY what pa_align returns -- cost, CIGAR string and
// statistics (tests/test_gpu_apa2_batch.py, test_gpu_apa2_full.py, test_gpu_restated_fixtures.py).  So callers that are inside
// pa_align AT THE SAME TIME with the same parameters are combined: a caller that finds nobody gathering gathers -- for 300 us, or until
// everybody who is inside has queued --, aligns the gathered requests as ONE batch (pa_batch_create_params + pa_batch_align) and hands
// the results out; requests that arrive meanwhile are gathered by the next caller, whose batch runs beside the first.  No timer: below a dozen concurrent callers (crowd_threshold below) everybody keeps the
// single-pair path and its latency; above, the batch grows with the number of callers by itself.  PA_COMBINE=0 switches it off.
namespace {
struct CombineReq {
    const uint8_t* a;
    size_t a_len;
    const uint8_t* b;
    size_t b_len;
    int32_t cost = 0;
    std::string cigar;
    pa_astarpa2_stats stats{};
    int rc = 0;
    bool done = false;
    bool queued = false;  // still in the gatherer's pending list (combine_logic.hpp: only such a caller gathers)
    std::string err;
};
struct Combiner {
    pa_astarpa2_params params;  // the key (byte-wise: a parameter set is plain data) ...
    int device = 0;             // ... together with the device the callers are bound to (pa_set_device is per thread): callers on different
                                // GPUs are not mixed, a batch runs on the device of those who asked for it
    combine::Gatherer<CombineReq> g;  // the gathering protocol (combine_logic.hpp; oracle/combine_emu.cpp runs it on host threads under TSan)
};
std::mutex& g_comb_mu = *new std::mutex;
std::vector<Combiner*>& g_combs = *new std::vector<Combiner*>;  // (never destroyed: callers may be inside at exit)
std::atomic<int> g_inside{0};              // callers inside the traced path of align_hip right now
std::atomic<uint64_t> g_comb_calls{0}, g_comb_batches{0};
thread_local bool t_in_combiner = false;   // the leader's own batch may hand a pair back to pa_align's engine: that call is not combined again
constexpr int kNotCombined = 1;
// Longer pairs keep the single-pair engine (many wavefronts per pass).  PA_COMBINE_MAX_LEN overrides (experiments).
inline size_t combine_max_len() {
    const char* e = std::getenv("PA_COMBINE_MAX_LEN");
    return e ? (size_t)std::atoll(e) : (size_t)32768;
}
constexpr size_t kCombineMaxGroup = 8192;
constexpr int kCombineInFlight = 8;        // batches of one parameter set on the GPU at a time
constexpr int kCombineWindowUs = 300;      // how long a gathering caller waits for more callers
// Who takes which route.  A batch costs what its slowest pair costs ONE wavefront -- band search and traceback of a 10 kbp pair at 15 %:
// 6-8 ms -- whatever its size, while the single-pair path runs a pair's passes on many wavefronts (2 ms) and eight callers side by side
// reach 1 300-1 400 pairs/s: combining pays from about a dozen concurrent callers on.  And the two routes do not mix: every single-pair
// call keeps several persistent kernels in flight that poll each other, a batch queued behind them waits (measured: 64 threads, eight of
// them on the single-pair path: 875 pairs/s; all combined: 6 000; sixty-four single-pair calls at once starve one another into their
// bounded waits; profiles/r05_runs/dropin_threads.log).  So the library is in one of two modes: as long as fewer than kCrowd callers are
// inside at a time, everybody takes the single-pair path; once kCrowd are, everybody is combined -- and stays so for kSticky after the
// crowd was last seen (the callers of a finished batch leave together and come back one by one: the first ones back must not find the
// place empty and start single-pair calls again).  PA_COMBINE_MIN overrides kCrowd (tests: 2).
std::atomic<int64_t> g_crowded_until{0};  // steady-clock nanoseconds
constexpr int64_t kStickyNs = 20 * 1000 * 1000;
inline int crowd_threshold() {  // (read at every call: tests switch it inside one process)
    const char* e = std::getenv("PA_COMBINE_MIN");
    const int v = e ? std::atoi(e) : 12;
    return v < 2 ? 2 : v;
}
inline bool combine_now() {
    const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (g_inside.load(std::memory_order_relaxed) >= crowd_threshold()) {
        g_crowded_until.store(now + kStickyNs, std::memory_order_relaxed);
        return true;
    }
    return now < g_crowded_until.load(std::memory_order_relaxed);
}

Combiner& combiner_for(const pa_astarpa2_params& params) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    // the last combiner this thread used: no global lock, no scan of the (never shrinking) list on the common path
    thread_local Combiner* t_last = nullptr;
    if (t_last && t_last->device == dev && std::memcmp(&t_last->params, &params, sizeof(params)) == 0) return *t_last;
    std::lock_guard<std::mutex> lk(g_comb_mu);
    for (Combiner* c : g_combs)
        if (c->device == dev && std::memcmp(&c->params, &params, sizeof(params)) == 0) return *(t_last = c);
    Combiner* c = new Combiner;
    c->params = params;
    c->device = dev;
    g_combs.push_back(c);
    return *(t_last = c);
}

void run_group(std::vector<CombineReq*>& group, const pa_astarpa2_params& params) {
    const size_t n = group.size();
    std::vector<const uint8_t*> ap(n), bp(n);
    std::vector<size_t> al(n), bl(n);
    for (size_t i = 0; i < n; ++i) {
        ap[i] = group[i]->a;
        bp[i] = group[i]->b;
        al[i] = group[i]->a_len;
        bl[i] = group[i]->b_len;
    }
    std::vector<int32_t> costs(n, 0);
    std::vector<pa_astarpa2_stats> st(n);
    // RAII (round 5's advisor): a std::string assignment below may throw; the CIGARs the batch malloc'ed and the batch itself go either way,
    // and no request is left half filled (rc is written last, per request, and the caller's catch sets rc_failed for the whole group)
    struct Cigars {
        std::vector<char*> p;
        explicit Cigars(size_t k) : p(k, nullptr) {}
        ~Cigars() {
            for (char* q : p) std::free(q);
        }
    } cigars(n);
    struct InCombiner {
        InCombiner() { t_in_combiner = true; }
        ~InCombiner() { t_in_combiner = false; }
    };
    int rc = 0;
    {
        InCombiner guard;
        std::unique_ptr<pa_batch, void (*)(pa_batch*)> bt(pa_batch_create_params(ap.data(), al.data(), bp.data(), bl.data(), n, &params), pa_batch_destroy);
        if (!bt) rc = kNotCombined;  // (every caller falls back to the single-pair path, which reports its own errors)
        else {
            rc = pa_batch_align(bt.get(), costs.data(), cigars.p.data(), nullptr, nullptr);
            if (rc == 0) rc = pa_batch_pair_stats(bt.get(), st.data());
            if (rc != 0) rc = kNotCombined;
        }
    }
    for (size_t i = 0; i < n; ++i) {
        CombineReq& r = *group[i];
        if (rc == 0) {
            r.cigar = cigars.p[i] ? cigars.p[i] : "";  // (may throw: nothing of r has been touched yet)
            r.cost = costs[i];
            r.stats = st[i];
        }
        r.rc = rc;
    }
    g_comb_calls += n;
    g_comb_batches += 1;
}

// 0: done (results filled in); kNotCombined: the caller runs the single-pair path.
int combine_align(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, const pa_astarpa2_params& params, int32_t* cost_out,
                  std::string* cigar_out, pa_astarpa2_stats* stats_out) {
    Combiner& c = combiner_for(params);
    CombineReq req{a, a_len, b, b_len};
    // A call lasts as long as its batch, and a batch of short pairs takes about 6 ms whatever its size, so N callers complete N calls per
    // (batch + window): the window is cheap and decides the batch size; several batches run side by side on streams of their own.
    c.g.submit(
        req,
        [&](std::vector<CombineReq*>& group) {
            try {
                run_group(group, params);
            } catch (...) {  // (out of host memory while gathering: every caller of the group takes the single-pair path)
                t_in_combiner = false;
                throw;
            }
        },
        [] { return g_inside.load(std::memory_order_relaxed); }, kCombineMaxGroup, kCombineInFlight, kCombineWindowUs, kNotCombined);
    if (req.rc != 0) return kNotCombined;
    if (cost_out) *cost_out = req.cost;
    if (cigar_out) *cigar_out = std::move(req.cigar);
    if (stats_out) *stats_out = req.stats;
    return 0;
}
struct InsideGuard {
    InsideGuard() { g_inside.fetch_add(1, std::memory_order_relaxed); }
    ~InsideGuard() { g_inside.fetch_sub(1, std::memory_order_relaxed); }
};
}  // namespace

static std::atomic<int> g_reference_cost_only{0};
static bool reference_cost_only_requested() {
    if (g_reference_cost_only.load(std::memory_order_relaxed)) return true;
    const char* e = std::getenv("PA_COST_ONLY_MODE");
    return e && std::strcmp(e, "reference") == 0;
}
extern "C" void pa_set_reference_cost_only(int on) { g_reference_cost_only.store(on ? 1 : 0, std::memory_order_relaxed); }

// Diagnostics: calls served through the combiner so far, and the batches they went out in.
extern "C" void pa_combine_stats(uint64_t* calls, uint64_t* batches) {
    if (calls) *calls = g_comb_calls.load();
    if (batches) *batches = g_comb_batches.load();
}

int align_hip(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, const pa_astarpa2_params& params,
              bool trace, bool self_check, int32_t* cost_out, std::string* cigar_out, pa_astarpa2_stats* stats_out) {
    if (!engine::params_valid(params)) {
        set_error("invalid A*PA2 parameters");
        return PA_E_ARG;
    }
    if (a_len > (size_t)(1u << 30) || b_len > (size_t)(1u << 30)) {
        set_error("sequence too long for i32 coordinates");
        return PA_E_ARG;
    }
    const engine::AstarPa2Params p = engine::params_from_c(params);
    if (!trace && !self_check && a_len > 0 && b_len > 0 && p.domain == engine::DomainKind::Full && p.doubling == engine::DoublingKind::None) {
        // AstarPa2Params::nw().make_aligner(false): the whole matrix, cost only (blocks.rs:252-277: one block updated in
        // place, 256 columns per operator call).  The values do not depend on the schedule, so the cost comes from ONE
        // launch of chained strips (pa_batch of one pair) instead of |a|/256 launches; the statistics come from the host
        // engine walked over a backend that computes nothing (they depend on the lengths only).
        const uint8_t* aa[1] = {a};
        const uint8_t* bb[1] = {b};
        const size_t al[1] = {a_len}, bl[1] = {b_len};
        pa_batch* bt = pa_batch_create(aa, al, bb, bl, 1);
        if (!bt) return PA_E_HIP;
        int32_t c = 0;
        const int rc = pa_batch_run(bt, &c, nullptr);
        pa_batch_destroy(bt);
        if (rc != 0) return rc;
        if (stats_out) {
            StatsOnlyBackend sb{a, a_len, b, b_len};
            try {
                const engine::AlignResult r = engine::cost_or_align(p, sb, false, false);
                engine::stats_to_c(r.stats, stats_out);
            } catch (const engine::EnginePanic& e) {
                set_error("astarpa2 engine panic: %s", e.what());
                return PA_E_INTERNAL;
            }
        }
        if (cost_out) *cost_out = c;
        if (cigar_out) cigar_out->clear();
        return 0;
    }
    // Several callers inside at once, a parameter set the batch kernels take: one batch for all of them (see combine_align above)
    // (only callers that COULD be combined count as the crowd: a lone short-pair caller among many cost-only / long-pair / unsupported
    //  callers keeps the 2 ms single-pair path instead of a 300 us window plus a 6-8 ms batch -- round 5's advisor)
    std::optional<InsideGuard> inside;
    if (trace && !self_check && !t_in_combiner && a_len > 0 && b_len > 0 && a_len < combine_max_len() && b_len < combine_max_len() && pa_batch_params_supported(&params)) {
        inside.emplace();
        static const bool combine_off = std::getenv("PA_COMBINE") != nullptr && std::getenv("PA_COMBINE")[0] == '0';
        if (!combine_off && combine_now() && combine_align(a, a_len, b, b_len, params, cost_out, cigar_out, stats_out) == 0) return 0;
    }
    HipBackend& be = pooled_backend();
    be.bind(a, a_len, b, b_len);
    if (!be.ok) return be.err ? be.err : PA_E_HIP;
    engine::AlignResult r;
    // Domain::Astar with a closed-form / per-column heuristic and the sparse, non-incremental block engine (the `simple`
    // preset and its relatives): every align_for_bounded_dist pass is ONE persistent launch with the band logic in the kernel
    // (sweep_wave.hpp).  A pass the kernel hands back (SweepFallback) is redone by the host-driven engine below.
    // pa_set_reference_cost_only(1) / PA_COST_ONLY_MODE=reference: trace == 0 runs the REFERENCE's cost-only arm (blocks.rs:252-277: one block
    // updated in place) through the host-driven engine over the HIP kernels -- the value the reference's make_aligner(false) returns, upper
    // bounds included (include/pa_astarpa2.h, DEVIATIONS.md)
    const bool ref_cost_only = !trace && reference_cost_only_requested();
    const bool no_sweep = ref_cost_only || std::getenv("PA_ENGINE_NO_SWEEP") != nullptr;  // (PA_ENGINE_NO_SWEEP: diagnostics / tests, the host-driven engine)
    if (!no_sweep && !self_check && sweep::sweep_supported(p, a_len, b_len)) {
        try {
            HipSweepLauncher launcher(be, sweep_pool());
            launcher.heur_kind = p.heuristic == engine::HeuristicKind::Gap ? sweep::kHeurGap
                                 : p.heuristic == engine::HeuristicKind::SH ? sweep::kHeurSH : sweep::kHeurNone;
            sweep::SweepAligner<HipBackend, HipSweepLauncher> al(p, be, launcher, trace);
            r = al.align();
            if (cost_out) *cost_out = r.cost;
            if (cigar_out) *cigar_out = r.has_cigar ? r.cigar.to_string() : std::string();
            if (stats_out) engine::stats_to_c(r.stats, stats_out);
            return 0;
        } catch (const sweep::SweepFallback& e) {
            if (std::getenv("PA_SWEEP_TIMING")) std::fprintf(stderr, "sweep fallback: %s (%d)\n", e.what(), e.reason);
            be.bind(a, a_len, b, b_len);  // fresh backend state for the host-driven engine
            if (!be.ok) return be.err ? be.err : PA_E_HIP;
        } catch (const engine::EnginePanic& e) {
            if (be.err) return be.err;
            set_error("astarpa2 engine panic: %s", e.what());
            return PA_E_INTERNAL;
        }
    }
    // Cost only, for the parameters the sweep serves: the sweep computes the band of the TRACED mode (its answer is the distance
    // itself, its statistics those of the traced band; include/pa_astarpa2.h).  A pair it hands back gets the same: the host-driven
    // engine in traced mode with the CIGAR dropped -- not the reference's cost-only single-block mode (blocks.rs:252-277), whose
    // fixed range is the union over all columns (a triangle of the matrix) and which this restatement has seen end on an upper
    // bound.  PA_ENGINE_NO_SWEEP (tests, diagnostics) still runs that mode as restated.
    const bool traced_for_cost = !trace && !no_sweep && !self_check && sweep::sweep_supported(p, a_len, b_len);
    try {
        r = engine::cost_or_align(p, be, trace || traced_for_cost, self_check);
    } catch (const engine::EnginePanic& e) {
        if (be.err) return be.err;
        set_error("astarpa2 engine panic: %s", e.what());
        return PA_E_INTERNAL;
    }
    if (traced_for_cost) {
        r.has_cigar = false;
        r.cigar = engine::Cigar();
        r.stats.trace_stats = engine::TraceStats();
    }
    if (cost_out) *cost_out = r.cost;
    if (cigar_out) *cigar_out = r.has_cigar ? r.cigar.to_string() : std::string();
    if (stats_out) engine::stats_to_c(r.stats, stats_out);
    return 0;
}

}  // namespace pa

using namespace pa;

// ---- device-resident operator handles (route 2 of INTEGRATION.md: the reference's engine over HIP operators) ------------------
// The sequences, the profile and the persistent h row of one pair stay on the GPU between calls; a call moves only the v
// words of its rectangle.  This is HipBackend behind a C handle.
struct pa_bp_ctx {
    HipBackend be;
    engine::BlockParams bp;
    int device = -1;  // the device its buffers and its stream live on
};
// A handle launches on the device it was created on: a call from a thread whose current device is another one is refused instead of
// launching there with pointers of this one.
static bool ctx_device_ok(const pa_bp_ctx* c, const char* what) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device) {
        set_error("%s: the handle belongs to device %d, this thread's current device is %d (pa_set_device first)", what, c->device, cur);
        return false;
    }
    return true;
}

extern "C" pa_bp_ctx* pa_bp_ctx_create(const uint8_t* a, size_t n, const uint8_t* b, size_t m) {
    if (n > (size_t)(1u << 30) || m > (size_t)(1u << 30)) {
        set_error("sequence too long for i32 coordinates");
        return nullptr;
    }
    std::unique_ptr<pa_bp_ctx> c(new (std::nothrow) pa_bp_ctx());
    if (!c) {
        set_error("out of memory");
        return nullptr;
    }
    c->be.bind(a, n, b, m);
    if (!c->be.ok) return nullptr;
    (void)hipGetDevice(&c->device);
    try {
        c->be.enable_h_row();
    } catch (const engine::EnginePanic&) {
        return nullptr;
    }
    return c.release();
}

extern "C" int pa_bp_ctx_compute(pa_bp_ctx* c, int32_t i0, int32_t i1, size_t w0, size_t w1, uint64_t* v, int h_mode, int32_t* sum_out) {
    if (!c || i0 < 0 || i1 < i0 || i1 > c->be.n() || w1 < w0 || w1 > (size_t)((c->be.m() + 63) / 64) || (!v && w1 > w0) || h_mode < 0 || h_mode > 3) {
        set_error("pa_bp_ctx_compute: bad arguments");
        return PA_E_ARG;
    }
    if (!ctx_device_ok(c, "pa_bp_ctx_compute")) return PA_E_ARG;
    static_assert(sizeof(engine::V) == 16, "V is (p: u64, m: u64)");
    try {
        const engine::Cost s = c->be.compute(i0, i1, w0, w1, reinterpret_cast<engine::V*>(v), (engine::HMode)h_mode, c->bp);
        if (sum_out) *sum_out = s;
    } catch (const engine::EnginePanic&) {
        return c->be.err ? c->be.err : PA_E_INTERNAL;
    }
    return 0;
}

extern "C" int pa_bp_ctx_fill(pa_bp_ctx* c, int32_t i0, int32_t i1, size_t w0, size_t w1, uint64_t* v, uint64_t* values, int8_t* h_bottom) {
    if (!c || i0 < 0 || i1 < i0 || i1 > c->be.n() || w1 < w0 || w1 > (size_t)((c->be.m() + 63) / 64) || !v || !values) {
        set_error("pa_bp_ctx_fill: bad arguments");
        return PA_E_ARG;
    }
    if (!ctx_device_ok(c, "pa_bp_ctx_fill")) return PA_E_ARG;
    try {
        std::vector<int8_t> hb((size_t)(i1 - i0) + 1, 0);
        c->be.fill(i0, i1, w0, w1, reinterpret_cast<engine::V*>(v), reinterpret_cast<engine::V*>(values), hb.data(), c->bp);
        if (h_bottom) std::memcpy(h_bottom, hb.data(), (size_t)(i1 - i0));
    } catch (const engine::EnginePanic&) {
        return c->be.err ? c->be.err : PA_E_INTERNAL;
    }
    return 0;
}

extern "C" void pa_bp_ctx_destroy(pa_bp_ctx* c) { delete c; }

extern "C" void pa_params_nw(pa_astarpa2_params* p) { engine::params_to_c(engine::AstarPa2Params::nw(), p); }
extern "C" void pa_params_simple(pa_astarpa2_params* p) { engine::params_to_c(engine::AstarPa2Params::simple(), p); }
extern "C" void pa_params_full(pa_astarpa2_params* p) { engine::params_to_c(engine::AstarPa2Params::full(), p); }

extern "C" int pa_align(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, const pa_astarpa2_params* params,
                        int trace, int32_t* cost_out, char** cigar_out, pa_astarpa2_stats* stats_out) {
    if (!params) return PA_E_ARG;
    std::string cigar;
    const bool self_check = std::getenv("PA_ENGINE_SELF_CHECK") != nullptr;
    const int rc = align_hip(a, a_len, b, b_len, *params, trace != 0, self_check, cost_out, &cigar, stats_out);
    if (cigar_out) {
        *cigar_out = nullptr;
        if (rc == 0 && trace) {
            *cigar_out = (char*)std::malloc(cigar.size() + 1);
            if (!*cigar_out) {
                set_error("out of memory");
                return PA_E_NOMEM;
            }
            std::memcpy(*cigar_out, cigar.c_str(), cigar.size() + 1);
        }
    }
    return rc;
}
