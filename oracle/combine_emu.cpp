/*
 * oracle/combine_emu.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The gathering protocol of the call combiner (astar-pairwise-aligner_amd/csrc/combine_logic.hpp: concurrent callers of pa_align become
 * one batch) WITHOUT a GPU: T host threads submit requests through the very template engine_hip.hip instantiates; the "batch" is a
 * stand-in that takes a while and computes a function of every request's input.  What it checks: every caller gets ITS result, nobody is
 * left waiting (lost wake-ups, a batch that throws), requests really travel in groups, several batches run side by side -- and, built
 * with -fsanitize=thread (`make -C oracle tsan_combine`), that the requests (stack objects of their owners, filled in by another thread)
 * are ordered by the protocol's mutex alone.  Usage of the driver: oracle/_build/combine_emu_tsan [threads] [calls per thread].
 */
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../astar-pairwise-aligner_amd/csrc/combine_logic.hpp"

namespace {
struct Req {
    uint64_t x;
    uint64_t y = 0;
    int rc = 0;
    bool done = false;
    bool queued = false;
    long long t_done = 0;  // when the batch that served it finished (steady clock ticks)
};
inline uint64_t f(uint64_t x) { return x * 0x9E3779B97F4A7C15ull + 12345; }
}  // namespace

// T threads x `calls` requests each.  fail_every > 0: every fail_every-th batch throws.  out[0] = wrong results, out[1] = batches,
// out[2] = largest group, out[3] = most batches running at once, out[4] = requests whose batch failed (rc != 0), out[5] = EMPTY groups that were run,
// out[6] = callers that returned more than a (window + batch) + 50 ms after their own batch was done (generous: host threads
// outnumber the cores; the structural symptom of the round-5 flaw is out[5]).  Returns 0.
extern "C" int pa_combine_emu_run(int threads, int calls, int batch_us, int fail_every, int64_t* out) {
    pa::combine::Gatherer<Req> g;
    std::atomic<int> inside{0}, running{0}, max_running{0}, batches{0}, max_group{0};
    std::atomic<int64_t> wrong{0}, failed{0}, empty_groups{0}, late_returns{0};
    auto run = [&](std::vector<Req*>& group) {
        const int now = running.fetch_add(1) + 1;
        int m = max_running.load();
        while (now > m && !max_running.compare_exchange_weak(m, now)) {
        }
        if (group.empty()) empty_groups.fetch_add(1);
        const int b = batches.fetch_add(1) + 1;
        int mg = max_group.load();
        while ((int)group.size() > mg && !max_group.compare_exchange_weak(mg, (int)group.size())) {
        }
        std::this_thread::sleep_for(std::chrono::microseconds(batch_us));
        if (fail_every > 0 && b % fail_every == 0) {
            running.fetch_sub(1);
            throw std::runtime_error("stand-in batch failed");
        }
        const auto t_done = std::chrono::steady_clock::now().time_since_epoch().count();
        for (Req* r : group) {
            r->y = f(r->x);
            r->rc = 0;
            r->t_done = t_done;
        }
        running.fetch_sub(1);
    };
    auto worker = [&](int t) {
        for (int c = 0; c < calls; ++c) {
            inside.fetch_add(1);
            Req req{(uint64_t)t * 1000003ull + (uint64_t)c};
            g.submit(req, run, [&] { return inside.load(); }, 64, 4, 200, 7);
            // a caller returns as soon as its own batch is done: not a window (200 us) plus another batch (batch_us) later
            const long long t_ret = std::chrono::steady_clock::now().time_since_epoch().count();
            if (req.rc == 0 && req.t_done && (t_ret - req.t_done) > (long long)(batch_us + 200) * 1000 + 50000000) late_returns.fetch_add(1);
            if (req.rc == 7) failed.fetch_add(1);
            else if (req.rc != 0 || req.y != f(req.x)) wrong.fetch_add(1);
            inside.fetch_sub(1);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
    out[0] = wrong.load();
    out[1] = batches.load();
    out[2] = max_group.load();
    out[3] = max_running.load();
    out[4] = failed.load();
    out[5] = empty_groups.load();
    out[6] = late_returns.load();
    return 0;
}
