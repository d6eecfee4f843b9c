// combine_logic.hpp -- the gathering protocol of the call combiner behind pa_align and the astarpa-c symbols (engine_hip.hip), host
// code only, written once so that it can run without a GPU: oracle/combine_emu.cpp drives it from many host threads with a stand-in for
// the batch, under ThreadSanitizer (tests/test_combine_emu.py).
//
// What it replaces: nothing in the reference -- astarpa-c's entry points are stateless and re-entrant (astarpa-c/src/lib.rs:8-46) and a
// multi-threaded caller aligns one pair per thread at a time; here callers that are inside at the same time become one batch.
//
// The protocol: a caller queues its request.  Somebody has to gather a batch: the first caller that finds nobody gathering (and fewer
// than `max_in_flight` batches running) does -- for `window_us`, or until everybody who is inside has queued -- then takes what is
// queued, runs it as ONE batch OUTSIDE the lock (`run(group)` fills in every request's result and rc), marks the requests done and wakes
// their owners.  Requests that arrive meanwhile are gathered by the next caller, whose batch runs beside the first.  A request is a stack
// object of its owner, who does not return before it is done.  A batch that throws releases its group with rc = `rc_failed`: nobody is
// left waiting.
#pragma once
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

namespace pa {
namespace combine {

// Req: any struct with `bool done`, `bool queued` (both false at submission) and `int rc`.
template <class Req>
struct Gatherer {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Req*> pending;
    bool collecting = false;  // a caller is gathering the next batch
    int in_flight = 0;        // batches running right now

    // Returns when `req.done`.  `inside()`: how many callers are inside the library right now (the gatherer stops waiting once all of
    // them have queued).
    template <class Run, class Inside>
    void submit(Req& req, Run&& run, Inside&& inside, size_t max_group, int max_in_flight, int window_us, int rc_failed) {
        std::unique_lock<std::mutex> lk(mu);
        req.queued = true;
        pending.push_back(&req);
        cv.notify_all();  // (a gathering caller counts the arrivals)
        while (!req.done) {
            // Only a caller whose own request is still QUEUED gathers.  One whose request already travels in a running batch just waits for
            // it (round 5 let it gather too: it woke on the leader's notify, waited the whole window and often ran an EMPTY group -- a batch
            // of 0 pairs, one of the in-flight slots, and the caller returned a window plus a batch later than its result was there).
            if (!req.queued || collecting || in_flight >= max_in_flight) {
                cv.wait(lk);
                continue;
            }
            collecting = true;
            // (system_clock: condition_variable::wait_until on it is pthread_cond_timedwait, which ThreadSanitizer understands; the steady
            //  clock's pthread_cond_clockwait it does not -- and for a window of a few hundred microseconds the clocks do not differ)
            const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(window_us);
            while (pending.size() < max_group && (int)pending.size() < inside() && cv.wait_until(lk, deadline) != std::cv_status::timeout) {
            }
            std::vector<Req*> group;
            if (pending.size() <= max_group) group.swap(pending);
            else {
                group.assign(pending.begin(), pending.begin() + (long)max_group);
                pending.erase(pending.begin(), pending.begin() + (long)max_group);
            }
            for (Req* r : group) r->queued = false;
            collecting = false;
            if (group.empty()) {  // (cannot happen while the gatherer's own request is queued; kept so that an empty batch is never run)
                cv.notify_all();
                continue;
            }
            in_flight += 1;
            cv.notify_all();  // (whoever is still pending may gather the next batch)
            lk.unlock();
            try {
                run(group);
            } catch (...) {
                for (Req* r : group) r->rc = rc_failed;
            }
            lk.lock();
            for (Req* r : group) r->done = true;
            in_flight -= 1;
            cv.notify_all();  // the served ones leave
        }
    }
};

}  // namespace combine
}  // namespace pa
