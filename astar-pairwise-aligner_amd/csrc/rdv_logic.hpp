// rdv_logic.hpp -- the rendezvous of the batched A*PA2 kernels: two wavefronts of one workgroup whose blocks both fit a HALF wave run
// their two strips as ONE (strip2_kernel.hpp: pair A in lanes 0..31, pair B in lanes 32..63).
//
// Why: a 256-column block of a short or similar pair is about ten 64-row words tall -- twenty of a wavefront's 64 lanes -- and its
// strip is 60-85 % of everything the band-search kernels issue (apa2_kernel.hpp, apa2_full_kernel.hpp: both are bound by VALU issue).
// Two such strips side by side cost 25 instead of 2 x 23 (24) instructions per step.  The reference has no counterpart: it aligns
// one pair at a time on one core (pa-bin/src/main.rs:24-35 over astarpa2/src/domain.rs:356-541).
//
// How, without a second program per wavefront: every wavefront keeps running its own pair's band search.  When it reaches a block
// that fits half a wave it writes the strip's description into its mail area and looks at one shared word:
//   * somebody else has POSTED: take that job (one compare-and-swap), run the fused strip for both, mark it DONE;
//   * nobody has: post its own (the same compare-and-swap, so at most ONE wavefront waits at any time and two never wait for each
//     other), sleep (s_sleep issues nothing) until a partner marks it DONE -- or until its patience runs out or it is the last
//     wavefront of the workgroup still working, then withdraw (again one compare-and-swap) and run alone.
// A taker never waits; a poster waits for a running strip at most: nothing can dead-lock, every wait is bounded.
// All of a strip's state sits in global memory behind the job's pointers, so the results need no way back.
//
// The protocol is written once against a policy P (the shared word, the clock, the nap):
//   * apa2_kernel.hpp / apa2_full_kernel.hpp: LDS atomics at workgroup scope (RdvLds below strip2_kernel.hpp);
//   * oracle/rdv_emu.cpp (tests only): std::atomic on host threads, under ThreadSanitizer.
// P provides:  uint32_t load()                      the state word (acquire)
//              bool cas(uint32_t expect, uint32_t desired)
//              void add(uint32_t delta)              (release; delta may wrap: used to move ONE byte up or down)
//              uint32_t live()                       wavefronts of the workgroup still inside their pair loop
//              uint64_t now(); void nap()
#pragma once
#include <stdint.h>

#include "sweep_logic.hpp"  // PA_HD

namespace pa {
namespace rdv {

// One byte per wavefront of the workgroup: the state of ITS offer.
enum : uint32_t { kNone = 0, kPosted = 1, kTaken = 2, kDone = 3 };
enum : int32_t {
    kAlone = 0,   // run the strip alone (nobody came, or nobody is left)
    kTook = 1,    // *partner's job is ours to run next to our own; then finish(partner)
    kServed = 2,  // a partner ran our strip: the results are in memory
    kStuck = 3,   // a partner took our job and never finished (bounded wait exceeded): a device failure
};
constexpr int kMaxWaves = 4;

PA_HD uint32_t byte_of(uint32_t st, int w) { return (st >> (8 * w)) & 0xFFu; }
PA_HD uint32_t with_byte(uint32_t st, int w, uint32_t v) { return (st & ~(0xFFu << (8 * w))) | (v << (8 * w)); }

struct Counters {  // diagnostics
    uint32_t took = 0, served = 0, alone = 0, withdrawn = 0;
};

// Wavefront `w` (0 .. nwaves - 1) has written its job into its mail area and arrives.  `patience`: clock ticks a posted job waits for a
// partner; `hard`: ticks after which a TAKEN job that does not become DONE is a failure.
template <class P>
PA_HD int32_t arrive(P& p, int w, int nwaves, uint64_t patience, uint64_t hard, int* partner, Counters* cnt) {
    for (;;) {
        if (p.live() <= 1u) {
            cnt->alone += 1;
            return kAlone;
        }
        const uint32_t cur = p.load();
        int v = -1;
        for (int x = 0; x < nwaves; ++x)
            if (x != w && byte_of(cur, x) == kPosted) v = v < 0 ? x : v;
        if (v >= 0) {
            if (p.cas(cur, with_byte(cur, v, kTaken))) {
                *partner = v;
                cnt->took += 1;
                return kTook;
            }
            continue;  // (somebody else took it, or it was withdrawn: look again)
        }
        if (p.cas(cur, with_byte(cur, w, kPosted))) break;  // nobody is posted: we are (atomically: at most one wavefront waits)
    }
    const uint64_t t0 = p.now();
    for (;;) {
        p.nap();
        const uint32_t cur = p.load();
        const uint32_t b = byte_of(cur, w);
        if (b == kDone) {
            p.add(0u - (kDone << (8 * w)));  // back to kNone (only this wavefront moves its byte from kDone)
            cnt->served += 1;
            return kServed;
        }
        const uint64_t waited = p.now() - t0;
        if (b == kPosted && (waited > patience || p.live() <= 1u)) {
            if (p.cas(cur, with_byte(cur, w, kNone))) {
                cnt->withdrawn += 1;
                cnt->alone += 1;
                return kAlone;
            }
            continue;  // (the word changed: taken just now, or another byte moved)
        }
        if (b == kTaken && waited > hard) return kStuck;
    }
}

// The taker, after the fused strip's results are in memory: the partner's job moves from kTaken to kDone.
template <class P>
PA_HD void finish(P& p, int partner) {
    p.add((kDone - kTaken) << (8 * partner));
}

}  // namespace rdv
}  // namespace pa
