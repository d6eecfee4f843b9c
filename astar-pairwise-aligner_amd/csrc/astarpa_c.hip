// astarpa_c.hip -- the reference's C ABI (astarpa-c/src/lib.rs:8-101, astarpa-c/astarpa.h:15-65),
// exported verbatim by libastarpa_c_hip.so and served by the HIP-backed block engine.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/astarpa.h"
#include "../../include/pa_astarpa2.h"
#include "../../include/pa_bitpacking_hip.h"

namespace pa {
int align_hip(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, const pa_astarpa2_params& params,
              bool trace, bool self_check, int32_t* cost_out, std::string* cigar_out, pa_astarpa2_stats* stats_out);
}

namespace {

// Shared tail of every entry point: cigar.to_string() -> CString::into_raw, *cigar_len = strlen (lib.rs:19-23).
uint64_t run(const uint8_t* a, uintptr_t a_len, const uint8_t* b, uintptr_t b_len, const pa_astarpa2_params& p,
             uint8_t** cigar_ptr, uintptr_t* cigar_len, const char* who) {
    int32_t cost = 0;
    std::string cigar;
    int rc = pa::align_hip(a, a_len, b, b_len, p, true, false, &cost, &cigar, nullptr);
    if (rc == PA_E_TIMEOUT) rc = pa::align_hip(a, a_len, b, b_len, p, true, false, &cost, &cigar, nullptr);  // a device-side bounded
    // spin expired (another process starving the GPU): the buffers are re-initialised, try once more before giving up
    if (rc != 0) {
        // The reference has no error channel: invalid input panics across FFI (lib.rs has no Result).
        std::fprintf(stderr, "%s: fatal: %s (rc=%d)\n", who, pa_last_error(), rc);
        std::abort();
    }
    char* out = (char*)std::malloc(cigar.size() + 1);
    if (!out) {
        std::fprintf(stderr, "%s: fatal: out of memory\n", who);
        std::abort();
    }
    std::memcpy(out, cigar.c_str(), cigar.size() + 1);
    if (cigar_len) *cigar_len = cigar.size();
    if (cigar_ptr) *cigar_ptr = (uint8_t*)out;
    else std::free(out);
    return (uint64_t)cost;
}

}  // namespace

extern "C" uint64_t astarpa2_simple(const uint8_t* a, uintptr_t a_len, const uint8_t* b, uintptr_t b_len,
                                    uint8_t** cigar_ptr, uintptr_t* cigar_len) {
    pa_astarpa2_params p;
    pa_params_simple(&p);  // astarpa2::astarpa2_simple, astarpa2/src/lib.rs:43-47
    return run(a, a_len, b, b_len, p, cigar_ptr, cigar_len, "astarpa2_simple");
}

extern "C" uint64_t astarpa2_full(const uint8_t* a, uintptr_t a_len, const uint8_t* b, uintptr_t b_len,
                                  uint8_t** cigar_ptr, uintptr_t* cigar_len) {
    pa_astarpa2_params p;
    pa_params_full(&p);  // astarpa2::astarpa2_full, astarpa2/src/lib.rs:49-53 (GCSH k=12 p=14, pruning, incremental doubling)
    return run(a, a_len, b, b_len, p, cigar_ptr, cigar_len, "astarpa2_full");
}

// A*PA v1 (priority-queue A*) is outside the data-parallel hot path: the symbols are kept so the
// library links as a drop-in, and are served by the same exact block engine -- identical cost, a valid optimal
// CIGAR, not necessarily v1's tie-breaks.  r, k, prune_end only tune v1's heuristic and do not affect the result.
extern "C" uint64_t astarpa(const uint8_t* a, uintptr_t a_len, const uint8_t* b, uintptr_t b_len,
                            uint8_t** cigar_ptr, uintptr_t* cigar_len) {
    return astarpa_gcsh(a, a_len, b, b_len, 2, 15, false, cigar_ptr, cigar_len);  // lib.rs:54-65
}

extern "C" uint64_t astarpa_gcsh(const uint8_t* a, uintptr_t a_len, const uint8_t* b, uintptr_t b_len, uintptr_t r,
                                 uintptr_t k, bool prune_end, uint8_t** cigar_ptr, uintptr_t* cigar_len) {
    (void)r;
    (void)k;
    (void)prune_end;
    pa_astarpa2_params p;
    pa_params_simple(&p);
    return run(a, a_len, b, b_len, p, cigar_ptr, cigar_len, "astarpa_gcsh");
}

extern "C" void astarpa_free_cigar(uint8_t* cigar) { std::free(cigar); }  // lib.rs:99-101
// All strings of a batch result at once (bindings: one call instead of one per pair); entries may be NULL, they are set to NULL.
extern "C" void pa_free_cigars(char** cigars, size_t n) {
    if (!cigars) return;
    for (size_t i = 0; i < n; ++i) {
        std::free(cigars[i]);
        cigars[i] = nullptr;
    }
}
