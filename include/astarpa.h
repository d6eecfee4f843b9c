/*
 * astarpa.h -- the reference's C ABI, exported verbatim by libastarpa_c_hip.so.
 *
 * Each prototype replaces the identically named symbol of the reference's `astarpa-c` crate
 * (/root/reference/astarpa-c/astarpa.h:15-65, implemented at astarpa-c/src/lib.rs:8-101).
 * Conventions kept from the reference:
 *   - `a`,`b` are borrowed for the duration of the call (lib.rs:17-18);
 *   - `*cigar_ptr` is a heap, NUL-terminated string owned by the library; release it with
 *     astarpa_free_cigar (lib.rs:22,99-101); `*cigar_len` excludes the NUL (lib.rs:21);
 *   - the return value is the edit distance; there is no error channel.  Where the reference
 *     panics (a base outside "ACGT", profile.rs:113-126) this library prints a message and abort()s.
 *   - stateless and re-entrant from the caller's point of view.
 * The DP rectangles are computed by hand-written HIP kernels on an MI355X; a GPU is required.
 */
#ifndef ASTARPA_H
#define ASTARPA_H

#include <stdarg.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* astarpa-c/astarpa.h:15-25, src/lib.rs:8-24: A*PA2-simple (GapCost band doubling, sparse blocks, DT trace). */
uint64_t astarpa2_simple(const uint8_t *a, uintptr_t a_len, const uint8_t *b, uintptr_t b_len,
                         uint8_t **cigar_ptr, uintptr_t *cigar_len);

/* astarpa-c/astarpa.h:27-37, src/lib.rs:30-46: A*PA2-full. */
uint64_t astarpa2_full(const uint8_t *a, uintptr_t a_len, const uint8_t *b, uintptr_t b_len,
                       uint8_t **cigar_ptr, uintptr_t *cigar_len);

/* astarpa-c/astarpa.h:39-51, src/lib.rs:54-65: A*PA v1 entry point.
 * DEVIATION (stated here, not only in INTEGRATION.md): A*PA v1 is a priority-queue A* outside this library's scope.  The two v1
 * symbols below are served by the A*PA2-simple engine: the returned cost is the exact edit distance and the CIGAR is a valid
 * optimal alignment, but NOT necessarily v1's choice among equally good alignments (astarpa-c/example.cpp:16 pins v1's "=I4=X=");
 * `r`, `k` and `prune_end` only tune v1's heuristic and are ignored. */
uint64_t astarpa(const uint8_t *a, uintptr_t a_len, const uint8_t *b, uintptr_t b_len,
                 uint8_t **cigar_ptr, uintptr_t *cigar_len);

/* astarpa-c/astarpa.h:53-63, src/lib.rs:69-96 (see the DEVIATION note above: r, k, prune_end are accepted and ignored) */
uint64_t astarpa_gcsh(const uint8_t *a, uintptr_t a_len, const uint8_t *b, uintptr_t b_len,
                      uintptr_t r, uintptr_t k, bool prune_end,
                      uint8_t **cigar_ptr, uintptr_t *cigar_len);

/* astarpa-c/astarpa.h:65, src/lib.rs:99-101 */
void astarpa_free_cigar(uint8_t *cigar);

#ifdef __cplusplus
}
#endif
#endif
