/*
 * oracle/pa_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's bit-parallel edit-distance hot path
 * (RagnarGrootKoerkamp/astar-pairwise-aligner, crate `pa-bitpacking`), written from reading the
 * Rust sources; every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this
 * directory, and only as the checker / the reported CPU baseline.  The shipped library
 * (astar-pairwise-aligner_amd/csrc -> libastarpa_c_hip.so) never links or loads anything here.
 *
 * Parity pinning: the reference cannot be built in this image (no Rust toolchain; un-vendored
 * git deps), so the restatement is pinned against the reference's own known answers:
 *   - search("AC","CTTACTTA",0.0)  == [0,0,1,2,1,0,1,2,1,0,0]   pa-bitpacking/src/search.rs:30-31
 *   - search("CT","ACTG",1.0)      == [2,2,1,0,1,2,2]           pa_python/readme.md:13-16
 *   - edit distance 2 for ("ACTCGCT","AACTCGTT")                 astarpa-c/example.c:8-29
 *   - rule: h=+1,v=+1 => every schedule returns lev(a,b)-|b|     pa-bitpacking/benches/nw/main.rs:145-149
 * Exact A*PA2 CIGAR strings are "parity unpinned" (the reference tests never compare them).
 */
#ifndef PA_ORACLE_H
#define PA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* V(p,m): 64 vertical deltas, bit j of p/m <=> D[j+1]-D[j] = +1/-1.   encoding.rs:5-6 */
typedef struct { uint64_t p, m; } pa_v_t;
/* H = (u64,u64): one horizontal delta, p/m each 0 or 1.                encoding.rs:141-169 */
typedef struct { uint64_t p, m; } pa_h_t;
/* Bits(b0,b1): BitProfile character.                                   profile.rs:90-92 */
typedef struct { uint64_t b0, b1; } pa_bits_t;

/* encoding.rs:21-38 */
int32_t pa_or_v_value(pa_v_t v);
int32_t pa_or_v_value_of_prefix(pa_v_t v, int32_t j); /* 0 <= j < 64 */
int32_t pa_or_v_value_of_suffix(pa_v_t v, int32_t j); /* 0 <  j <= 64 */

/* BitProfile::build, profile.rs:112-133.  pa has n entries, pb has ceil(m/64) entries.
 * Returns 0, or -1 when a character is not one of "ACGT" (the reference panics there). */
int pa_or_bitprofile_build(const uint8_t* a, size_t n, const uint8_t* b, size_t m,
                           pa_bits_t* pa, pa_bits_t* pb);

/* myers::compute_block, myers.rs:27-55 (BitProfile eq, profile.rs:141-144). */
void pa_or_compute_block(pa_h_t* h0, pa_v_t* v, pa_bits_t ca, pa_bits_t cb);

/* scalar::row scalar.rs:37-46; scalar::col scalar.rs:9-18.  Return sum of bottom h. */
int32_t pa_or_scalar_row(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                         pa_h_t* h, pa_v_t* v);
int32_t pa_or_scalar_col(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                         pa_h_t* h, pa_v_t* v);
/* scalar::fill scalar.rs:405-425.  values is n x w row-major: values[i*w + j] = v[j] after column i. */
int32_t pa_or_scalar_fill(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                          pa_h_t* h, pa_v_t* v, pa_v_t* values);

/* simd::compute::<2,(u64,u64),4>, simd.rs:98-226, including its dispatch on small n / w==1 and
 * the non-exact tail (pad rows Bits(0,0), V(0,0), subtract their right-edge sum; simd.rs:184-225).
 * DP values are schedule independent, so this is computed with the scalar row schedule on the
 * same (padded) rectangle; `h` out therefore equals the reference's `h` out in BOTH modes
 * (in non-exact mode that is the bottom of the padded rows, i.e. what the reference leaves there).
 * `ilp_n` is the reference's N (2, or 1 for `no_ilp`). */
int32_t pa_or_simd_compute(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                           pa_h_t* h, pa_v_t* v, int exact_end, int ilp_n);
/* Number of pad rows simd::compute::<N,_,4> appends for (n, w, exact_end).  simd.rs:112-126,184-218 */
size_t pa_or_simd_pad_rows(size_t n, size_t w, int exact_end, int ilp_n);

/* simd::fill::<2,H,4>, simd.rs:326-437 (exact mode only). */
int32_t pa_or_simd_fill(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                        pa_h_t* h, pa_v_t* v, pa_v_t* values);

/* The real 8-row anti-diagonal strip schedule (simd.rs:228-315) with AVX2, used only as the timed
 * CPU baseline ("port" of compute::<2,(u64,u64),4>).  Same contract as pa_or_simd_compute. */
int32_t pa_or_strip_compute_avx2(const pa_bits_t* a, size_t n, const pa_bits_t* b, size_t w,
                                 pa_h_t* h, pa_v_t* v, int exact_end);

/* Full n x m cost-only DP the way AstarPa2Params::nw() cost mode drives the operator:
 * one call per 256 columns over all words, h=+1 fresh, v carried (blocks.rs:252-277,730-734).
 * use_avx2 selects the strip schedule; returns the edit distance. */
int32_t pa_or_nw_cost(const uint8_t* a, size_t n, const uint8_t* b, size_t m, int use_avx2);

/* ScatterProfile + semi-global search, profile.rs:25-75, search.rs:46-120.
 * out must hold |pattern|+|text|+1 costs.  Returns 0 or -1 on an unknown base. */
int pa_or_search(const uint8_t* pattern, size_t plen, const uint8_t* text, size_t tlen,
                 float unmatched_cost, int32_t* out);

/* SearchResult::trace(idx), search.rs:104-228: the alignment ending at output index idx (bottom row left to right, then the
 * right column upwards).  cigar_buf receives the "=I4=X=" string ('I' consumes a pattern row, 'D' a text column), path_buf
 * the visited (text index, pattern index) positions from start to end.  0, or -1 where the reference would panic. */
int pa_or_search_trace(const uint8_t* pattern, size_t plen, const uint8_t* text, size_t tlen, float unmatched_cost,
                       size_t idx, char* cigar_buf, size_t cigar_cap, int32_t* path_buf, size_t path_cap, size_t* npos_out);

/* Plain O(nm) unit-cost Levenshtein (stands in for triple_accel::levenshtein_exp, pa-test/src/lib.rs:76). */
int32_t pa_or_levenshtein(const uint8_t* a, size_t n, const uint8_t* b, size_t m);

/* Cigar::verify at unit cost (pa-test/src/lib.rs:98): parse "=I4=X=" style strings
 * (count omitted when 1; '=' match, 'X' sub, 'I' advances b, 'D' advances a),
 * walk a/b, and return the cost, or -1 if the string is malformed / a '=' is not a match /
 * an 'X' is not a mismatch / it does not end at (n,m). */
int32_t pa_or_cigar_verify(const char* cigar, const uint8_t* a, size_t n, const uint8_t* b, size_t m);

#ifdef __cplusplus
}
#endif
#endif
