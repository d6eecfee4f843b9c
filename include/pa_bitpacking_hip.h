/*
 * pa_bitpacking_hip.h -- operator-level C ABI of libastarpa_c_hip.so (MI355X / gfx950).
 *
 * These are the entry points a Rust shim crate binds to replace the `pa_bitpacking` operators the
 * A*PA2 block engine calls (see INTEGRATION.md for the `extern "C"` block):
 *   pa_bp_profile_build  <- pa_bitpacking::BitProfile::build      pa-bitpacking/src/profile.rs:112-133
 *   pa_bp_compute        <- pa_bitpacking::simd::compute::<2,H,4> pa-bitpacking/src/simd.rs:98-226
 *                           (call sites astarpa2/src/blocks.rs:719-724)
 *   pa_bp_fill           <- pa_bitpacking::simd::fill::<2,H,4>    pa-bitpacking/src/simd.rs:326-437
 *                           (call site  astarpa2/src/blocks.rs:631)
 * plus batched, device-resident forms with no reference counterpart (the reference aligns pairs one
 * after another, pa-bin/src/main.rs:24-35) that keep many independent pairs in flight on one GPU.
 *
 * Layouts (plain pointers, host memory unless stated):
 *   Bits / V / H = two u64 each, exactly the reference's (u64,u64) tuples once made repr(C):
 *     a2[2*i+{0,1}]   BitProfile char of a: (-(r&1), -((r>>1)&1)), r = rank in "ACGT"
 *     b2[2*j+{0,1}]   negated bit-planes of 64 rows of b; rows >= |b| are (0,0)
 *     v2[2*j+{0,1}]   V(p,m): bit k of p/m <=> D[64j+k+1]-D[64j+k] = +1/-1
 *     h2[2*i+{0,1}]   H=(p,m), each 0 or 1
 * Errors: functions returning int return 0 on success, <0 on error (PA_E_*); pa_last_error()
 * gives a message.  Cost-returning operators return INT32_MIN on error.
 */
#ifndef PA_BITPACKING_HIP_H
#define PA_BITPACKING_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_E_INVALID_BASE (-1) /* a character outside "ACGT" (the reference panics, profile.rs:113-126) */
#define PA_E_HIP (-2)          /* HIP runtime failure / no GPU */
#define PA_E_TIMEOUT (-3)      /* device-side bounded spin expired */
#define PA_E_ARG (-4)
#define PA_E_INTERNAL (-5)     /* engine invariant violated (where the reference would panic) */
#define PA_E_NOMEM (-6)        /* host allocation failed; no output pointer is left owned by the caller */

const char* pa_last_error(void);

/* Number of visible GPUs (0 if none / HIP unavailable). */
int pa_device_count(void);
/* Select the GPU used by this thread's subsequent calls (one process per GPU: call once with LOCAL_RANK). */
int pa_set_device(int device);

/* BitProfile::build.  a2 has 2*n u64, b2 has 2*ceil(m/64) u64.  Runs on the GPU. */
int pa_bp_profile_build(const uint8_t* a, size_t n, const uint8_t* b, size_t m, uint64_t* a2, uint64_t* b2);

/* simd::compute: rectangle a[0..n) x b[0..w) words.  Updates h2 (top -> bottom deltas) and v2
 * (left -> right deltas) in place and returns the sum of the bottom deltas.
 * The return value and v2 are exact in both modes.  With exact_end != 0, h2 receives the exact bottom
 * row; with exact_end == 0 the reference leaves h unspecified (padded tail, simd.rs:184-225) and this
 * library leaves h2 untouched (the callers -- HMode::None / Input, blocks.rs:730-741 -- discard it). */
int32_t pa_bp_compute(const uint64_t* a2, size_t n, const uint64_t* b2, size_t w, uint64_t* h2, uint64_t* v2,
                      int exact_end);

/* simd::fill: as compute (exact), additionally values[(i*w + j)*2 + {0,1}] = V of word j after column i. */
int32_t pa_bp_fill(const uint64_t* a2, size_t n, const uint64_t* b2, size_t w, uint64_t* h2, uint64_t* v2,
                   uint64_t* values);

/* ---- device-resident operator handles ---------------------------------------------------------------------------------
 * For a host engine that calls the operators block by block (astarpa2/src/blocks.rs:112 BitProfile::build once,
 * :719-724 compute per block range, :631 fill during traceback): the sequences, the profile and the persistent row of
 * horizontal deltas (`Blocks::h`, blocks.rs:103-105) stay on the GPU; a call uploads and downloads only the `v` words of its
 * rectangle -- columns [i0, i1) of a, 64-row words [w0, w1) of b.
 * h_mode is blocks.rs:665-671: 0 None (top row +1, bottom row dropped), 1 Input (top row from the stored h), 2 Update (stored
 * h in, bottom row stored back), 3 Output (top row +1, bottom row stored).  *sum_out = sum of the bottom-row deltas. */
typedef struct pa_bp_ctx pa_bp_ctx;
pa_bp_ctx* pa_bp_ctx_create(const uint8_t* a, size_t n, const uint8_t* b, size_t m); /* ASCII "ACGT"; NULL on error */
int pa_bp_ctx_compute(pa_bp_ctx* ctx, int32_t i0, int32_t i1, size_t w0, size_t w1, uint64_t* v, int h_mode, int32_t* sum_out);
/* fill (blocks.rs:627-648): values[((i - i0) * (w1 - w0) + (j - w0)) * 2 + {0,1}] = V of word j after column i; h_bottom[i - i0]
 * (optional) = bottom-row delta of column i in {-1, 0, +1}.  The stored h row is not touched. */
int pa_bp_ctx_fill(pa_bp_ctx* ctx, int32_t i0, int32_t i1, size_t w0, size_t w1, uint64_t* v, uint64_t* values, int8_t* h_bottom);
void pa_bp_ctx_destroy(pa_bp_ctx* ctx);

/* pa_bitpacking::search(pattern, text, unmatched_cost).out (pa-bitpacking/src/search.rs:46-120; pa_python/src/lib.rs:4-7):
 * semi-global search of a short pattern (may contain N, n or an asterisk = any base, Y/y = C or T, R/r = A or G) in a text (actgACTG).
 * out[|pattern| + |text| + 1] = costs along the bottom row, then up the right column.  Runs the ScatterProfile
 * variant of the strip kernel (simd/scatter_profile.rs). */
int pa_search(const uint8_t* pattern, size_t plen, const uint8_t* text, size_t tlen, float unmatched_cost, int32_t* out);
/* SearchResult::trace(idx) (search.rs:104-228): the alignment that ends at output index idx of pa_search (bottom row left
 * to right, then the right column upwards).  The sub-rectangle text[end - width .. end) x pattern is re-filled on the GPU
 * (FILL variant of the scatter-profile strip kernel, width = 2|pattern| doubling) and walked back on the host in the
 * reference's order: matches, then 'D' (one text character), 'I' (one pattern character), 'X'.
 * *cigar_out: malloc'ed "=I4=X=" string; *path_out: malloc'ed (text index, pattern index) pairs from the start of the
 * alignment to its end, *npos_out of them.  Release both with free(). */
int pa_search_trace(const uint8_t* pattern, size_t plen, const uint8_t* text, size_t tlen, float unmatched_cost, size_t idx,
                    char** cigar_out, int32_t** path_out, size_t* npos_out);

/* ---- batched full-DP (cost only) on device-resident pairs ------------------------------------------ */
/* What `AstarPa2Params::nw().make_aligner(false).cost(a,b)` computes (astarpa2/src/params.rs:46-68,
 * blocks.rs:252-277) for many independent pairs at once. */
typedef struct pa_batch pa_batch;

/* Upload `pairs` sequence pairs (ASCII "ACGT") and plan their strips.  Returns NULL on error. */
pa_batch* pa_batch_create(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b,
                          const size_t* b_len, size_t pairs);
/* Banded variant (cost only): every pair is computed inside the diagonal band |i - j| + |(n - m) - (i - j)| <= t (the
 * reference's GapGap domain, astarpa2/src/domain.rs:97-116) with t = | |a| - |b| | + divergence_hint * max(|a|, |b|) + 32.
 * A pair whose cost comes out above its t only has an upper bound and is re-run with a wider band inside pa_batch_run
 * until it fits, so the results are exact for any hint; a good hint (the expected edit rate) just avoids the re-runs.
 * For 5 % divergent 100 kbp pairs the band holds about 1/7 of the matrix. */
pa_batch* pa_batch_create_banded(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b,
                                 const size_t* b_len, size_t pairs, float divergence_hint);
/* One pass: build profiles on the GPU, run every strip, read the costs back.  cost_out[pairs].
 * kernel_ms (optional) receives the duration of the strip kernel alone, measured with HIP events on
 * the launch stream. */
int pa_batch_run(pa_batch* plan, int32_t* cost_out, float* kernel_ms);
/* Totals for reporting: DP cells (sum n*m), word updates (64 cells each), strips, algorithmic HBM bytes. */
void pa_batch_stats(const pa_batch* plan, double* cells, double* word_updates, double* strips, double* algo_bytes);
/* How the batch was laid out on the GPU (reporting only): k = 32-row subwords per lane of the tall strips (1, 2, 4, 8),
 * sequential = 1 when one wavefront runs a whole pair strip after strip (pair_kernel), 0 for chained strips
 * (strip_kernel); valu_instructions = wavefront VALU instructions one pass executes: (11 + 12 k) per strip step,
 * (10 + 10 k) for k >= 4 (eq words read from LDS). */
void pa_batch_shape(const pa_batch* plan, int* k, int* sequential, double* valu_instructions);
/* Big cost-only batches (pa_batch_create with 64 pairs or more, when its estimate is the lower one; PA_SLICE=0 never, PA_SLICE=1 always,
 * PA_SLICE=<rows> forces the rows per lane) run BIT-SLICED (csrc/slice_kernel.hpp): the pairs are sorted by length and cut into groups of
 * 32; bit p of every register belongs to pair p of the group and a register is one DP row, which turns the add and the shifts of the Myers
 * step (pa-bitpacking/src/myers.rs:27-55) into eight boolean instructions per row -- the same distances, 1.5x the cells per second.
 * Returns the rows per lane of that kernel (52 down to 28), 0 when the batch runs on the strip kernels (pa_batch_shape then says which);
 * groups, (group, strip) jobs, cells computed including the padding to whole strips and to the group's longest sequences, device bytes
 * of the plan, and how many of them (the boundary rows between strips) are reset before every pass. */
int pa_batch_slice_info(const pa_batch* plan, double* groups, double* jobs, double* computed_cells, double* device_bytes, double* boundary_bytes);
void pa_batch_destroy(pa_batch* plan);

/* ---- batched global alignment WITH traceback ------------------------------------------------------- */
/* Cost and CIGAR of every pair, both computed on the GPU: the forward pass keeps the right-edge column of every
 * 256-column block (the reference's sparse blocks, astarpa2/src/blocks.rs:322-339) and one wavefront per pair walks
 * back through them like Blocks::trace without DT-trace (blocks/trace.rs:21-228: re-fill of a 5/4-width-high
 * sub-block, doubling; greedy matches, then insertion / deletion / substitution).  The result is what
 * `pa_align(.., pa_params_batch_align(), trace = 1, ..)` returns for each pair, i.e. AstarPa2Params::nw() with
 * front.sparse = true.  cigar_out[i] is a malloc'ed "=I4=X="-style string (release with astarpa_free_cigar / free).
 * A pair whose traceback needs a re-fill taller than 8192 rows (an indel of that size inside one 256-column block), or that
 * reaches a state the reference itself would panic on, is redone by the host engine transparently. */
pa_batch* pa_batch_create_trace(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b,
                                const size_t* b_len, size_t pairs);
/* The same with the traceback options of trace_params->front (struct pa_astarpa2_params, pa_astarpa2.h): with dt_trace set,
 * every block is first tried with the diagonal-transition trace (blocks/trace.rs:231-416; max_g <= 40, fr_drop) and re-filled
 * only where that gives up -- the result is what pa_align(.., params, trace = 1, ..) returns for Full-domain params with that
 * `front`, e.g. the `simple` preset's { sparse, dt_trace, max_g = 40, fr_drop = 10 }.  NULL = pa_batch_create_trace. */
struct pa_astarpa2_params;
pa_batch* pa_batch_create_trace_params(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len,
                                       size_t pairs, const struct pa_astarpa2_params* trace_params);
int pa_batch_align(pa_batch* plan, int32_t* cost_out, char** cigar_out, float* forward_ms, float* trace_ms);
/* pa_batch_align without one malloc'ed string per pair: text_out[i] points at text_len_out[i] characters of pair i's CIGAR -- NOT
 * NUL-terminated -- in host memory the plan owns, valid until the next alignment call on this plan or pa_batch_destroy; nothing to free.
 * For callers that copy the text into objects of their own anyway (a language binding, the writer of pa-bin's CSV). */
int pa_batch_align_view(pa_batch* plan, int32_t* cost_out, const char** text_out, uint32_t* text_len_out, float* forward_ms, float* trace_ms);
/* Releases cigars[0 .. n) of a batch result in one call (entries may be NULL; they are set to NULL). */
void pa_free_cigars(char** cigars, size_t n);
/* Pairs (summed over all pa_batch_align calls of this plan) whose traceback was redone by the host engine. */
size_t pa_batch_trace_fallbacks(const pa_batch* plan);
/* Batched A*PA2: the block-column store of a pair is a window of words around the main diagonal per 256-column block (the reference keeps the
 * block's rows only: astarpa2/src/block.rs:8-21), sized from the lengths (PA_APA2_WINDOW=<words> overrides, 0 = full columns).  Pairs
 * (summed over all pa_batch_align calls of this plan) whose band left the window and that were aligned again with full-height columns. */
size_t pa_batch_window_retries(const pa_batch* plan);
/* Those pairs are aligned again in sub-batches, one at a time, whose full-height stores -- (|a| / 256 + 2) x ceil(|b| / 64) x 16 bytes per
 * pair: 9.8 MB for 100 kbp, 1 GB for 1 Mbp -- stay below 24 GB each (a single pair may exceed it; PA_WINDOW_RETRY_BYTES overrides the
 * bound).  The largest store such a sub-batch of this plan held, in bytes (0: no pair ever left its window). */
double pa_batch_window_retry_bytes(const pa_batch* plan);

/* ---- batched A*PA2 (band-limited alignment of many pairs) ------------------------------------------------------------ */
/* What a loop over pa_align(a, b, params, trace = 1, ..) returns -- cost, CIGAR and statistics of AstarPa2Params::simple(),
 * AstarPa2Params::full() and their relatives (astarpa2/src/params.rs:70-128; the loop of pa-bin/src/main.rs:24-35) -- for many pairs at once: ONE
 * WAVEFRONT runs a pair's whole band search on the GPU (every align_for_bounded_dist pass of domain.rs:356-541, the doubling
 * of band.rs:100-141 included, no host round trip per block, pass or pair), a second kernel walks the blocks of the
 * successful pass back (Blocks::trace with DT-trace, blocks/trace.rs:21-416).  Only the band is computed, not the matrix.
 * Supported parameters: Domain::Astar with NoCost / GapCost / SH / GCSH (exact matches, local pruning p), block_width 256,
 * front.sparse, with or without incremental doubling and pruning of matches, BandDoubling or LinearSearch (NULL otherwise: use
 * pa_align).  GCSH, pruning and incremental doubling -- the `full` preset -- run in a second kernel (csrc/apa2_full_kernel.hpp) that
 * keeps the heuristic on the GPU: its matches are found by a kernel of their own, ONCE, when the batch is created (seeds, exact k-mer
 * matches, transform filter, local pruning: csrc/gcsh_build_kernel.hpp; PA_GCSH_HOST_BUILD=1 = on host threads instead); every
 * pa_batch_align of the batch starts from them again (the flags of pruned matches are reset, the matches are not searched again), the
 * contours are derived and probed by the pair's wavefront (pa-heuristic csh.rs:341-376,
 * hint_contours.rs:213-272), the matches of a block are pruned by one lane per seed (prune.rs:245-292), the stored row of
 * incremental doubling (blocks.rs:342-469) is tapped out of the block's single strip.  Results come from pa_batch_align(); a pair the
 * kernels hand back (an empty sequence, a re-fill taller than 8192 rows, a state the reference would panic on) is redone by
 * pa_align's engine transparently.  pa_batch_pair_stats: the statistics of every pair of the last pa_batch_align (timers 0).
 * pa_batch_run() on such a batch (or pa_batch_align with cigar_out == NULL) runs the band search without the traceback: the
 * costs alone -- the distances, over the band of the traced mode as pa_align(trace = 0) returns them (pa_astarpa2.h).
 * A batch of one or two pairs of >= 32 768 bases each is aligned by pa_align's engine instead (many wavefronts per pass: 14 ms
 * against 47 ms for a 100 kbp pair; same results, forward_ms then covers the whole alignment); PA_BATCH_SMALL_ROUTE=0 disables. */
struct pa_astarpa2_stats;
pa_batch* pa_batch_create_params(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len,
                                 size_t pairs, const struct pa_astarpa2_params* params);
int pa_batch_pair_stats(const pa_batch* plan, struct pa_astarpa2_stats* stats_out);
int pa_batch_params_supported(const struct pa_astarpa2_params* params); /* 1: pa_batch_create_params takes them; 0: use pa_align */
/* Reporting for batches of the `full` family: ms spent on the heuristic's matches when the batch was created (host threads: positive; the GPU's
 * build kernel: NEGATIVE), their number, and (PA_APA2_PROBE_STATS
 * set) the h probes of the last forward pass, the 64-layer load rounds they took, and phase_wave_ms[0..7): wavefront-milliseconds (summed over
 * wavefronts) deriving contours, in DP strips, in h probes, in Block::index, in prune_block, initialising columns, in total. */
void pa_batch_full_info(const pa_batch* plan, double* build_ms, double* matches, double* probes, double* rounds, double* phase_wave_ms);
/* Diagnostics of the batched A*PA2 kernels' rendezvous (round 5: two blocks of at most 16 words from two pairs run as ONE wavefront strip,
 * csrc/strip2_kernel.hpp; PA_APA2_RDV=0 turns it off, PA_APA2_RDV_PATIENCE_US sets how long a block waits for a partner): for the last
 * forward pass out4[0] = strips that ran fused with another pair's, [1] = strips a partner ran, [2] = strips that ran alone, [3] = of
 * those, strips that had waited for a partner first.  Results never depend on any of it. */
int pa_batch_rdv_stats(const pa_batch* plan, uint64_t* out4);
/* Diagnostics / tests: the matches of GCSH (seeds, exact k-mer matches, transform filter, local pruning p_local <= 14) of one pair as the GPU
 * finds them (csrc/gcsh_build_kernel.hpp), by start: out_ij[2 t] = column, out_ij[2 t + 1] = row.  Returns their number, < 0 on error. */
long pa_debug_gcsh_matches(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, int32_t k, int32_t p_local, int32_t* out_ij, size_t cap_out);
/* Diagnostics / tests: GCSH (seed length k, local pruning p_local) of one pair AS THE GPU COMPUTES IT -- the contours derived and probed
 * by one wavefront -- at nq positions (queries[2 t] = i, queries[2 t + 1] = j): out[t] = h(i, j), out[nq] = number of contour layers. */
int pa_debug_gcsh_probe(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, int32_t k, int32_t p_local, const int32_t* queries,
                        size_t nq, int32_t* out);

/* Many-pair mode over several GPUs from ONE process (independent pairs shard with no data-path exchange; the
 * reference runs them one after another, pa-bin/src/main.rs:24-35): a WORK QUEUE.  The pairs are sorted by estimated work
 * (heaviest first) and cut into chunks; one host thread per entry of devices[0..ndevices) binds its device and pulls chunk after
 * chunk from one atomic counter, each chunk one pa_batch_align (the cost-only batch when cigar_out is NULL); results land at the
 * pairs' original indices.  The balance is dynamic: nobody has to know beforehand how much work a pair is.  A device may be listed
 * twice (two chunks in flight on one GPU).  One process per GPU with torch.distributed (sharding.py) is the other recipe.
 * _params: the batched A*PA2 of pa_batch_create_params for every chunk; stats_out (optional) receives every pair's statistics. */
int pa_batch_align_multi(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len, size_t pairs,
                         const int* devices, int ndevices, int32_t* cost_out, char** cigar_out);
int pa_batch_align_multi_params(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len, size_t pairs,
                                const int* devices, int ndevices, const struct pa_astarpa2_params* params, int32_t* cost_out,
                                char** cigar_out, struct pa_astarpa2_stats* stats_out);

/* ---- pa-bin's data formats (pa-bin/src/lib.rs:67-114, pa-bin/src/main.rs:24-35) ------------------------- */
/* Input: `.seq` (line pairs, '>' then '<' markers), `.txt` (plain line pairs), `.fna`/`.fa`/`.fasta` (records taken two at
 * a time), or a directory of such files.  Output: one line "{cost},{cigar}" per pair.  Host code only. */
typedef struct pa_pairs pa_pairs;
pa_pairs* pa_pairs_read(const char* path); /* NULL on error (pa_last_error) */
size_t pa_pairs_count(const pa_pairs* pairs);
int pa_pairs_get(const pa_pairs* pairs, size_t i, const uint8_t** a, size_t* a_len, const uint8_t** b, size_t* b_len);
void pa_pairs_free(pa_pairs* pairs);
int pa_write_results_csv(const char* path, const int32_t* costs, const char* const* cigars, size_t n);
/* pa-bin's main loop for a whole input at once: read, pa_batch_align every pair on the GPU, write the CSV. */
int pa_align_file(const char* input_path, const char* output_path, size_t* pairs_out);
/* The same with an aligner's parameters (pa-bin's `--aligner astarpa2 ...`): parameters of the batched A*PA2 family
 * (pa_batch_create_params) run as batches on the current device, others as a loop over pa_align; NULL = pa_align_file. */
struct pa_astarpa2_params;
int pa_align_file_params(const char* input_path, const char* output_path, const struct pa_astarpa2_params* params, size_t* pairs_out);

#ifdef __cplusplus
}
#endif
#endif
