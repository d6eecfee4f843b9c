// pa_hip_internal.hpp -- shared declarations of the host side of libastarpa_c_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pa_astarpa2.h"
#include "../../include/pa_bitpacking_hip.h"
#include "strip_kernel.hpp"

namespace pa {

constexpr int kWordsPerStrip = 32;  // per subword-per-lane: a strip of k subwords/lane covers 32*k reference words (2048*k rows)

void set_error(const char* fmt, ...);
bool hip_ok(hipError_t e, const char* what);
bool ensure_device();

void release_alloc_cache();  // the cached device blocks back to the driver (pa_release_pools)
void release_scope_begin();  // one device wait now; DeviceBuf::release calls of this thread skip theirs until release_scope_end()
void release_scope_end();
void release_scope_begin_waited();  // the caller has waited for everything that used its buffers (its own streams)

struct DeviceBuf {
    void* ptr = nullptr;
    size_t size = 0;
    int device = 0;  // the device ptr lives on
    DeviceBuf() = default;
    DeviceBuf(const DeviceBuf&) = delete;
    DeviceBuf& operator=(const DeviceBuf&) = delete;
    ~DeviceBuf() { release(); }
    bool alloc(size_t bytes);
    // grow-only: keeps the buffer when it is large enough; *grew = true when a new (uninitialised) buffer was allocated
    bool reserve(size_t bytes, bool* grew = nullptr);
    void release();
    template <class T>
    T* as() const { return reinterpret_cast<T*>(ptr); }
};

// One rectangle = words [w0,w1) x n columns of one pair, split into chained strips.
struct RectPlan {
    const uint32_t* a_codes = nullptr;  // device, packed codes of the whole sequence
    int col0 = 0;                       // absolute first column of the rectangle
    const uint32_t* b_prof = nullptr;   // device, u32 view of the pair's profile (word 0)
    uint32_t* v = nullptr;              // device, u32 view of the v column (word 0 of the same indexing as b_prof)
    int n = 0, w0 = 0, w1 = 0;
    const uint8_t* hin_arr = nullptr;   // per-absolute-column top deltas or nullptr (+1)
    uint8_t* hout_arr = nullptr;        // per-absolute-column bottom deltas out or nullptr
    uint64_t* gran = nullptr;           // (S-1) * gran_stride granules, zeroed before launch
    size_t gran_stride = 0;             // >= ceil(n/16)
    int32_t* sum_out = nullptr;
    bool exact_end = false;
    bool v_init_one = false;
    int tail_rows = -1;  // see StripJob::tail_rows
    uint32_t* values = nullptr;  // fill mode
    int fill_stride = 0, fill_word0 = 0;
    bool pingpong = false;  // sequential-pairs mode: two granule rows per rectangle, strip s writes row s&1;
                            // the ragged bottom is planned as short k = 1 strips (strip_plan)
    int k = 1;  // 32-row subwords per lane (1: lowest latency; 2, 4: fewer instructions per cell, cost-only strips)
    uint32_t* ckpt = nullptr;  // traced batches: V column after every 256th column (StripJob::ckpt)
    int ckpt_stride = 0;
};

void plan_rect(std::vector<StripJob>& jobs, const RectPlan& r);
// Strips of a rectangle of w words: `full` strips of 32*k words, then `tail1` strips of up to 32 words (sequential mode only).
struct StripPlan {
    int full = 0, tail1 = 0;
    int strips() const { return full + tail1; }
};
StripPlan strip_plan(int w, int k, bool sequential);
size_t rect_granules(int n, int w, int k = 1, bool pingpong = false);
bool launch_strips(const StripJob* d_jobs, int njobs, bool fill, uint32_t* d_ticket_err, hipStream_t s, bool zero_ticket = true,
                   bool scatter = false, int k = 1, int block_waves = kStripBlockWaves, bool ckpt = false);
bool launch_pairs(const StripJob* d_jobs, const int32_t* d_first, int npairs, uint32_t* d_ticket_err, hipStream_t s, int k, bool ckpt = false);
int align_hip(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, const pa_astarpa2_params& params, bool trace, bool self_check,
              int32_t* cost_out, std::string* cigar_out, pa_astarpa2_stats* stats_out);
bool encode_a_device(const uint8_t* d_a, int n, uint32_t* d_codes, uint32_t* d_bad, hipStream_t s);
bool build_b_device(const uint8_t* d_b, int m, uint64_t* d_prof, uint32_t* d_bad, hipStream_t s);
// both in one launch; all `code_words` words of d_codes are written (zero beyond the sequence); `bad` may be host-mapped memory
bool encode_pair_device(const uint8_t* d_a, int n, uint32_t* d_codes, int code_words, const uint8_t* d_b, int m, uint64_t* d_prof, uint32_t* bad,
                        hipStream_t s);

}  // namespace pa
