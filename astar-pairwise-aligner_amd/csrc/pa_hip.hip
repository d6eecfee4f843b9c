// pa_hip.hip -- host side of libastarpa_c_hip.so: device buffers, strip planning, the operator
// C ABI (include/pa_bitpacking_hip.h) and the batched full-DP plan.  gfx950 only.
#include "pa_hip_internal.hpp"
#include "slice_plan.hpp"
#include "engine_capi.hpp"
#include "trace_kernel.hpp"
#include "apa2_units.hpp"

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <atomic>
#include <mutex>
#include <set>
#include <thread>
#include <string>
#include <vector>

namespace pa {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    set_error("HIP error in %s: %s", what, hipGetErrorString(e));
    return false;
}

// ---- device kernels: profile building ---------------------------------------------------------

// rank in "ACGT" (bio RankTransform as used by BitProfile::build, profile.rs:113); -1 otherwise
__device__ __forceinline__ int rank_acgt(uint8_t c) {
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}

// One thread per 16 columns: ASCII -> packed 2-bit codes.
__global__ void encode_a_kernel(const uint8_t* __restrict__ a, int n, uint32_t* __restrict__ codes, int nwords,
                                uint32_t* __restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    uint32_t w = 0;
    bool invalid = false;
    for (int k = 0; k < 16; ++k) {
        const int c = i * 16 + k;
        if (c < n) {
            const int r = rank_acgt(a[c]);
            invalid |= r < 0;
            w |= (uint32_t)(r & 3) << (2 * k);
        }
    }
    codes[i] = w;
    if (invalid) atomicOr(bad, 1u);
}

// One wave per 64-row word: negated bit-planes via ballot; rows >= m stay (0,0) (profile.rs:127-132).
__global__ void build_b_kernel(const uint8_t* __restrict__ b, int m, uint64_t* __restrict__ prof, int nwords,
                               uint32_t* __restrict__ bad) {
    const int word = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (word >= nwords) return;
    const int lane = threadIdx.x & 63;
    const int j = word * 64 + lane;
    int r = 3;  // (r&1)^1 == 0 and ((r>>1)&1)^1 == 0 => pad rows contribute 0 bits
    bool invalid = false;
    if (j < m) {
        r = rank_acgt(b[j]);
        invalid = r < 0;
        r &= 3;
    }
    const uint64_t nb0 = __ballot(((r & 1) ^ 1) != 0);
    const uint64_t nb1 = __ballot((((r >> 1) & 1) ^ 1) != 0);
    if (lane == 0) {
        prof[2 * word] = nb0;
        prof[2 * word + 1] = nb1;
    }
    if (invalid) atomicOr(bad, 1u);
}

// Batched forms (one launch for all pairs of a pa_batch): blockIdx.y = pair.
struct PairDesc {
    unsigned long long a_off, b_off, code_off, prof_off;  // element offsets into the concatenated buffers
    int n, m;
};

__global__ void encode_a_batch_kernel(const uint8_t* __restrict__ a_cat, uint32_t* __restrict__ codes_cat,
                                      const PairDesc* __restrict__ desc, uint32_t* __restrict__ bad) {
    const PairDesc d = desc[blockIdx.y];
    const int nwords = (d.n + 15) / 16;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    const uint8_t* a = a_cat + d.a_off;
    uint32_t w = 0;
    bool invalid = false;
    for (int k = 0; k < 16; ++k) {
        const int c = i * 16 + k;
        if (c < d.n) {
            const int r = rank_acgt(a[c]);
            invalid |= r < 0;
            w |= (uint32_t)(r & 3) << (2 * k);
        }
    }
    codes_cat[d.code_off + i] = w;
    if (invalid) atomicOr(bad, 1u);
}

__global__ void build_b_batch_kernel(const uint8_t* __restrict__ b_cat, uint64_t* __restrict__ prof_cat,
                                     const PairDesc* __restrict__ desc, uint32_t* __restrict__ bad) {
    const PairDesc d = desc[blockIdx.y];
    const int nwords = (d.m + 63) / 64;
    const int word = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (word >= nwords) return;
    const uint8_t* b = b_cat + d.b_off;
    const int lane = threadIdx.x & 63;
    const int j = word * 64 + lane;
    int r = 3;
    bool invalid = false;
    if (j < d.m) {
        r = rank_acgt(b[j]);
        invalid = r < 0;
        r &= 3;
    }
    const uint64_t nb0 = __ballot(((r & 1) ^ 1) != 0);
    const uint64_t nb1 = __ballot((((r >> 1) & 1) ^ 1) != 0);
    if (lane == 0) {
        prof_cat[2 * (d.prof_off + word)] = nb0;
        prof_cat[2 * (d.prof_off + word) + 1] = nb1;
    }
    if (invalid) atomicOr(bad, 1u);
}

template __global__ void strip_kernel<1, false, false>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void strip_kernel<1, true, false>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void strip_kernel<1, false, true>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void strip_kernel<1, true, true>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void strip_kernel<2, false, false>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void strip_kernel<4, false, false>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void strip_kernel<8, false, false>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void strip_kernel<1, false, false, true>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void strip_kernel<2, false, false, true>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void strip_kernel<4, false, false, true>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void strip_kernel<8, false, false, true>(const StripJob*, int, uint32_t*, uint32_t*);
template __global__ void pair_kernel<1>(const StripJob*, const int32_t*, int, uint32_t*);
template __global__ void pair_kernel<2>(const StripJob*, const int32_t*, int, uint32_t*);
template __global__ void pair_kernel<4>(const StripJob*, const int32_t*, int, uint32_t*);
template __global__ void pair_kernel<8>(const StripJob*, const int32_t*, int, uint32_t*);
template __global__ void pair_kernel<1, true>(const StripJob*, const int32_t*, int, uint32_t*);
template __global__ void pair_kernel<2, true>(const StripJob*, const int32_t*, int, uint32_t*);
template __global__ void pair_kernel<4, true>(const StripJob*, const int32_t*, int, uint32_t*);
template __global__ void pair_kernel<8, true>(const StripJob*, const int32_t*, int, uint32_t*);

// ---- device context -----------------------------------------------------------------------------

static thread_local int g_device_props_cus = 0;  // of the device this thread last initialised (pa_set_device is per thread)
static thread_local int g_device_props_dev = -1;

bool ensure_device() {
    static thread_local bool inited = false;
    if (inited) {
        int cur = 0;
        if (hipGetDevice(&cur) == hipSuccess && cur == g_device_props_dev) return true;
    }
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt == 0) {
        set_error("no HIP device available: the MI355X path is required (there is no CPU fallback)");
        return false;
    }
    int dev = 0;
    if (!hip_ok(hipGetDevice(&dev), "hipGetDevice")) return false;
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties")) return false;
    g_device_props_cus = prop.multiProcessorCount;
    g_device_props_dev = dev;
    inited = true;
    return true;
}

// ---- device memory -----------------------------------------------------------------------------------------------------------
// Large buffers are CACHED: hipMalloc + hipFree of the 40 GB block-column store of a 4096 x 100 kbp batch cost about a second, seven
// times the alignment of the pairs it holds, and pa_align_file / the work queue create a batch per chunk.  A buffer of at least
// kCacheMin bytes goes to a free list when its owner lets go of it and is handed to the next request on the same device that it fits
// (at most a quarter larger than asked for).  Nothing in this library reads device memory it has not written, and a cached block is
// as undefined as a fresh one.  The list is bounded per device (cache_limit, oldest out first), emptied when an allocation fails, and
// returned to the driver by pa_release_pools().  PA_NO_ALLOC_CACHE=1 switches it off; PA_POISON_ALLOC=1 fills every buffer handed
// out with 0xA5 (tests: nothing may depend on fresh memory being zero).
namespace {
constexpr size_t kCacheMin = size_t(16) << 20, kCacheMaxDefault = size_t(16) << 30;
// The bound is PER DEVICE: PA_ALLOC_CACHE_MAX (bytes, or with a K / M / G suffix) if set, else half of the device's memory, at most 16 GB
// (round 4: with band-proportional block columns a 4096 x 100 kbp A*PA2 batch holds 5 GB, not 40)
// -- other users of the device in the same process (torch, RCCL) cannot make this library let go of what it caches.
size_t cache_limit(int dev) {
    static std::mutex mu;
    static std::vector<size_t> lim;
    std::lock_guard<std::mutex> lk(mu);
    if ((size_t)dev < lim.size() && lim[(size_t)dev]) return lim[(size_t)dev];
    size_t v = 0;
    if (const char* e = getenv("PA_ALLOC_CACHE_MAX")) {
        char* end = nullptr;
        double x = std::strtod(e, &end);
        if (end && (*end == 'G' || *end == 'g')) x *= double(size_t(1) << 30);
        else if (end && (*end == 'M' || *end == 'm')) x *= double(size_t(1) << 20);
        else if (end && (*end == 'K' || *end == 'k')) x *= 1024.0;
        v = x > 0 ? (size_t)x : 1;
    } else {
        size_t free_b = 0, total_b = 0;
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != dev) (void)hipSetDevice(dev);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = 0;
        if (cur != dev) (void)hipSetDevice(cur);
        v = total_b ? std::min(kCacheMaxDefault, total_b / 2) : kCacheMaxDefault;
    }
    if ((size_t)dev >= lim.size()) lim.resize((size_t)dev + 1, 0);
    lim[(size_t)dev] = v;
    return v;
}
struct CachedBlock {
    int dev;
    void* ptr;
    size_t size;
};
// (never destroyed: buffers of thread-local pools are released after the statics of this file at process exit)
std::mutex& g_cache_mu = *new std::mutex;
std::vector<CachedBlock>& g_cache = *new std::vector<CachedBlock>;  // oldest first
size_t g_cache_bytes = 0;
std::atomic<uint64_t> g_cache_hits{0}, g_cache_misses{0};

bool cache_on() {
    static const bool off = getenv("PA_NO_ALLOC_CACHE") != nullptr;
    return !off;
}
void* cache_take(int dev, size_t bytes, size_t* got) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    size_t best = g_cache.size();
    for (size_t i = 0; i < g_cache.size(); ++i)
        if (g_cache[i].dev == dev && g_cache[i].size >= bytes && g_cache[i].size <= bytes + bytes / 4 &&
            (best == g_cache.size() || g_cache[i].size < g_cache[best].size))
            best = i;
    if (best == g_cache.size()) return nullptr;
    void* p = g_cache[best].ptr;
    *got = g_cache[best].size;
    g_cache_bytes -= g_cache[best].size;
    g_cache.erase(g_cache.begin() + (long)best);
    return p;
}
// -> blocks the caller has to hipFree (outside the lock)
std::vector<CachedBlock> cache_put(int dev, void* ptr, size_t size) {
    std::vector<CachedBlock> out;
    const size_t limit = cache_limit(dev);
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_cache.push_back({dev, ptr, size});
    g_cache_bytes += size;
    size_t on_dev = 0;
    for (const CachedBlock& b : g_cache)
        if (b.dev == dev) on_dev += b.size;
    for (size_t i = 0; i < g_cache.size() && on_dev > limit;) {  // this device's oldest blocks go first
        if (g_cache[i].dev != dev) {
            ++i;
            continue;
        }
        out.push_back(g_cache[i]);
        g_cache_bytes -= g_cache[i].size;
        on_dev -= g_cache[i].size;
        g_cache.erase(g_cache.begin() + (long)i);
    }
    return out;
}
void free_blocks(const std::vector<CachedBlock>& blocks) {
    if (blocks.empty()) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const CachedBlock& b : blocks) {
        if (b.dev != cur) (void)hipSetDevice(b.dev);
        (void)hipFree(b.ptr);
        if (b.dev != cur) (void)hipSetDevice(cur);
    }
}
}  // namespace

// PA_POISON_ALLOC: on a stream of its own that does not synchronise with the null stream (persistent kernels may be in flight)
static bool poison_fill(void* ptr, size_t size) {
    static thread_local hipStream_t st = nullptr;
    if (!st && !hip_ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "poison stream")) return false;
    return hip_ok(hipMemsetAsync(ptr, 0xA5, size, st), "poison") && hip_ok(hipStreamSynchronize(st), "poison sync");
}

void pinned_release_all();
void release_alloc_cache() {
    pinned_release_all();
    std::vector<CachedBlock> all;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        all.swap(g_cache);
        g_cache_bytes = 0;
    }
    free_blocks(all);
}

extern "C" void pa_alloc_cache_stats(uint64_t* hits, uint64_t* misses, uint64_t* cached_bytes) {
    if (hits) *hits = g_cache_hits.load();
    if (misses) *misses = g_cache_misses.load();
    if (cached_bytes) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        *cached_bytes = g_cache_bytes;
    }
}

bool DeviceBuf::alloc(size_t bytes) {
    release();
    if (bytes < 64) bytes = 64;
    static const bool poison = getenv("PA_POISON_ALLOC") != nullptr;
    int dev = 0;
    if (!hip_ok(hipGetDevice(&dev), "hipGetDevice")) return false;
    // Small buffers are cached too (round 5), in size classes (2^k and 1.5 x 2^k): a batch of a few pairs -- what the call combiner behind
    // pa_align creates a thousand times a second -- made some thirty hipMalloc / hipFree calls of 50-100 us each, every hipFree a device wait.
    const bool big = cache_on();
    if (big && bytes < kCacheMin) {
        size_t c = 64;
        while (c < bytes) c = (c + c / 2 >= bytes && (c & (c - 1)) == 0) ? c + c / 2 : ((c & (c - 1)) == 0 ? c * 2 : (c / 3) * 4);
        bytes = c;
    }
    if (big) {
        if (bytes >= kCacheMin) bytes = (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);  // (2 MB steps: requests of almost the same size meet)
        size_t got = 0;
        if (void* p = cache_take(dev, bytes, &got)) {
            ptr = p;
            size = got;
            device = dev;
            g_cache_hits += 1;
            if (poison && !poison_fill(ptr, size)) return false;
            return true;
        }
        g_cache_misses += 1;
    }
    hipError_t e = hipMalloc(&ptr, bytes);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {  // give the cached blocks back and try once more
        (void)hipGetLastError();
        release_alloc_cache();
        e = hipMalloc(&ptr, bytes);
    }
    if (!hip_ok(e, "hipMalloc")) {
        ptr = nullptr;
        return false;
    }
    size = bytes;
    device = dev;
    if (poison && !poison_fill(ptr, size)) return false;
    return true;
}
bool DeviceBuf::reserve(size_t bytes, bool* grew) {
    if (grew) *grew = false;
    if (ptr && size >= bytes) return true;
    const size_t want = bytes + bytes / 4 + 256;  // geometric growth: a pool of buffers reused from call to call
    if (!alloc(want)) return false;
    if (grew) *grew = true;
    return true;
}
// Pinned host buffers (the packed CIGAR text of a batch, its per-pair lengths) are POOLED for the life of the process: hipHostMalloc /
// hipHostFree of a few tens of megabytes cost 5-25 ms each, which a batch that lives for one alignment (the work queue's chunks, pa_align_file)
// paid twice (round 4: `close` of the C4 batch 12-50 ms).  At most kPinnedPoolMax bytes are kept; pa_release_pools() frees them.
namespace {
constexpr size_t kPinnedPoolMax = size_t(1) << 30;
struct PinnedBlock {
    void* ptr;
    size_t size;
};
std::mutex& g_pin_mu = *new std::mutex;
std::vector<PinnedBlock>& g_pin = *new std::vector<PinnedBlock>;
size_t g_pin_bytes = 0;
}  // namespace
void* pinned_take(size_t bytes, size_t* got) {
    {  // size classes (2^k and 1.5 x 2^k from 64 KB up): batches of slightly different sizes -- the call combiner's -- meet in the pool
        size_t c = size_t(64) << 10;
        while (c < bytes) c = ((c & (c - 1)) == 0 && c + c / 2 >= bytes) ? c + c / 2 : ((c & (c - 1)) == 0 ? c * 2 : (c / 3) * 4);
        bytes = c;
    }
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        size_t best = g_pin.size();
        for (size_t i = 0; i < g_pin.size(); ++i)
            // (best fit, and never a block more than twice the request + 64 KB: a 100 KB request must not take the pooled 30 MB text
            //  buffer and send the next text request back to hipHostMalloc)
            if (g_pin[i].size >= bytes && g_pin[i].size <= 2 * bytes + 65536 && (best == g_pin.size() || g_pin[i].size < g_pin[best].size)) best = i;
        if (best != g_pin.size()) {
            void* p = g_pin[best].ptr;
            *got = g_pin[best].size;
            g_pin_bytes -= g_pin[best].size;
            g_pin.erase(g_pin.begin() + (long)best);
            return p;
        }
    }
    void* hp = nullptr;
    if (!hip_ok(hipHostMalloc(&hp, bytes, hipHostMallocDefault), "hipHostMalloc(pinned pool)")) return nullptr;
    *got = bytes;
    return hp;
}
void pinned_give(void* ptr, size_t size) {
    if (!ptr) return;
    std::vector<PinnedBlock> drop;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        g_pin.push_back({ptr, size});
        g_pin_bytes += size;
        while (g_pin_bytes > kPinnedPoolMax && !g_pin.empty()) {
            drop.push_back(g_pin.front());
            g_pin_bytes -= g_pin.front().size;
            g_pin.erase(g_pin.begin());
        }
    }
    for (const PinnedBlock& b : drop) (void)hipHostFree(b.ptr);
}
// ... and so are the chunk streams of pa_batch_align (hipStreamDestroy costs ~3 ms each: a C4 batch of four chunks spent 12 ms of its
// `close` there); per device, idle when they are handed back (the batch's destructor has waited for the device).
namespace {
std::vector<std::pair<int, hipStream_t>>& g_streams = *new std::vector<std::pair<int, hipStream_t>>;
}
hipStream_t stream_take() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_streams.size(); ++i)
            if (g_streams[i].first == dev) {
                hipStream_t s = g_streams[i].second;
                g_streams.erase(g_streams.begin() + (long)i);
                return s;
            }
    }
    hipStream_t s = nullptr;
    if (!hip_ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate")) return nullptr;
    return s;
}
// The batch's own stream (a default, blocking stream: its work orders with the synchronous copies of the creation) is pooled as well:
// hipStreamCreate costs 2 ms, a quarter of what a batch of sixteen short pairs takes from creation to destruction (round 5: the call
// combiner behind pa_align creates such batches a hundred times a second).  A stream goes back only after its batch has waited for the device.
namespace {
std::vector<std::pair<int, hipStream_t>>& g_bstreams = *new std::vector<std::pair<int, hipStream_t>>;
}
hipStream_t bstream_take() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_bstreams.size(); ++i)
            if (g_bstreams[i].first == dev) {
                hipStream_t s = g_bstreams[i].second;
                g_bstreams.erase(g_bstreams.begin() + (long)i);
                return s;
            }
    }
    hipStream_t s = nullptr;
    if (!hip_ok(hipStreamCreate(&s), "hipStreamCreate")) return nullptr;
    return s;
}
void bstream_give(hipStream_t s, int dev) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (g_bstreams.size() < 32) {
            g_bstreams.emplace_back(dev, s);
            return;
        }
    }
    (void)hipStreamDestroy(s);
}
void stream_give(hipStream_t s, int dev) {  // dev: the device the stream was created on (a batch may be destroyed from a thread bound to another)
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (g_streams.size() < 64) {
            g_streams.emplace_back(dev, s);
            return;
        }
    }
    (void)hipStreamDestroy(s);
}
void pinned_release_all() {
    {
        std::vector<std::pair<int, hipStream_t>> st;
        {
            std::lock_guard<std::mutex> lk(g_pin_mu);
            st.swap(g_streams);
        }
        for (auto& x : st) (void)hipStreamDestroy(x.second);
    }
    std::vector<PinnedBlock> all;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        all.swap(g_pin);
        g_pin_bytes = 0;
    }
    for (const PinnedBlock& b : all) (void)hipHostFree(b.ptr);
}

// A batch lets go of a dozen buffers at once: its destructor waits for the device ONCE and the releases that follow skip their wait.
static thread_local bool g_release_synced = false;
void release_scope_begin() {
    (void)hipDeviceSynchronize();
    g_release_synced = true;
}
// ... or the owner waited for everything that ever touched its buffers itself (a batch: its own streams) and only declares the scope:
// a device-wide wait also waits for every OTHER batch in flight -- eight batches of the call combiner side by side each waited for the
// other seven's kernels (round 5: 40 ms per call at 64 callers instead of 8)
void release_scope_begin_waited() { g_release_synced = true; }
void release_scope_end() { g_release_synced = false; }

void DeviceBuf::release() {
    if (ptr) {
        if (cache_on() && g_release_synced) {  // (inside a release scope the device has been waited for: any size goes to the cache)
            int cur = device;
            (void)hipGetDevice(&cur);
            if (cur == device) {
                free_blocks(cache_put(device, ptr, size));
                ptr = nullptr;
                size = 0;
                return;
            }
        }
        if (size >= kCacheMin && cache_on()) {
            // hipFree waits for the device before it lets a buffer go; a cached block may be handed to another thread at once, so
            // this waits too (whoever must not wait -- the sweep's pool while passes are in flight -- never frees, engine_hip.hip)
            int cur = device;
            (void)hipGetDevice(&cur);
            if (cur != device) (void)hipSetDevice(device);  // (a batch destroyed from a thread bound to another GPU)
            (void)hipDeviceSynchronize();
            if (cur != device) (void)hipSetDevice(cur);
            free_blocks(cache_put(device, ptr, size));
        } else {
            (void)hipFree(ptr);
        }
    }
    ptr = nullptr;
    size = 0;
}

// Both encodings of ONE pair in one launch (the single-pair engine's per-call set-up, round 6: eight stream operations were four): the
// first blocks pack a -- every word of `codes` up to code_words, so the padding the kernels read past the last column is zeroed here --,
// the others build b's profile.  `bad` may be host-mapped: every writer stores the same 1.
__global__ void encode_pair_kernel(const uint8_t* __restrict__ a, int n, uint32_t* __restrict__ codes, int code_words, int a_blocks,
                                   const uint8_t* __restrict__ b, int m, uint64_t* __restrict__ prof, int prof_words, uint32_t* bad) {
    if ((int)blockIdx.x < a_blocks) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= code_words) return;
        uint32_t w = 0;
        bool invalid = false;
        for (int k = 0; k < 16; ++k) {
            const int c = i * 16 + k;
            if (c < n) {
                const int r = rank_acgt(a[c]);
                invalid |= r < 0;
                w |= (uint32_t)(r & 3) << (2 * k);
            }
        }
        codes[i] = w;
        if (invalid) __hip_atomic_store(bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const int word = ((int)blockIdx.x - a_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (word >= prof_words) return;
    const int lane = threadIdx.x & 63;
    const int j = word * 64 + lane;
    int r = 3;
    bool invalid = false;
    if (j < m) {
        r = rank_acgt(b[j]);
        invalid = r < 0;
        r &= 3;
    }
    const uint64_t nb0 = __ballot(((r & 1) ^ 1) != 0);
    const uint64_t nb1 = __ballot((((r >> 1) & 1) ^ 1) != 0);
    if (lane == 0) {
        prof[2 * word] = nb0;
        prof[2 * word + 1] = nb1;
    }
    if (invalid) __hip_atomic_store(bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
bool encode_pair_device(const uint8_t* d_a, int n, uint32_t* d_codes, int code_words, const uint8_t* d_b, int m, uint64_t* d_prof, uint32_t* bad,
                        hipStream_t s) {
    const int a_blocks = (code_words + 255) / 256, prof_words = (m + 63) / 64, b_blocks = (prof_words + 3) / 4;
    if (a_blocks + b_blocks == 0) return true;
    hipLaunchKernelGGL(encode_pair_kernel, dim3((unsigned)(a_blocks + b_blocks)), dim3(256), 0, s, d_a, n, d_codes, code_words, a_blocks, d_b, m, d_prof,
                       prof_words, bad);
    return hip_ok(hipGetLastError(), "encode_pair_kernel");
}

bool encode_a_device(const uint8_t* d_a, int n, uint32_t* d_codes, uint32_t* d_bad, hipStream_t s) {
    const int nwords = (n + 15) / 16;
    if (nwords == 0) return true;
    hipLaunchKernelGGL(encode_a_kernel, dim3((nwords + 255) / 256), dim3(256), 0, s, d_a, n, d_codes, nwords, d_bad);
    return hip_ok(hipGetLastError(), "encode_a_kernel");
}

bool build_b_device(const uint8_t* d_b, int m, uint64_t* d_prof, uint32_t* d_bad, hipStream_t s) {
    const int nwords = (m + 63) / 64;
    if (nwords == 0) return true;
    hipLaunchKernelGGL(build_b_kernel, dim3((nwords + 3) / 4), dim3(256), 0, s, d_b, m, d_prof, nwords, d_bad);
    return hip_ok(hipGetLastError(), "build_b_kernel");
}

// How a rectangle of w words is cut into strips.  Chained strips all have the kernel's height (32*k words).  A sequential
// pair may finish with up to kMaxTail1[k] short strips of 32 words when that is cheaper than one mostly empty tall strip
// (a k = 1 step costs about 0.67 / 0.43 / 0.25 of a k = 2 / 4 / 8 step).
StripPlan strip_plan(int w, int k, bool sequential) {
    StripPlan p;
    const int wps = kWordsPerStrip * k;
    p.full = w / wps;
    const int r = w - p.full * wps;
    if (r == 0) return p;
    const int max_tail1 = !sequential ? 0 : (k == 2 ? 1 : (k == 4 ? 2 : (k >= 8 ? 3 : 0)));
    const int t1 = (r + kWordsPerStrip - 1) / kWordsPerStrip;
    if (t1 <= max_tail1) p.tail1 = t1;
    else p.full += 1;
    return p;
}

// Plan the strips of one rectangle: words [w0, w1) x n columns, r.k subwords per lane.
void plan_rect(std::vector<StripJob>& jobs, const RectPlan& r) {
    const int w = r.w1 - r.w0;
    const int wps = kWordsPerStrip * r.k;
    const StripPlan sp = strip_plan(w, r.k, r.pingpong);
    const int S = sp.strips();
    int word = 0;
    for (int s = 0; s < S; ++s) {
        StripJob j;
        std::memset(&j, 0, sizeof j);
        const bool tall = s < sp.full;
        j.k = tall ? r.k : 1;
        j.a_codes = r.a_codes;
        j.b_prof = r.b_prof;
        j.v = r.v;
        j.n = r.n;
        j.col0 = r.col0;
        j.word0 = r.w0 + word;
        const int words = std::min(tall ? wps : kWordsPerStrip, w - word);
        j.nlanes = 2 * words;
        j.flags = r.v_init_one ? kJobVInitOne : 0;
        j.tail_rows = -1;
        if (s == 0) {
            j.hin_arr = r.hin_arr;  // nullptr => +1
        } else {
            j.hin_gran = r.gran + (size_t)(r.pingpong ? ((s - 1) & 1) : (s - 1)) * r.gran_stride;
        }
        if (s + 1 < S) {
            j.hout_gran = r.gran + (size_t)(r.pingpong ? (s & 1) : s) * r.gran_stride;
            j.exact_tail = 1;  // full strips anyway
        } else {
            j.hout_arr = r.hout_arr;
            j.sum_out = r.sum_out;
            j.tail_rows = r.tail_rows;
            j.exact_tail = (r.exact_end || r.hout_arr) ? 1 : 0;
        }
        if (r.values) {
            j.values = r.values;
            j.fill_stride = r.fill_stride;
            j.fill_word0 = r.fill_word0 + word;
        }
        j.ckpt = r.ckpt;
        j.ckpt_stride = r.ckpt_stride;
        word += words;
        jobs.push_back(j);
    }
}

size_t rect_granules(int n, int w, int k, bool pingpong) {
    const int S = strip_plan(w, k, pingpong).strips();
    const size_t G = (size_t)(n + 31) / 32;  // one 8-byte granule per 32 columns per strip boundary
    const int rows = pingpong ? std::min(S - 1, 2) : S - 1;
    return S > 1 ? (size_t)rows * G : 0;
}

// Residency cap.  Every strip of a rectangle advances at the pace of the most crowded SIMD it touches, and the
// dispatcher does not balance SIMDs by itself.  Blocks are 4 wavefronts (one per SIMD of a CU); an unused dynamic-LDS
// request sized so that only `W = ceil(blocks / CUs)` blocks fit in a CU's 160 KB makes W the hard maximum of
// wavefronts per SIMD instead of an average.
static unsigned residency_lds_bytes(int blocks) {
    static const bool off = getenv("PA_STRIP_NO_LDS_CAP") != nullptr;
    if (off) return 0;
    const int cus = g_device_props_cus > 0 ? g_device_props_cus : 256;
    const int W = (blocks + cus - 1) / cus;
    if (W > 7) return 0;  // beyond the register-file limit nothing is gained
    const unsigned lds_total = 160u * 1024u;
    return ((lds_total / (unsigned)(W + 1)) + 1024u) & ~1023u;  // W blocks fit, W + 1 do not
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device), from any
// thread (the library is re-entrant; pa_set_device selects the device per thread).
static bool ensure_max_lds(const void* kern) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    if (!hip_ok(hipGetDevice(&dev), "hipGetDevice")) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kern, dev})) return true;
    if (!hip_ok(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "hipFuncSetAttribute(max dynamic LDS)")) return false;
    done.insert({kern, dev});
    return true;
}

template <class Kern>
static bool launch_one(Kern kern, int grid, int block_waves, unsigned lds, hipStream_t s, const StripJob* d_jobs, int njobs,
                       uint32_t* d_ticket_err) {
    if (!ensure_max_lds(reinterpret_cast<const void*>(kern))) return false;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * block_waves), lds, s, d_jobs, njobs, d_ticket_err, d_ticket_err + 1);
    return hip_ok(hipGetLastError(), "strip_kernel launch");
}

bool launch_strips(const StripJob* d_jobs, int njobs, bool fill, uint32_t* d_ticket_err, hipStream_t s, bool zero_ticket, bool scatter,
                   int k, int block_waves, bool ckpt) {
    if (njobs == 0) return true;
    // d_ticket_err[0] = ticket, [1] = err
    if (zero_ticket && !hip_ok(hipMemsetAsync(d_ticket_err, 0, 2 * sizeof(uint32_t), s), "memset ticket")) return false;
    if (block_waves < 1 || block_waves > kStripMaxBlockWaves) block_waves = kStripBlockWaves;
    if (const char* e = getenv("PA_STRIP_BLOCK_WAVES")) block_waves = std::min(std::max(atoi(e), 1), kStripMaxBlockWaves);  // experiments
    if (k < 4 && block_waves > kStripBlockWaves) block_waves = kStripBlockWaves;  // (the k = 1, 2 kernels are built for 256 threads)
    const int grid = (njobs + block_waves - 1) / block_waves;  // one wave per job; jobs beyond residency queue behind their
                                                               // producers (ticket order)
    const unsigned lds = block_waves >= kStripBlockWaves ? residency_lds_bytes(grid) : 0;
    if ((scatter || fill) && (k != 1 || ckpt)) {
        set_error("fill / scatter strips are built for k = 1 without checkpoints only");
        return false;
    }
    // tall cost-only strips take their eq words from LDS: one slice per wavefront of the block
    static const bool no_ldseq = getenv("PA_STRIP_NO_LDSEQ") != nullptr;
    if (!no_ldseq && !scatter && !fill && (k == 4 || k == 8)) {
        const unsigned need = (unsigned)block_waves * (k == 8 ? LdsEq<8>::kWaveBytes : LdsEq<4>::kWaveBytes);
        const unsigned l = std::max(lds, need);
        if (ckpt && k == 4) return launch_one(strip_kernel<4, false, false, true, true>, grid, block_waves, l, s, d_jobs, njobs, d_ticket_err);
        if (ckpt && k == 8) return launch_one(strip_kernel<8, false, false, true, true>, grid, block_waves, l, s, d_jobs, njobs, d_ticket_err);
        if (k == 4) return launch_one(strip_kernel<4, false, false, false, true>, grid, block_waves, l, s, d_jobs, njobs, d_ticket_err);
        return launch_one(strip_kernel<8, false, false, false, true>, grid, block_waves, l, s, d_jobs, njobs, d_ticket_err);
    }
    if (ckpt) {
        if (k == 1) return launch_one(strip_kernel<1, false, false, true>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
        if (k == 2) return launch_one(strip_kernel<2, false, false, true>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
        if (k == 4) return launch_one(strip_kernel<4, false, false, true>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
        if (k == 8) return launch_one(strip_kernel<8, false, false, true>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
    }
    if (scatter && fill) return launch_one(strip_kernel<1, true, true>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
    if (scatter) return launch_one(strip_kernel<1, false, true>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
    if (fill) return launch_one(strip_kernel<1, true, false>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
    if (k == 1) return launch_one(strip_kernel<1, false, false>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
    if (k == 2) return launch_one(strip_kernel<2, false, false>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
    if (k == 4) return launch_one(strip_kernel<4, false, false>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
    if (k == 8) return launch_one(strip_kernel<8, false, false>, grid, block_waves, lds, s, d_jobs, njobs, d_ticket_err);
    set_error("unsupported strip height k=%d", k);
    return false;
}

template <int K, bool CKPT>
static bool launch_pairs_k(const StripJob* d_jobs, const int32_t* d_first, int npairs, uint32_t* d_err, hipStream_t s, int grid, unsigned lds) {
    // tall strips take their eq words from LDS (strip_kernel.hpp LdsEq): one slice per wavefront of the block
    static const bool no_ldseq = getenv("PA_PAIR_NO_LDSEQ") != nullptr;
    if (K >= 4 && !no_ldseq) {
        constexpr bool L = K >= 4;  // (keeps the K < 4 instantiations out of the binary)
        const unsigned need = (unsigned)kStripBlockWaves * LdsEq<K>::kWaveBytes;
        if (!ensure_max_lds(reinterpret_cast<const void*>(pair_kernel<K, CKPT, L>))) return false;
        hipLaunchKernelGGL((pair_kernel<K, CKPT, L>), dim3(grid), dim3(64 * kStripBlockWaves), std::max(lds, need), s, d_jobs, d_first, npairs, d_err);
        return hip_ok(hipGetLastError(), "pair_kernel launch");
    }
    if (!ensure_max_lds(reinterpret_cast<const void*>(pair_kernel<K, CKPT, false>))) return false;
    hipLaunchKernelGGL((pair_kernel<K, CKPT, false>), dim3(grid), dim3(64 * kStripBlockWaves), lds, s, d_jobs, d_first, npairs, d_err);
    return hip_ok(hipGetLastError(), "pair_kernel launch");
}

bool launch_pairs(const StripJob* d_jobs, const int32_t* d_first, int npairs, uint32_t* d_ticket_err, hipStream_t s, int k, bool ckpt) {
    if (npairs == 0) return true;
    const int grid = (npairs + kStripBlockWaves - 1) / kStripBlockWaves;
    const unsigned lds = residency_lds_bytes(grid);
    uint32_t* e = d_ticket_err + 1;
    if (!ckpt) {
        if (k == 1) return launch_pairs_k<1, false>(d_jobs, d_first, npairs, e, s, grid, lds);
        if (k == 2) return launch_pairs_k<2, false>(d_jobs, d_first, npairs, e, s, grid, lds);
        if (k == 4) return launch_pairs_k<4, false>(d_jobs, d_first, npairs, e, s, grid, lds);
        if (k == 8) return launch_pairs_k<8, false>(d_jobs, d_first, npairs, e, s, grid, lds);
        if (k == 16) return launch_pairs_k<16, false>(d_jobs, d_first, npairs, e, s, grid, 0);
    } else {
        if (k == 1) return launch_pairs_k<1, true>(d_jobs, d_first, npairs, e, s, grid, lds);
        if (k == 2) return launch_pairs_k<2, true>(d_jobs, d_first, npairs, e, s, grid, lds);
        if (k == 4) return launch_pairs_k<4, true>(d_jobs, d_first, npairs, e, s, grid, lds);
        if (k == 8) return launch_pairs_k<8, true>(d_jobs, d_first, npairs, e, s, grid, lds);
    }
    set_error("unsupported strip height k=%d", k);
    return false;
}

// CIGAR text on the GPU.  trace_kernel leaves each pair's elements (count << 2 | op) from the END of the alignment to its
// start; one wavefront per pair turns them into the reference's string form (count omitted when 1, ops "=XID",
// pa-types Cigar::to_string as pinned by astarpa-c/example.cpp:16) and writes it straight into the chunk's packed region: a first
// pass over the elements counts the characters, one atomic add claims that much of the region, a second pass writes them.
// Block b handles pair list[b]; its text length and offset land at position b of tlen / dst (kTextFailed: the traceback handed the
// pair back).
enum : uint32_t { kTextFailed = 0xFFFFFFFFu };
__global__ __launch_bounds__(64) void format_pack_kernel(const uint32_t* __restrict__ elems, const uint64_t* __restrict__ off,
                                                         const uint32_t* __restrict__ len, const int32_t* __restrict__ list,
                                                         uint8_t* __restrict__ packed, unsigned long long* __restrict__ total,
                                                         uint32_t* __restrict__ tlen, uint64_t* __restrict__ dst) {
    const int pair = list[blockIdx.x];
    const uint32_t n = len[pair];
    const int lane = (int)threadIdx.x;
    if (n == kTraceFailed) {
        if (lane == 0) {
            tlen[blockIdx.x] = kTextFailed;
            dst[blockIdx.x] = 0;
        }
        return;
    }
    const uint32_t* e = elems + off[pair];
    auto chars_of = [](uint32_t v) -> uint32_t {
        const uint32_t cnt = v >> 2;
        uint32_t chars = 1;
        if (cnt != 1) {
            uint32_t c = cnt;
            do {
                ++chars;
                c /= 10;
            } while (c);
        }
        return chars;
    };
    uint32_t mine = 0;
    for (uint32_t k = (uint32_t)lane; k < n; k += 64) mine += chars_of(e[k]);
    mine = (uint32_t)wave_add((int32_t)mine);
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(total, (unsigned long long)mine);
    base = ((unsigned long long)rfl((uint32_t)(base >> 32)) << 32) | (unsigned long long)rfl((uint32_t)base);
    uint8_t* out = packed + base;
    uint32_t pos = 0;
    for (uint32_t b0 = 0; b0 < n; b0 += 64) {
        const uint32_t k = b0 + (uint32_t)lane;  // k-th element of the OUTPUT = element n-1-k of the stored run
        uint32_t v = 0, chars = 0;
        if (k < n) {
            v = e[n - 1 - k];
            chars = chars_of(v);
        }
        const uint32_t incl = (uint32_t)wave_scan_add((int32_t)chars);  // inclusive prefix sum of `chars` over the wavefront
        const uint32_t start = pos + incl - chars;
        if (k < n) {
            uint8_t* w = out + start + chars - 1;
            *w-- = (uint8_t)"=XID"[v & 3u];
            const uint32_t cnt = v >> 2;
            if (cnt != 1) {
                uint32_t c = cnt;
                do {
                    *w-- = (uint8_t)('0' + c % 10);
                    c /= 10;
                } while (c);
            }
        }
        pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) {
        tlen[blockIdx.x] = pos;
        dst[blockIdx.x] = base;
    }
}

}  // namespace pa

using namespace pa;

// ================================================================================================
// C ABI
// ================================================================================================

extern "C" const char* pa_last_error(void) { return g_last_error.c_str(); }

extern "C" int pa_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

extern "C" int pa_set_device(int device) {
    if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return PA_E_HIP;
    return 0;
}

extern "C" int pa_bp_profile_build(const uint8_t* a, size_t n, const uint8_t* b, size_t m, uint64_t* a2,
                                   uint64_t* b2) {
    if (!ensure_device()) return PA_E_HIP;
    const size_t w = (m + 63) / 64, cw = (n + 15) / 16;
    DeviceBuf d_a, d_b, d_codes, d_prof, d_bad;
    if (!d_a.alloc(n) || !d_b.alloc(m) || !d_codes.alloc(cw * 4) || !d_prof.alloc(w * 16) || !d_bad.alloc(4))
        return PA_E_HIP;
    hipStream_t s = 0;
    if (n && !hip_ok(hipMemcpyAsync(d_a.ptr, a, n, hipMemcpyHostToDevice, s), "H2D a")) return PA_E_HIP;
    if (m && !hip_ok(hipMemcpyAsync(d_b.ptr, b, m, hipMemcpyHostToDevice, s), "H2D b")) return PA_E_HIP;
    if (!hip_ok(hipMemsetAsync(d_bad.ptr, 0, 4, s), "memset")) return PA_E_HIP;
    if (!encode_a_device(d_a.as<uint8_t>(), (int)n, d_codes.as<uint32_t>(), d_bad.as<uint32_t>(), s)) return PA_E_HIP;
    if (!build_b_device(d_b.as<uint8_t>(), (int)m, d_prof.as<uint64_t>(), d_bad.as<uint32_t>(), s)) return PA_E_HIP;
    std::vector<uint32_t> codes(cw);
    uint32_t bad = 0;
    if (cw && !hip_ok(hipMemcpyAsync(codes.data(), d_codes.ptr, cw * 4, hipMemcpyDeviceToHost, s), "D2H codes")) return PA_E_HIP;
    if (w && !hip_ok(hipMemcpyAsync(b2, d_prof.ptr, w * 16, hipMemcpyDeviceToHost, s), "D2H prof")) return PA_E_HIP;
    if (!hip_ok(hipMemcpyAsync(&bad, d_bad.ptr, 4, hipMemcpyDeviceToHost, s), "D2H bad")) return PA_E_HIP;
    if (!hip_ok(hipStreamSynchronize(s), "sync")) return PA_E_HIP;
    if (bad) {
        set_error("sequence contains a base outside ACGT");
        return PA_E_INVALID_BASE;
    }
    for (size_t i = 0; i < n; ++i) {  // exploded Bits of a (profile.rs:116-125)
        const uint32_t r = (codes[i / 16] >> (2 * (i % 16))) & 3u;
        a2[2 * i] = 0ull - (uint64_t)(r & 1);
        a2[2 * i + 1] = 0ull - (uint64_t)((r >> 1) & 1);
    }
    return 0;
}

// Shared implementation of pa_bp_compute / pa_bp_fill on host buffers.
static int32_t rect_host(const uint64_t* a2, size_t n, const uint64_t* b2, size_t w, uint64_t* h2, uint64_t* v2,
                         int exact_end, uint64_t* values) {
    if (!ensure_device()) return INT32_MIN;
    if (n > (size_t)INT32_MAX / 2 || w > (size_t)INT32_MAX / 64) {
        set_error("rectangle too large");
        return INT32_MIN;
    }
    if (n == 0) return 0;
    if (w == 0) {  // no rows: bottom == top
        int32_t s = 0;
        for (size_t i = 0; i < n; ++i) s += (int32_t)h2[2 * i] - (int32_t)h2[2 * i + 1];
        return s;
    }
    const size_t cw = (n + 15) / 16;
    std::vector<uint32_t> codes(cw, 0);
    std::vector<uint8_t> hin(n, 0);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t r = (uint32_t)(a2[2 * i] & 1) | ((uint32_t)(a2[2 * i + 1] & 1) << 1);
        codes[i / 16] |= r << (2 * (i % 16));
        hin[i] = (uint8_t)((h2[2 * i] & 1) | ((h2[2 * i + 1] & 1) << 1));
    }
    const size_t ngran = rect_granules((int)n, (int)w);
    DeviceBuf d_codes, d_prof, d_v, d_hin, d_hout, d_gran, d_jobs, d_misc, d_values;
    if (!d_codes.alloc(cw * 4) || !d_prof.alloc(w * 16) || !d_v.alloc(w * 16) || !d_hin.alloc(n) ||
        !d_hout.alloc(n) || !d_gran.alloc(ngran * 8) || !d_misc.alloc(16))
        return INT32_MIN;
    if (values && !d_values.alloc(n * w * 16)) return INT32_MIN;
    hipStream_t s = 0;
    bool ok = hip_ok(hipMemcpyAsync(d_codes.ptr, codes.data(), cw * 4, hipMemcpyHostToDevice, s), "H2D") &&
              hip_ok(hipMemcpyAsync(d_prof.ptr, b2, w * 16, hipMemcpyHostToDevice, s), "H2D") &&
              hip_ok(hipMemcpyAsync(d_v.ptr, v2, w * 16, hipMemcpyHostToDevice, s), "H2D") &&
              hip_ok(hipMemcpyAsync(d_hin.ptr, hin.data(), n, hipMemcpyHostToDevice, s), "H2D") &&
              hip_ok(hipMemsetAsync(d_gran.ptr, 0, std::max<size_t>(ngran * 8, 16), s), "memset gran") &&
              hip_ok(hipMemsetAsync(d_misc.ptr, 0, 16, s), "memset misc");
    if (!ok) return INT32_MIN;

    std::vector<StripJob> jobs;
    RectPlan r;
    r.a_codes = d_codes.as<uint32_t>();
    r.b_prof = d_prof.as<uint32_t>();
    r.v = d_v.as<uint32_t>();
    r.n = (int)n;
    r.w0 = 0;
    r.w1 = (int)w;
    r.hin_arr = d_hin.as<uint8_t>();
    r.hout_arr = d_hout.as<uint8_t>();
    r.gran = d_gran.as<uint64_t>();
    r.gran_stride = (n + 31) / 32;
    r.sum_out = d_misc.as<int32_t>() + 2;
    r.exact_end = exact_end != 0 || values != nullptr;
    if (!exact_end && !values) r.hout_arr = nullptr;  // padded-tail path: the bottom row itself is not an output
    r.values = values ? d_values.as<uint32_t>() : nullptr;
    r.fill_stride = (int)w;
    r.fill_word0 = 0;
    plan_rect(jobs, r);
    if (!d_jobs.alloc(jobs.size() * sizeof(StripJob))) return INT32_MIN;
    ok = hip_ok(hipMemcpyAsync(d_jobs.ptr, jobs.data(), jobs.size() * sizeof(StripJob), hipMemcpyHostToDevice, s), "H2D jobs") &&
         launch_strips(d_jobs.as<StripJob>(), (int)jobs.size(), values != nullptr, d_misc.as<uint32_t>(), s);
    if (!ok) return INT32_MIN;
    uint32_t misc[4] = {0, 0, 0, 0};
    std::vector<uint8_t> hout(n, 0);
    ok = hip_ok(hipMemcpyAsync(misc, d_misc.ptr, 16, hipMemcpyDeviceToHost, s), "D2H") &&
         hip_ok(hipMemcpyAsync(v2, d_v.ptr, w * 16, hipMemcpyDeviceToHost, s), "D2H") &&
         (r.hout_arr == nullptr || hip_ok(hipMemcpyAsync(hout.data(), d_hout.ptr, n, hipMemcpyDeviceToHost, s), "D2H")) &&
         (!values || hip_ok(hipMemcpyAsync(values, d_values.ptr, n * w * 16, hipMemcpyDeviceToHost, s), "D2H values")) &&
         hip_ok(hipStreamSynchronize(s), "sync");
    if (!ok) return INT32_MIN;
    if (misc[1] != PA_ERR_NONE) {
        set_error("device spin timeout (err=%u)", misc[1]);
        return INT32_MIN;
    }
    if (r.hout_arr) {
        for (size_t i = 0; i < n; ++i) {
            const uint32_t x = hout[i] & 3u;
            h2[2 * i] = x & 1;
            h2[2 * i + 1] = x >> 1;
        }
    }
    // exact_end == 0: the reference leaves h unspecified (simd.rs:184-225); h2 is left untouched.
    return (int32_t)misc[2];
}

extern "C" int32_t pa_bp_compute(const uint64_t* a2, size_t n, const uint64_t* b2, size_t w, uint64_t* h2, uint64_t* v2,
                                 int exact_end) {
    return rect_host(a2, n, b2, w, h2, v2, exact_end, nullptr);
}

extern "C" int32_t pa_bp_fill(const uint64_t* a2, size_t n, const uint64_t* b2, size_t w, uint64_t* h2, uint64_t* v2,
                              uint64_t* values) {
    return rect_host(a2, n, b2, w, h2, v2, 1, values);
}

// ---- semi-global search (pa_bitpacking::search, pa-bitpacking/src/search.rs:46-120) ------------------------------

// One thread per 16 columns: text ASCII -> packed CC codes (A0 C1 T2 G3, either case; profile.rs:30-38).
__global__ void encode_text_cc_kernel(const uint8_t* __restrict__ a, int n, uint32_t* __restrict__ codes, int nwords,
                                      uint32_t* __restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    uint32_t w = 0;
    bool invalid = false;
    for (int k = 0; k < 16; ++k) {
        const int c = i * 16 + k;
        if (c < n) {
            const uint8_t ch = a[c] & 0xDF;  // upper-case
            const int r = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'T' ? 2 : ch == 'G' ? 3 : -1;
            invalid |= r < 0;
            w |= (uint32_t)(r & 3) << (2 * k);
        }
    }
    codes[i] = w;
    if (invalid) atomicOr(bad, 1u);
}

// ---- semi-global search (pa-bitpacking/src/search.rs) ------------------------------------------------------------
namespace {

// ScatterProfile of the pattern (profile.rs:39-63: wildcards N/* (any), Y (C|T), R (A|G); padding rows match everything)
// and the left column of the search (every ceil(i / unmatched_cost)-th row costs 1, search.rs:57-65).
int search_profile(const uint8_t* pattern, size_t plen, float unmatched_cost, std::vector<uint64_t>& prof, std::vector<uint64_t>& v0) {
    const size_t w = (plen + 63) / 64;
    prof.assign(4 * std::max<size_t>(w, 1), 0);
    for (size_t j = 0; j < plen; ++j) {
        int mask;
        switch (pattern[j]) {
            case 'a': case 'A': mask = 1; break;
            case 'c': case 'C': mask = 2; break;
            case 't': case 'T': mask = 4; break;
            case 'g': case 'G': mask = 8; break;
            case 'n': case 'N': case '*': mask = 15; break;
            case 'y': case 'Y': mask = 6; break;
            case 'r': case 'R': mask = 9; break;
            default: set_error("Unknown base in pattern"); return PA_E_INVALID_BASE;
        }
        for (int c = 0; c < 4; ++c)
            if (mask & (1 << c)) prof[4 * (j / 64) + c] |= 1ull << (j % 64);
    }
    for (size_t j = plen; j < w * 64; ++j)
        for (int c = 0; c < 4; ++c) prof[4 * (j / 64) + c] |= 1ull << (j % 64);
    v0.assign(2 * std::max<size_t>(w, 1), 0);
    if (unmatched_cost > 0.0f) {
        for (size_t i = 0;; ++i) {
            const size_t idx = (size_t)std::ceil((float)i / unmatched_cost);
            if (idx >= plen) break;
            v0[2 * (idx / 64)] |= 1ull << (idx % 64);
        }
    }
    return 0;
}

// scatter_profile::compute::<2, _, 4, FILL>(text[0..n), pattern profile, h = zeros, v, exact_end = true, values)
// (search.rs:71,152) on the GPU: v is updated in place, hrow[n] receives the bottom-row deltas (bit0 = +1, bit1 = -1),
// values (optional) the V of every word after every column (values[col * w + word], two u64 each).
int search_rect(const uint8_t* text, size_t n, const std::vector<uint64_t>& prof, size_t w, std::vector<uint64_t>& v,
                std::vector<uint8_t>& hrow, std::vector<uint64_t>* values) {
    hrow.assign(std::max<size_t>(n, 1), 0);
    if (values) values->assign(n * w * 2, 0);
    if (n == 0 || w == 0) return 0;
    const bool fill = values != nullptr;
    const size_t cw = (n + 15) / 16 + 2, ngran = rect_granules((int)n, (int)w);
    DeviceBuf d_text, d_codes, d_prof, d_v, d_hin, d_hout, d_gran, d_jobs, d_misc, d_values;
    if (!d_text.alloc(n) || !d_codes.alloc(cw * 4) || !d_prof.alloc(w * 32) || !d_v.alloc(w * 16) || !d_hin.alloc(n) ||
        !d_hout.alloc(n) || !d_gran.alloc(ngran * 8) || !d_misc.alloc(16) || (fill && !d_values.alloc(n * w * 16)))
        return PA_E_HIP;
    hipStream_t s = 0;
    bool ok = hip_ok(hipMemcpyAsync(d_text.ptr, text, n, hipMemcpyHostToDevice, s), "H2D") &&
              hip_ok(hipMemsetAsync(d_codes.ptr, 0, cw * 4, s), "memset") && hip_ok(hipMemsetAsync(d_misc.ptr, 0, 16, s), "memset") &&
              hip_ok(hipMemsetAsync(d_hin.ptr, 0, n, s), "memset h") &&  // zeros along the top: start anywhere in the text
              hip_ok(hipMemsetAsync(d_gran.ptr, 0, std::max<size_t>(ngran * 8, 16), s), "memset gran") &&
              hip_ok(hipMemcpyAsync(d_prof.ptr, prof.data(), w * 32, hipMemcpyHostToDevice, s), "H2D") &&
              hip_ok(hipMemcpyAsync(d_v.ptr, v.data(), w * 16, hipMemcpyHostToDevice, s), "H2D");
    if (!ok) return PA_E_HIP;
    const int nwords = (int)((n + 15) / 16);
    hipLaunchKernelGGL(encode_text_cc_kernel, dim3((nwords + 255) / 256), dim3(256), 0, s, d_text.as<uint8_t>(), (int)n,
                       d_codes.as<uint32_t>(), nwords, d_misc.as<uint32_t>() + 3);
    std::vector<StripJob> jobs;
    RectPlan r;
    r.a_codes = d_codes.as<uint32_t>();
    r.b_prof = d_prof.as<uint32_t>();
    r.v = d_v.as<uint32_t>();
    r.n = (int)n;
    r.w0 = 0;
    r.w1 = (int)w;
    r.hin_arr = d_hin.as<uint8_t>();
    r.hout_arr = d_hout.as<uint8_t>();
    r.gran = d_gran.as<uint64_t>();
    r.gran_stride = (n + 31) / 32;
    r.sum_out = d_misc.as<int32_t>() + 2;
    r.exact_end = true;
    if (fill) {
        r.values = d_values.as<uint32_t>();
        r.fill_stride = (int)w;
        r.fill_word0 = 0;
    }
    plan_rect(jobs, r);
    if (!d_jobs.alloc(jobs.size() * sizeof(StripJob))) return PA_E_HIP;
    uint32_t misc[4] = {0, 0, 0, 0};
    ok = hip_ok(hipMemcpyAsync(d_jobs.ptr, jobs.data(), jobs.size() * sizeof(StripJob), hipMemcpyHostToDevice, s), "H2D jobs") &&
         launch_strips(d_jobs.as<StripJob>(), (int)jobs.size(), fill, d_misc.as<uint32_t>(), s, false, /*scatter=*/true) &&
         hip_ok(hipMemcpyAsync(misc, d_misc.ptr, 16, hipMemcpyDeviceToHost, s), "D2H") &&
         hip_ok(hipMemcpyAsync(v.data(), d_v.ptr, w * 16, hipMemcpyDeviceToHost, s), "D2H") &&
         hip_ok(hipMemcpyAsync(hrow.data(), d_hout.ptr, n, hipMemcpyDeviceToHost, s), "D2H") &&
         (!fill || hip_ok(hipMemcpyAsync(values->data(), d_values.ptr, n * w * 16, hipMemcpyDeviceToHost, s), "D2H values")) &&
         hip_ok(hipStreamSynchronize(s), "sync");
    if (!ok) return PA_E_HIP;
    if (misc[3]) {
        set_error("text must be actgACTG only");
        return PA_E_INVALID_BASE;
    }
    if (misc[1] != PA_ERR_NONE) {
        set_error("device spin timeout (err=%u)", misc[1]);
        return PA_E_TIMEOUT;
    }
    return 0;
}

int32_t v_value(uint64_t p, uint64_t m) { return (int32_t)__builtin_popcountll(p) - (int32_t)__builtin_popcountll(m); }
int32_t v_suffix(uint64_t p, uint64_t m, int j) {  // V::value_of_suffix, encoding.rs:35-40
    const uint64_t mask = ~((1ull << (64 - j)) - 1);
    return (int32_t)__builtin_popcountll(p & mask) - (int32_t)__builtin_popcountll(m & mask);
}
int32_t vec_value_to(const uint64_t* v, int64_t j) {  // V::value_to, encoding.rs:57-66
    int32_t s = 0;
    for (int64_t k = 0; k < j / 64; ++k) s += v_value(v[2 * k], v[2 * k + 1]);
    if (j % 64 != 0) {
        const uint64_t mask = (1ull << (j % 64)) - 1;
        s += (int32_t)__builtin_popcountll(v[2 * (j / 64)] & mask) - (int32_t)__builtin_popcountll(v[2 * (j / 64) + 1] & mask);
    }
    return s;
}
int32_t vec_value_from(const uint64_t* v, size_t w, int64_t j) {  // V::value_from, encoding.rs:67-76
    int32_t s = 0;
    if (j % 64 != 0) s += v_suffix(v[2 * (j / 64)], v[2 * (j / 64) + 1], (int)(64 - j % 64));
    for (size_t k = (size_t)((j + 63) / 64); k < w; ++k) s += v_value(v[2 * k], v[2 * k + 1]);
    return s;
}

int search_out(const uint8_t* pattern, size_t plen, const uint8_t* text, size_t tlen, float unmatched_cost, std::vector<int32_t>& out,
               std::vector<uint64_t>& prof, std::vector<uint64_t>& v0) {
    if (!ensure_device()) return PA_E_HIP;
    if (!(unmatched_cost >= 0.0f && unmatched_cost <= 1.0f) || plen > (size_t)(1u << 30) || tlen > (size_t)(1u << 30)) {
        set_error("pa_search: bad argument");
        return PA_E_ARG;
    }
    const size_t w = (plen + 63) / 64, n = tlen;
    if (const int rc = search_profile(pattern, plen, unmatched_cost, prof, v0)) return rc;
    std::vector<uint64_t> v(v0);
    std::vector<uint8_t> hrow;
    if (const int rc = search_rect(text, n, prof, w, v, hrow, nullptr)) return rc;
    // Assemble the bottom row then the right column in reverse (search.rs:73-100).
    out.clear();
    const size_t padding = w * 64 - plen;
    int32_t bsum = 0;
    for (size_t j = 0; j < w; ++j) bsum += v_value(v0[2 * j], v0[2 * j + 1]);
    size_t skipped = 0;
    out.push_back(bsum);
    for (size_t i = 0; i < n; ++i) {
        bsum += (int32_t)(hrow[i] & 1) - (int32_t)((hrow[i] >> 1) & 1);
        if (skipped < padding) skipped++;
        else out.push_back(bsum);
    }
    for (size_t jj = w; jj-- > 0;) {
        for (int j = 1; j <= 64; ++j) {
            const int32_t val = bsum - v_suffix(v[2 * jj], v[2 * jj + 1], j) + v_suffix(v0[2 * jj], v0[2 * jj + 1], j);
            if (skipped < padding) skipped++;
            else out.push_back(val);
        }
        bsum -= v_value(v[2 * jj], v[2 * jj + 1]);
        bsum += v_value(v0[2 * jj], v0[2 * jj + 1]);
    }
    if (out.size() != plen + tlen + 1) {
        set_error("pa_search: internal length mismatch");
        return PA_E_INTERNAL;
    }
    return 0;
}

}  // namespace

extern "C" int pa_search(const uint8_t* pattern, size_t plen, const uint8_t* text, size_t tlen, float unmatched_cost,
                         int32_t* out) {
    std::vector<int32_t> o;
    std::vector<uint64_t> prof, v0;
    if (const int rc = search_out(pattern, plen, text, tlen, unmatched_cost, o, prof, v0)) return rc;
    std::memcpy(out, o.data(), o.size() * sizeof(int32_t));
    return 0;
}

// SearchResult::trace(idx), search.rs:104-228.
extern "C" int pa_search_trace(const uint8_t* pattern, size_t plen, const uint8_t* text, size_t tlen, float unmatched_cost, size_t idx,
                               char** cigar_out, int32_t** path_out, size_t* npos_out) {
    if (cigar_out) *cigar_out = nullptr;
    if (path_out) *path_out = nullptr;
    if (npos_out) *npos_out = 0;
    std::vector<int32_t> out;
    std::vector<uint64_t> prof, v0;
    if (const int rc = search_out(pattern, plen, text, tlen, unmatched_cost, out, prof, v0)) return rc;
    const size_t w = (plen + 63) / 64;
    if (idx >= out.size() || w == 0) {
        set_error("pa_search_trace: idx out of range");
        return PA_E_ARG;
    }
    // idx_to_pos, search.rs:105-115
    int64_t pi, pj;
    if (idx <= tlen) {
        pi = (int64_t)idx;
        pj = (int64_t)plen;
    } else {
        pi = (int64_t)tlen;
        pj = (int64_t)plen - ((int64_t)idx - (int64_t)tlen);
    }
    int32_t target = out[idx];
    if ((size_t)pi == tlen) target -= vec_value_from(v0.data(), w, pj);
    // re-fill text[start..end) x pattern, doubling the width until the cost at `pos` is reproduced (search.rs:132-177)
    size_t width = 2 * plen, start = 0;
    const size_t end = (size_t)pi;
    std::vector<uint64_t> values, first;
    for (;;) {
        start = end > width ? end - width : 0;
        first = start == 0 ? v0 : std::vector<uint64_t>();
        if (start != 0) {
            first.assign(2 * w, 0);
            for (size_t k = 0; k < w; ++k) first[2 * k] = ~0ull;
        }
        std::vector<uint64_t> v(first);
        std::vector<uint8_t> hrow;
        if (const int rc = search_rect(text + start, end - start, prof, w, v, hrow, &values)) return rc;
        const int32_t cost = vec_value_to(v.data(), pj);
        if (cost < target) {
            set_error("pa_search_trace: found a path cheaper than the target cost");
            return PA_E_INTERNAL;
        }
        if (cost == target) break;
        if (start == 0) {
            set_error("pa_search_trace: the full text does not reproduce the target cost");
            return PA_E_INTERNAL;
        }
        width *= 2;
    }
    auto column = [&](int64_t i) -> const uint64_t* {  // fill[i - start]
        return (size_t)i == start ? first.data() : values.data() + ((size_t)i - start - 1) * w * 2;
    };
    auto cost_at = [&](int64_t i, int64_t j) { return vec_value_to(column(i), j); };
    auto tcode = [&](int64_t i) {  // CC order A C T G (profile.rs:23)
        switch (text[i]) {
            case 'a': case 'A': return 0;
            case 'c': case 'C': return 1;
            case 't': case 'T': return 2;
            default: return 3;
        }
    };
    engine::Cigar cigar;
    std::vector<int32_t> path{(int32_t)pi, (int32_t)pj};
    int32_t g = target;
    while (pi > (int64_t)start && pj > 0) {  // search.rs:185-224
        engine::I cnt = 0;
        while (pi > (int64_t)start && pj > 0 && ((prof[4 * ((pj - 1) / 64) + tcode(pi - 1)] >> ((pj - 1) % 64)) & 1)) {
            ++cnt;
            --pi;
            --pj;
            path.push_back((int32_t)pi);
            path.push_back((int32_t)pj);
        }
        if (cnt > 0) {
            cigar.push_elem(engine::CigarElem{engine::CigarOp::Match, cnt});
            continue;
        }
        if (cost_at(pi - 1, pj) == g - 1) {
            --g;
            --pi;
            cigar.push_elem(engine::CigarElem{engine::CigarOp::Del, 1});
        } else if (cost_at(pi, pj - 1) == g - 1) {
            --g;
            --pj;
            cigar.push_elem(engine::CigarElem{engine::CigarOp::Ins, 1});
        } else if (cost_at(pi - 1, pj - 1) == g - 1) {
            --g;
            --pi;
            --pj;
            cigar.push_elem(engine::CigarElem{engine::CigarOp::Sub, 1});
        } else {
            set_error("pa_search_trace: bad trace, stuck at (%lld, %lld)", (long long)pi, (long long)pj);
            return PA_E_INTERNAL;
        }
        path.push_back((int32_t)pi);
        path.push_back((int32_t)pj);
    }
    if (!(pi == 0 || g == 0)) {
        set_error("pa_search_trace: trace ended inside the text with cost left");
        return PA_E_INTERNAL;
    }
    cigar.reverse();
    const std::string text_cigar = cigar.to_string();
    const size_t np = path.size() / 2;
    if (cigar_out) {
        *cigar_out = (char*)std::malloc(text_cigar.size() + 1);
        if (!*cigar_out) {
            set_error("out of memory");
            return PA_E_NOMEM;
        }
        std::memcpy(*cigar_out, text_cigar.c_str(), text_cigar.size() + 1);
    }
    if (path_out) {
        *path_out = (int32_t*)std::malloc(std::max<size_t>(np, 1) * 2 * sizeof(int32_t));
        if (!*path_out) {
            if (cigar_out) {  // nothing half-delivered: the caller owns outputs only on success
                std::free(*cigar_out);
                *cigar_out = nullptr;
            }
            set_error("out of memory");
            return PA_E_NOMEM;
        }
        for (size_t k = 0; k < np; ++k) {
            (*path_out)[2 * k] = path[2 * (np - 1 - k)];
            (*path_out)[2 * k + 1] = path[2 * (np - 1 - k) + 1];
        }
    }
    if (npos_out) *npos_out = np;
    return 0;
}

// ---- batched full DP ----------------------------------------------------------------------------

struct pa_batch {
    struct ReleaseScope {  // FIRST member = destroyed last: ends the scope the destructor's body opens (one device wait for all the buffers)
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~ReleaseScope() {
            release_scope_end();
            if (getenv("PA_ALIGN_PROFILE"))
                std::fprintf(stderr, "[pa_batch_destroy] buffers released %.3f ms after the batch was created\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
    } release_scope_;
    size_t pairs = 0;
    std::vector<size_t> n, m, a_off, b_off, code_off, prof_off, gran_off;
    DeviceBuf d_a, d_b, d_codes, d_prof, d_v, d_gran, d_jobs, d_sums, d_misc, d_desc, d_wavelog;
    size_t max_n = 0, max_m = 0;
    std::vector<StripJob> jobs;
    std::vector<int> last_job;  // per pair (or -1 when w == 0)
    size_t total_gran = 0;
    bool gran_dirty = true;  // the hand-off granules must be cleared before the next pass
    int k = 1;  // 32-row subwords per lane of this batch's strips
    bool sequential = false;  // one wavefront per pair (pair_kernel) instead of chained strips
    int block_waves = 1;
    DeviceBuf d_first;  // sequential: first job of every pair (+ end)
    // big cost-only batches: groups of 32 pairs, bit-sliced (slice_kernel.hpp); the strips of `jobs` are not planned then
    slice::Plan* sliced = nullptr;
    // banded mode (pa_batch_create_banded): per-pair cost threshold of the diagonal band that was planned
    bool banded = false;
    std::vector<int32_t> band_t;
    size_t band_retries = 0;  // pairs re-run with a wider band (summed over passes)
    DeviceBuf d_rjobs, d_rfirst;  // retry sub-batches
    // traceback mode (pa_batch_create_trace / pa_batch_align)
    bool trace = false;
    int dt_max_g = 0, dt_fr_drop = 0;  // DT-trace options of the batched traceback (0: re-fill only)
    size_t trace_fallbacks = 0;  // pairs whose traceback was redone by the host engine
    std::vector<size_t> ckpt_off, cigar_off, word_off;  // per pair, in u32 (ckpt) / elements (cigar) / words of b before this pair
    DeviceBuf d_scratch_gran;
    DeviceBuf d_ckpt, d_cigar, d_cigar_len, d_costs, d_scratch_v, d_scratch_vals, d_tjobs, d_cig_src_off, d_packed;
    hipEvent_t ev2 = nullptr;
    uint8_t* h_text = nullptr;  // pinned host buffer for the packed CIGAR text of one chunk
    size_t h_text_size = 0;
    // pa_batch_align_view: the texts stay in h_text (one chunk) and the caller gets pointers + lengths; strings that come from elsewhere
    // (the host engine, the second round, the small-batch route, several chunks) are malloc'ed as usual and owned by the plan
    bool view_mode = false;
    std::vector<uint32_t> view_len;
    std::vector<char*> view_owned;
    bool in_text(const char* q) const { return h_text && (const uint8_t*)q >= h_text && (const uint8_t*)q < h_text + h_text_size; }
    void free_view_owned() {
        for (char* q : view_owned) std::free(q);
        view_owned.clear();
    }
    // pa_batch_align can work in CHUNKS of the (heaviest-first) order, each on a stream of its own: forward pass (batched A*PA2),
    // traceback, CIGAR text and its copy-out of different chunks overlap (one chunk by default: see the chunk plan in batch_create)
    static constexpr int kMaxChunks = 8;
    std::vector<int32_t> order_host;   // position -> pair (A*PA2: heaviest first; else the identity)
    std::vector<int32_t> torder_host;  // the same chunks with the pairs of a chunk in index order: what the traceback and the text kernels walk
                                       // (neighbouring pairs of the input in one workgroup: C4 traceback 10.0 against 10.9 ms in the forward order)
    DeviceBuf d_torder;
    DeviceBuf d_tlist;     // the traceback's own order: each chunk's pairs by descending cost (trace_order_kernel)
    uint32_t max_nm = 1;   // the longest |a| + |b| of the batch: no cost is larger
    std::vector<size_t> chunk_lo;      // chunk c = positions [chunk_lo[c], chunk_lo[c + 1])
    std::vector<uint64_t> chunk_base;  // byte offset of chunk c's region of d_packed
    hipStream_t cstream[kMaxChunks] = {};
    hipEvent_t ev_pre = nullptr, evF0[kMaxChunks] = {}, evF1[kMaxChunks] = {}, evT1[kMaxChunks] = {};
    DeviceBuf d_cmeta, d_tlen_pos, d_dst_pos;  // d_cmeta: u64 text totals [kMaxChunks], then u32 tickets [kMaxChunks]
    uint8_t* h_meta = nullptr;                  // pinned: u64 totals [kMaxChunks], u32 tlen [pairs], u64 dst [pairs]
    size_t h_meta_size = 0;
    // A*PA2 mode (pa_batch_create_params): one wavefront runs the whole band search of a pair (apa2_kernel.hpp); d_ckpt is the
    // pairs' column store, the traceback reads the blocks of the successful pass from it
    bool astar = false;
    // the block-column store is band-proportional: slot width per pair in words (sweep_logic.hpp SlotGeom); a pair whose band leaves its
    // window is aligned again with full-height slots (second round of pa_batch_align)
    std::vector<uint32_t> win_words, slot_ratio;
    int window_override = -1;  // -1: the policy below; 0: full columns; > 0: that many words
    size_t window_retries = 0;
    double window_retry_peak_bytes = 0;  // the largest full-height block-column store a second round of this plan held at a time
    pa_astarpa2_params aparams_c{};
    apa2::SearchParams sp{};
    DeviceBuf d_rec, d_results, d_pjobs, d_order, d_tstats, d_sh;
    DeviceBuf d_sketch;  // [pairs] the divergence sketch (sketch_unit.hip), kept so that it is released with the batch's other buffers
    DeviceBuf d_rdv;  // 8 x u64: the rendezvous of half-wave blocks in the last forward pass (strips run fused, served by a partner, alone, withdrawn)
    // ... the whole family (pa_batch_create_params with GCSH / pruning / incremental doubling: apa2_full_kernel.hpp)
    bool astar_full = false;
    apa2::FullParams fsp{};
    DeviceBuf d_fjobs, d_jh, d_hrow, d_mi, d_mj, d_active, d_win, d_win0, d_lrec, d_cell, d_probe;
    size_t full_matches = 0, full_seeds = 0;
    double full_build_ms = 0;  // host time spent on the matches of the heuristic (reporting; 0 when the GPU finds them)
    // the matches found on the GPU (gcsh_build_kernel.hpp), inside every pa_batch_align / pa_batch_run
    bool device_build = false;
    DeviceBuf d_bjobs, d_bscratch, d_bstatus, d_bticket;
    hipEvent_t evB0 = nullptr, evB1 = nullptr;
    std::vector<pa_astarpa2_stats> pair_stats;  // of the last pa_batch_align
    double apa2_strip_instr = 0;  // modelled VALU instructions of the DP strips of the last pa_batch_align (reporting)
    double cells = 0, word_updates = 0, algo_bytes = 0;
    hipStream_t stream = nullptr;
    // INVARIANT (round 6, replaces a flag nothing ever set): everything that reads or writes this batch's buffers is queued on `stream` or on
    // one of cstream[] -- never on the null stream or a stream of another object.  The destructor relies on it: it waits for these streams
    // only and hands the buffers to the cache, where another thread may take them at once.  PA_POISON_ALLOC runs keep it honest.
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ~pa_batch() {
        static const bool prof = getenv("PA_ALIGN_PROFILE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        if (prof) {  // (diagnostics: which of the batch's streams is still busy)
            std::fprintf(stderr, "[pa_batch_destroy] busy: batch stream %d", stream && hipStreamQuery(stream) == hipErrorNotReady);
            for (int c = 0; c < kMaxChunks; ++c)
                if (cstream[c]) std::fprintf(stderr, " chunk%d %d", c, hipStreamQuery(cstream[c]) == hipErrorNotReady);
            std::fprintf(stderr, "\n");
            if (stream) (void)hipStreamSynchronize(stream);
            std::fprintf(stderr, "[pa_batch_destroy] batch stream wait %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        // everything that reads or writes this batch's buffers was queued on its own streams: wait for those, not for the device
        bool waited = true;
        if (stream) waited = hipStreamSynchronize(stream) == hipSuccess && waited;
        for (int c = 0; c < kMaxChunks; ++c)
            if (cstream[c]) waited = hipStreamSynchronize(cstream[c]) == hipSuccess && waited;
        if (waited) release_scope_begin_waited();
        else release_scope_begin();  // (a stream whose wait failed: wait for the whole device before the buffers go anywhere)
        if (prof) std::fprintf(stderr, "[pa_batch_destroy] device wait %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ev2) (void)hipEventDestroy(ev2);
        free_view_owned();
        pinned_give(h_text, h_text_size);
        pinned_give(h_meta, h_meta_size);
        if (ev_pre) (void)hipEventDestroy(ev_pre);
        if (evB0) (void)hipEventDestroy(evB0);
        if (evB1) (void)hipEventDestroy(evB1);
        for (int c = 0; c < kMaxChunks; ++c) {
            if (evF0[c]) (void)hipEventDestroy(evF0[c]);
            if (evF1[c]) (void)hipEventDestroy(evF1[c]);
            if (evT1[c]) (void)hipEventDestroy(evT1[c]);
            if (waited) stream_give(cstream[c], d_a.device);
            else if (cstream[c]) (void)hipStreamDestroy(cstream[c]);  // a stream whose synchronize failed is not pooled
        }
        if (prof) std::fprintf(stderr, "[pa_batch_destroy] events, streams, pinned %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (waited) bstream_give(stream, d_a.device);  // (the batch waited for its streams above: nothing of it is queued on the stream any more)
        else if (stream) (void)hipStreamDestroy(stream);
        slice::destroy(sliced);
    }
};

// Shape of a cost-only batch.  Measured on MI355X (profiles/r01_runs/k_sweep*.log, chain_probe2.log; ns per strip step):
//  * more 32-row subwords per lane (k) = fewer VALU instructions per DP cell (the kernel's bound) but slower steps and
//    more padding in the last strip of a pair;
//  * chained strips of one pair advance at the pace of the most crowded SIMD they touch, so a chained batch costs
//    about (wavefronts per SIMD + 1) saturated steps per column, or one lone step when every strip has its own SIMD;
//  * one wavefront can instead run a whole pair, strip after strip (sequential mode): no coupling at all, the best
//    shape as soon as there is about one pair per SIMD.
// The estimates below only rank the six candidates.  PA_STRIP_K=1|2|4 and PA_BATCH_MODE=seq|chain restrict them.
struct BatchShape {
    int k = 1;
    bool sequential = false;
    int block_waves = 1;
    double est_ns = -1;  // the estimate that ranked it (ns of one pass; < 0: nothing to compute)
};
static BatchShape choose_batch_shape(const size_t* a_len, const size_t* b_len, size_t pairs) {
    // ns per strip step (measured, profiles/r02_runs): a wavefront alone on its SIMD; one of W fairly served wavefronts of a SIMD
    // (rotating issue priority + paced top strips, per wavefront and per W); chained strips queueing beyond residency
    static const double kLone[4] = {52.9, 76.5, 121.0, 200.0}, kFair[4] = {40.0, 66.0, 82.0, 138.0}, kSatChain[4] = {50.8, 65.0, 100.0, 150.0};
    static const double kShare[4] = {1.0, 0.85, 0.80, 0.78};  // per-wavefront step cost at 1, 2, 3, >= 4 wavefronts per SIMD
    static const int kK[4] = {1, 2, 4, 8};
    const double simds = (double)(g_device_props_cus > 0 ? g_device_props_cus : 256) * 4.0;
    int env_k = 0, env_mode = 0;
    if (const char* e = getenv("PA_STRIP_K")) {
        const int k = atoi(e);
        if (k == 1 || k == 2 || k == 4 || k == 8) env_k = k;
    }
    if (const char* e = getenv("PA_BATCH_MODE")) env_mode = !strcmp(e, "seq") ? 2 : (!strcmp(e, "chain") ? 1 : 0);
    if (getenv("PA_STRIP_K") && atoi(getenv("PA_STRIP_K")) == 16) {  // experiment (profiles/README.md round 3): one wavefront per pair, 16 subwords per lane
        BatchShape sh16;
        sh16.k = 16;
        sh16.sequential = true;
        sh16.block_waves = kStripBlockWaves;
        return sh16;
    }
    BatchShape best_shape;
    double best = -1;
    for (int t = 0; t < 4; ++t) {
        const int k = kK[t];
        if (env_k && k != env_k) continue;
        // colsteps = sum over strips of their columns; the sequential figures count a short tail strip as the fraction
        // of a tall step it costs
        double strips = 0, live = 0, colsteps = 0, seq_colsteps = 0, seq_longest = 0;
        for (size_t i = 0; i < pairs; ++i) {
            if (a_len[i] == 0 || b_len[i] == 0) continue;
            const int w = (int)((b_len[i] + 63) / 64);
            const double S = (double)strip_plan(w, k, false).strips();
            const StripPlan sq = strip_plan(w, k, true);
            const double Sq = (double)sq.full + (double)sq.tail1 * kLone[0] / kLone[t];
            strips += S;
            live += 1;
            colsteps += S * (double)a_len[i];
            seq_colsteps += Sq * (double)a_len[i];
            seq_longest = std::max(seq_longest, Sq * (double)a_len[i]);
        }
        if (live == 0) return best_shape;
        if (env_mode != 2) {  // chained strips
            const double avg = strips / simds;
            const double wmax = std::ceil(std::ceil(strips / (simds / 4.0)) / 4.0);  // one workgroup per CU, waves round-robin over its SIMDs
            const double per_col = avg <= 1.0 ? kLone[t] : (strips <= 4.0 * simds ? wmax * kFair[t] : (avg + 1.0) * kSatChain[t]);
            const double cost = per_col * colsteps / strips;
            if (best < 0 || cost < best) {
                best = cost;
                best_shape.k = k;
                best_shape.sequential = false;
                // The waves of ONE workgroup are spread round-robin over the four SIMDs of its CU; separate workgroups are not
                // (PA_STRIP_WAVELOG: 1792 single-wave workgroups leave 8 SIMDs with three wavefronts, and every chain that
                // touches one runs at a third of a SIMD).  So: one workgroup per CU, as tall as the batch needs.
                best_shape.block_waves = (strips <= simds || strips > 4.0 * simds) ? kStripBlockWaves : (int)std::ceil(strips / (simds / 4.0));
            }
        }
        if (env_mode != 1) {  // one wavefront per pair
            const double W = std::max(1.0, std::ceil(live / simds));
            const double share = kShare[(int)std::min(W, 4.0) - 1];
            // W pairs share every busy SIMD for as long as the longest of them runs; huge batches stream and balance
            const double cost = std::max(std::min(W, 7.0) * seq_longest, seq_colsteps / std::min(live, simds)) * kLone[t] * share;
            if (best < 0 || cost < best) {
                best = cost;
                best_shape.k = k;
                best_shape.sequential = true;
                best_shape.block_waves = kStripBlockWaves;
            }
        }
    }
    best_shape.est_ns = best;
    return best_shape;
}

// ---- banded sequential pairs (Ukkonen band, the reference's GapGap domain: astarpa2/src/domain.rs:97-116) ----------------
// With a cost threshold t the optimal path of a pair whose distance is <= t stays on the diagonals x = i - j with
// |x| + |(n - m) - x| <= t, i.e. x in [(d - t) / 2, (d + t) / 2], d = n - m.  A strip of rows [R0, R1) therefore only needs
// the columns [R0 + xlo, R1 + xhi): everything left of them enters as +1 deltas (V::one on the left edge, H::one on the
// part of the top row the strip above did not reach), which can only over-estimate.  If the resulting cost is <= t it is
// exact (the optimal path never left the computed cells); otherwise the pair is re-run with a wider band.
static void plan_banded_pair(pa_batch* p, size_t i, int32_t t, std::vector<StripJob>& jobs) {
    const int n = (int)p->n[i], m = (int)p->m[i], w = (m + 63) / 64;
    if (n == 0 || w == 0) return;
    const long d = (long)n - (long)m;
    if ((long)t < std::labs(d)) t = (int32_t)std::labs(d);
    const long xlo = (d - t) / 2 - 1, xhi = (d + t + 1) / 2 + 1;  // one diagonal of slack on either side
    const StripPlan sp = strip_plan(w, p->k, p->sequential);
    const int S = sp.strips(), wps = kWordsPerStrip * p->k;
    const size_t G = (size_t)n / 32 + 2;  // granules of one bottom row, indexed by absolute column / 32
    uint64_t* rows = p->d_gran.as<uint64_t>() + p->gran_off[i];
    const bool pingpong = p->sequential;  // one wavefront per pair reuses two rows; chained strips get a row per boundary
    int word = 0, prev_c0 = 0, prev_c1 = 0;
    for (int s = 0; s < S; ++s) {
        const bool tall = s < sp.full;
        const int words = std::min(tall ? wps : kWordsPerStrip, w - word);
        const long R0 = 64L * word, R1 = 64L * (word + words);
        long c0 = std::max(0L, R0 + xlo) & ~31L, c1 = std::min<long>(n, (std::max(0L, R1 + xhi) + 31) & ~31L);
        if (s + 1 == S) c1 = n;
        if (s == 0) c0 = 0;
        c0 = std::min<long>(c0, std::max(0, prev_c1 - 32) & ~31);  // never leave a gap to the strip above ...
        c0 = std::max<long>(c0, prev_c0);                          // ... and never read granules it did not write
        if (c1 < prev_c1) c1 = prev_c1;
        if (c1 <= c0) c1 = std::min<long>(n, c0 + 32);
        StripJob j;
        std::memset(&j, 0, sizeof j);
        j.k = tall ? p->k : 1;
        j.a_codes = p->d_codes.as<uint32_t>() + p->code_off[i];
        j.b_prof = p->d_prof.as<uint32_t>() + p->prof_off[i] * 4;
        j.v = p->d_v.as<uint32_t>() + p->prof_off[i] * 4;
        j.n = (int)(c1 - c0);
        j.col0 = (int)c0;
        j.word0 = word;
        j.nlanes = 2 * words;
        j.flags = kJobVInitOne;
        if (!p->sequential && !getenv("PA_STRIP_NO_ROTATE")) j.flags |= kJobRotatePrio;  // chained strips share SIMDs: see strip_kernel.hpp
        j.tail_rows = m;
        j.exact_tail = 1;  // the bottom row feeds the strip below
        if (s > 0) {
            j.hin_gran = rows + (size_t)(pingpong ? ((s - 1) & 1) : (s - 1)) * G + (size_t)(c0 / 32);
            j.hin_n = std::max(32, prev_c1 - (int)c0);
        }
        if (s + 1 < S) j.hout_gran = rows + (size_t)(pingpong ? (s & 1) : s) * G + (size_t)(c0 / 32);
        else j.exact_tail = 0;
        j.vsum_out = p->d_sums.as<int32_t>() + i;
        jobs.push_back(j);
        prev_c0 = (int)c0;
        prev_c1 = (int)c1;
        word += words;
    }
}

// Shape of a banded batch.  With about one pair per SIMD one wavefront runs a whole pair (no coupling); fewer pairs run
// as chained strips, where a pair's time is set by the columns along the diagonal (n steps of the step latency) and
// low strips keep that latency low.  Strip height: minimise (strips x (strip rows + band width)) x step cost.
static void choose_band_shape(pa_batch* p) {
    static const double kLone[4] = {52.9, 76.5, 103.0, 178.0};  // (k = 4, 8: eq words from LDS, 50 / 90 instead of 59 / 107 instructions)
    static const int kK[4] = {1, 2, 4, 8};
    const double simds = (double)(g_device_props_cus > 0 ? g_device_props_cus : 256) * 4.0;
    size_t live = 0;
    for (size_t i = 0; i < p->pairs; ++i) live += (p->n[i] > 0 && p->m[i] > 0) ? 1 : 0;
    p->sequential = (double)live >= simds;
    if (const char* e = getenv("PA_BATCH_MODE")) {
        if (!strcmp(e, "seq")) p->sequential = true;
        if (!strcmp(e, "chain")) p->sequential = false;
    }
    int env_k = 0;
    if (const char* e = getenv("PA_STRIP_K")) {
        const int k = atoi(e);
        if (k == 1 || k == 2 || k == 4 || k == 8) env_k = k;
    }
    int best_k = 1;
    double best = -1;
    for (int t = 0; t < 4; ++t) {
        if (env_k && kK[t] != env_k) continue;
        double work = 0, strips = 0, longest = 0;
        for (size_t i = 0; i < p->pairs; ++i) {
            if (p->n[i] == 0 || p->m[i] == 0) continue;
            const double w = (double)((p->m[i] + 63) / 64), rows = 2048.0 * kK[t];
            const double S = std::ceil(w * 64.0 / rows);
            work += S * (std::min(rows, w * 64.0) + (double)p->band_t[i] + 128.0);
            strips += S;
            longest = std::max(longest, (double)p->n[i]);
        }
        // sequential: all the work, shared by the SIMDs; chained: the longest pair's diagonal, or the shared work
        double cost = work / std::min(std::max((double)live, 1.0), simds) * kLone[t];
        if (!p->sequential) cost = std::max(longest * kLone[t], work / simds * kLone[t]);
        if (best < 0 || cost < best) {
            best = cost;
            best_k = kK[t];
            // (chained banded strips mostly wait for the diagonal to reach them: single-wavefront workgroups, which the
            //  dispatcher places wherever a slot frees up, beat any grouping -- 10 Mbp pair: 0.60 s against 0.83-1.0 s)
            p->block_waves = (p->sequential || strips <= simds) ? kStripBlockWaves : 1;
        }
    }
    p->k = best_k;
}

// ---- A*PA2 for many pairs: one wavefront per pair runs the whole band search (apa2_logic.hpp / apa2_kernel.hpp) -----------
static bool apa2_supported(const engine::AstarPa2Params& p) {
    using namespace engine;
    return p.domain == DomainKind::Astar && (p.heuristic == HeuristicKind::None || p.heuristic == HeuristicKind::Gap || p.heuristic == HeuristicKind::SH) &&
           p.block_width == sweep::kBlockW && p.front.sparse && !p.front.incremental_doubling && !p.prune &&
           (p.doubling == DoublingKind::BandDoubling || p.doubling == DoublingKind::LinearSearch) &&
           (!p.front.dt_trace || (p.front.max_g >= 1 && p.front.max_g <= kDtMaxG));
}

// ... and what apa2_full_kernel.hpp takes on top: GCSH, pruning, incremental doubling -- every Domain::Astar parameter set over sparse
// 256-column blocks with a search around it (AstarPa2Params::full() among them).
static bool apa2_full_supported(const engine::AstarPa2Params& p) {
    using namespace engine;
    return p.domain == DomainKind::Astar && p.block_width == sweep::kBlockW && p.front.sparse &&
           (p.doubling == DoublingKind::BandDoubling || p.doubling == DoublingKind::LinearSearch) &&
           (!p.front.dt_trace || (p.front.max_g >= 1 && p.front.max_g <= kDtMaxG));
}

// The start order of the batched band search: the most expensive pairs first (sketch_unit.hip has the why).  Expected work of a pair:
// its length times the band its final pass needs, band ~ estimated cost = e (n + m) / 2 with e from the sketch ((1 - e)^16 = found / 64).
// PA_APA2_ORDER_INPUT keeps the caller's order, PA_APA2_ORDER_LENGTH the order of the lengths (round 4) -- experiments and tests.
static bool astar_start_order(pa_batch* p, std::vector<int32_t>& order) {
    const size_t P = p->pairs;
    order.resize(P);
    for (size_t i = 0; i < P; ++i) order[i] = (int32_t)i;
    if (getenv("PA_APA2_ORDER_INPUT") || P < 2) return true;
    static const bool cprof = getenv("PA_ALIGN_PROFILE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    static_assert(sizeof(apa2::SketchDesc) == sizeof(PairDesc), "the sketch reads the batch's pair descriptors");
    std::vector<uint8_t> found(P, 64);
    const bool sketch = !getenv("PA_APA2_ORDER_LENGTH");
    if (sketch) {
        if (!p->d_sketch.alloc(P) ||
            !hip_ok(apa2::launch_sketch_kernel(p->stream, p->d_a.as<uint8_t>(), p->d_b.as<uint8_t>(), (const apa2::SketchDesc*)p->d_desc.ptr, (int)P, p->d_sketch.as<uint8_t>()),
                    "sketch_kernel launch") ||
            !hip_ok(hipMemcpyAsync(found.data(), p->d_sketch.ptr, P, hipMemcpyDeviceToHost, p->stream), "D2H sketch") || !hip_ok(hipStreamSynchronize(p->stream), "sync"))
            return false;
    }
    const auto t1 = std::chrono::steady_clock::now();
    // e from found / 64 = (1 - e)^16, by table (nothing found: as if half a sample had been)
    double e_of[65];
    for (int f = 0; f <= 64; ++f) e_of[f] = 1.0 - std::pow(std::max(0.5, (double)f) / 64.0, 1.0 / 16.0);
    // descending by the expected work, ties in the caller's order: one sort of 64-bit words (float bits of a positive key order like integers)
    std::vector<uint64_t> keyed(P);
    for (size_t i = 0; i < P; ++i) {
        const float len = (float)(p->n[i] + p->m[i]);
        const float key = sketch ? len * ((float)e_of[std::min<int>(found[i], 64)] * len * 0.5f + 128.0f) : len;
        uint32_t bits;
        std::memcpy(&bits, &key, 4);
        keyed[i] = ((uint64_t)bits << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)i);
    }
    std::sort(keyed.begin(), keyed.end(), std::greater<uint64_t>());
    for (size_t i = 0; i < P; ++i) order[i] = (int32_t)(0xFFFFFFFFu - (uint32_t)(keyed[i] & 0xFFFFFFFFu));
    if (cprof)
        std::fprintf(stderr, "[pa_batch_create]   start order: sketch %.3f ms, sort %.3f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    return true;
}

// The per-pair descriptors of the A*PA2 mode; completes the trace jobs (banded blocks, statistics).
static bool astar_jobs(pa_batch* p, const uint8_t* const* a, const uint8_t* const* b, std::vector<TraceJob>& tjobs) {
    const engine::AstarPa2Params ap = engine::params_from_c(p->aparams_c);
    const size_t P = p->pairs;
    p->sp.heur = ap.heuristic == engine::HeuristicKind::Gap ? sweep::kHeurGap : (ap.heuristic == engine::HeuristicKind::SH ? sweep::kHeurSH : sweep::kHeurNone);
    p->sp.sparse_h = ap.sparse_h ? 1 : 0;
    p->sp.doubling = ap.doubling == engine::DoublingKind::LinearSearch ? apa2::kDoublingLinear : apa2::kDoublingBand;
    p->sp.start = (int32_t)ap.start;
    p->sp.factor = ap.factor;
    p->sp.delta = (int32_t)ap.delta;
    std::vector<size_t> rec_off(P), sh_off(P);
    size_t tr = 0, tsh = 0;
    for (size_t i = 0; i < P; ++i) {
        rec_off[i] = tr;
        tr += (p->n[i] + 255) / 256 + 2;
        sh_off[i] = tsh;
        if (p->sp.heur == sweep::kHeurSH) tsh += p->n[i] + 1;
    }
    if (!p->d_rec.alloc(std::max<size_t>(tr, 1) * sizeof(sweep::BlockRec)) || !p->d_results.alloc(std::max<size_t>(P, 1) * sizeof(apa2::PairResult)) ||
        !p->d_pjobs.alloc(std::max<size_t>(P, 1) * sizeof(apa2::PairJob)) || !p->d_order.alloc(std::max<size_t>(P, 1) * 4) ||
        !p->d_tstats.alloc(std::max<size_t>(P, 1) * 32) || !p->d_sh.alloc(std::max<size_t>(tsh, 1) * 4))
        return false;
    if (p->sp.heur == sweep::kHeurSH && tsh) {  // SeedHeuristicH (pa-heuristic sh.rs:47-106): host-built per-column table
        std::vector<int32_t> sh(tsh);
        for (size_t i = 0; i < P; ++i) {
            engine::SeedHeuristicH h(a[i], (engine::I)p->n[i], b[i], (engine::I)p->m[i], ap.heuristic_k, (int)ap.heuristic_p);
            std::copy(h.h_by_i.begin(), h.h_by_i.end(), sh.begin() + sh_off[i]);
        }
        if (!hip_ok(hipMemcpy(p->d_sh.ptr, sh.data(), tsh * 4, hipMemcpyHostToDevice), "H2D sh")) return false;
    }
    std::vector<apa2::PairJob> pj(P);
    std::vector<int32_t> order(P);
    for (size_t i = 0; i < P; ++i) {
        const size_t w = (p->m[i] + 63) / 64, nblk = (p->n[i] + 255) / 256;
        apa2::PairJob& j = pj[i];
        j.a_codes = p->d_codes.as<uint32_t>() + p->code_off[i];
        j.b_prof = p->d_prof.as<uint32_t>() + p->prof_off[i] * 4;
        j.rec = p->d_rec.as<sweep::BlockRec>() + rec_off[i];
        j.col = p->d_ckpt.as<uint32_t>() + p->ckpt_off[i];
        j.col_stride = (int64_t)p->win_words[i];
        j.slot_ratio = p->slot_ratio[i];
        j.pad0 = 0;
        j.sh_h = p->sp.heur == sweep::kHeurSH ? p->d_sh.as<int32_t>() + sh_off[i] : nullptr;
        j.gran = p->d_scratch_gran.as<uint64_t>() + i * 16;
        j.sum = p->d_sums.as<int32_t>() + i;
        j.result = p->d_results.as<apa2::PairResult>() + i;
        j.n = (int32_t)p->n[i];
        j.m = (int32_t)p->m[i];
        TraceJob& t = tjobs[i];
        t.rec = j.rec;
        t.res = j.result;
        t.tstats = p->d_tstats.as<uint32_t>() + 8 * i;
        t.final_v = p->d_ckpt.as<uint32_t>() + p->ckpt_off[i] + nblk * (size_t)p->win_words[i] * 4;  // (unused: banded blocks go through the slots)
        t.win = (int32_t)p->win_words[i];
        t.slot_ratio = p->slot_ratio[i];
        (void)w;
        order[i] = (int32_t)i;
    }
    if (!astar_start_order(p, order)) return false;  // the most expensive pairs first
    p->order_host = order;
    if (P && (!hip_ok(hipMemcpy(p->d_pjobs.ptr, pj.data(), P * sizeof(apa2::PairJob), hipMemcpyHostToDevice), "H2D pair jobs") ||
              !hip_ok(hipMemcpy(p->d_order.ptr, order.data(), P * 4, hipMemcpyHostToDevice), "H2D order")))
        return false;
    return true;
}

// Words per slot of a pair's block-column store.  Measured on the CPU-kernel engine (round 4): `simple` on 100 kbp at 5 % ends with bands of
// 133 words within 69 words of the main diagonal, 10 kbp at 15 % with 37 within 20; `full` (GCSH) on 100 kbp at 5 % with 12 within 7.
// The windows below hold those with room to spare; what does not fit (15 % on 100 kbp: 260 words) is aligned again with full columns.
static size_t window_words(size_t n, size_t m, bool gcsh, int override_) {
    const size_t wtot = std::max<size_t>((m + 63) / 64, 1);
    static const int env = getenv("PA_APA2_WINDOW") ? atoi(getenv("PA_APA2_WINDOW")) : -1;
    const int o = override_ >= 0 ? override_ : env;
    if (o == 0) return wtot;
    size_t W = o > 0 ? (size_t)o : (gcsh ? 64 : ((2 * ((std::max(n, m) + 1249) / 1250) + 32 + 7) & ~size_t(7)));
    return std::min(W, wtot);
}

// Host threads for per-pair host work of a batch (the matches of GCSH, the SH tables): as many as the process may run on.
static unsigned host_threads() {
    static const unsigned n = [] {
        if (const char* e = getenv("PA_HOST_THREADS")) return (unsigned)std::max(1, atoi(e));
        unsigned c = 0;
#if defined(__linux__)
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) c = (unsigned)CPU_COUNT(&set);
#endif
        if (c == 0) c = std::thread::hardware_concurrency();
        return std::max(1u, std::min(c, 64u));
    }();
    return n;
}
template <class F>
static void parallel_pairs(size_t P, F&& f) {
    const unsigned nt = (unsigned)std::min<size_t>(host_threads(), std::max<size_t>(P, 1));
    if (nt <= 1) {
        for (size_t i = 0; i < P; ++i) f(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&] {
            for (size_t i = next.fetch_add(1); i < P; i = next.fetch_add(1)) f(i);
        });
    for (auto& t : th) t.join();
}

// The per-pair descriptors of the whole-family mode (apa2_full_kernel.hpp); completes the trace jobs like astar_jobs.
// The matches of GCSH (seeds, exact k-mer matches in the reference's push order, the transform filter, local pruning p:
// csrc/gcsh.hpp) are found on host threads; the contours are derived on the device.
static bool astar_full_jobs(pa_batch* p, const uint8_t* const* a, const uint8_t* const* b, std::vector<TraceJob>& tjobs) {
    const engine::AstarPa2Params ap = engine::params_from_c(p->aparams_c);
    const size_t P = p->pairs;
    static const bool cprof = getenv("PA_ALIGN_PROFILE") != nullptr;  // diagnostics: where the creation time goes
    auto cnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double c_mark = cnow();
    auto cmark = [&](const char* what) {
        if (!cprof) return;
        const double t = cnow();
        std::fprintf(stderr, "[pa_batch_create]   full: %-22s %8.3f ms\n", what, t - c_mark);
        c_mark = t;
    };
    p->sp.heur = ap.heuristic == engine::HeuristicKind::Gap ? sweep::kHeurGap : (ap.heuristic == engine::HeuristicKind::SH ? sweep::kHeurSH : sweep::kHeurNone);
    p->sp.sparse_h = ap.sparse_h ? 1 : 0;
    p->sp.doubling = ap.doubling == engine::DoublingKind::LinearSearch ? apa2::kDoublingLinear : apa2::kDoublingBand;
    p->sp.start = (int32_t)ap.start;
    p->sp.factor = ap.factor;
    p->sp.delta = (int32_t)ap.delta;
    p->fsp.sparse_h = ap.sparse_h ? 1 : 0;
    p->fsp.prune = ap.prune ? 1 : 0;
    p->fsp.incremental = ap.front.incremental_doubling ? 1 : 0;
    p->fsp.doubling = ap.doubling == engine::DoublingKind::LinearSearch ? 2 : 1;
    p->fsp.start = (int32_t)ap.start;
    p->fsp.factor = ap.factor;
    p->fsp.delta = (int32_t)ap.delta;
    const bool gcsh = ap.heuristic == engine::HeuristicKind::GCSH, sh = ap.heuristic == engine::HeuristicKind::SH;
    const int32_t hk = ap.heuristic_k < 1 ? 1 : ap.heuristic_k;
    std::vector<size_t> rec_off(P), sh_off(P), col_off(P), seed_off(P), match_off(P + 1, 0);
    size_t tr = 0, tsh = 0, tn = 0, tseeds = 0;
    for (size_t i = 0; i < P; ++i) {
        rec_off[i] = tr;
        tr += (p->n[i] + 255) / 256 + 2;
        sh_off[i] = tsh;
        if (sh) tsh += p->n[i] + 1;
        col_off[i] = tn;
        tn += (p->n[i] + 63) & ~size_t(63);
        seed_off[i] = tseeds;
        if (gcsh) tseeds += p->n[i] >= (size_t)hk ? (p->n[i] - hk) / hk + 1 : 0;
    }
    // The matches of GCSH are found on the GPU, once, at the end of this function (gcsh_build_kernel.hpp), when the look-ahead of local
    // pruning fits its LDS arrays; PA_GCSH_HOST_BUILD=1 finds them on host threads at creation instead (tests compare the two).
    static const bool host_build_env = getenv("PA_GCSH_HOST_BUILD") != nullptr && getenv("PA_GCSH_HOST_BUILD")[0] != '0';
    p->device_build = gcsh && !host_build_env && ap.heuristic_p >= 0 && ap.heuristic_p <= apa2::kBuildMaxP && hk <= 31;
    // ---- host threads: the matches (GCSH, unless the GPU finds them) / the per-column table (SH) of every pair ----
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::vector<int32_t>> pmi(gcsh ? P : 0), pmj(gcsh ? P : 0);
    std::vector<apa2::GcshSeedWindow> win(p->device_build ? 0 : tseeds);
    std::vector<int32_t> shv(tsh);
    std::atomic<bool> bad_base{false};
    if ((gcsh && !p->device_build) || sh)
        parallel_pairs(P, [&](size_t i) {
            const engine::I n = (engine::I)p->n[i], m = (engine::I)p->m[i];
            if (n == 0 || m == 0) return;
            if (sh) {
                engine::SeedHeuristicH h(a[i], n, b[i], m, ap.heuristic_k, (int)ap.heuristic_p);
                std::copy(h.h_by_i.begin(), h.h_by_i.end(), shv.begin() + (long)sh_off[i]);
                return;
            }
            engine::GcshHeuristic gh(a[i], n, b[i], m, ap.heuristic_k, (int)ap.heuristic_p, ap.prune, false);
            pmi[i].reserve(gh.by_start.size());
            pmj[i].reserve(gh.by_start.size());
            for (const auto& mt : gh.by_start) {
                pmi[i].push_back(mt.i);
                pmj[i].push_back(mt.j);
            }
            for (size_t s = 0; s < gh.active_range.size(); ++s)
                win[seed_off[i] + s] = apa2::GcshSeedWindow{(int32_t)gh.active_range[s].b0, (int32_t)gh.active_range[s].b1, -1, 0};
        });
    size_t tm = 0;
    std::vector<size_t> cap(P, 0), tsz(P, 0);
    size_t ttab = 0;
    for (size_t i = 0; i < P; ++i) {
        match_off[i] = tm;
        if (gcsh && !p->device_build) tm += pmi[i].size();
        if (p->device_build) {
            // room for the candidates of a pair: every seed once and half of them again, plus 2048 (a pair that needs more -- a
            // repeat-rich sequence -- is flagged by the kernel and goes to the host engine)
            const size_t ns = p->n[i] >= (size_t)hk ? (p->n[i] - hk) / hk + 1 : 0;
            cap[i] = ns + ns / 2 + 2048;
            size_t t2 = 64;
            while (t2 < 2 * ns + 1) t2 *= 2;
            tsz[i] = t2;
            ttab += t2;
            tm += cap[i];
        }
    }
    match_off[P] = tm;
    p->full_matches = tm;
    p->full_seeds = tseeds;
    p->full_build_ms = p->device_build ? 0.0 : std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    cmark("host tables / sizes");
    if (!p->d_rec.alloc(std::max<size_t>(tr, 1) * sizeof(sweep::BlockRec)) || !p->d_jh.alloc(std::max<size_t>(tr, 1) * 4) ||
        !p->d_results.alloc(std::max<size_t>(P, 1) * sizeof(apa2::PairResult)) || !p->d_fjobs.alloc(std::max<size_t>(P, 1) * sizeof(apa2::FullJob)) ||
        !p->d_order.alloc(std::max<size_t>(P, 1) * 4) || !p->d_tstats.alloc(std::max<size_t>(P, 1) * 32) || !p->d_sh.alloc(std::max<size_t>(tsh, 1) * 4) ||
        !p->d_hrow.alloc(std::max<size_t>(tn, 64)) || !p->d_mi.alloc(std::max<size_t>(tm, 1) * 4) || !p->d_mj.alloc(std::max<size_t>(tm, 1) * 4) ||
        !p->d_active.alloc(std::max<size_t>(tm, 64)) || !p->d_win.alloc(std::max<size_t>(tseeds, 1) * sizeof(apa2::GcshSeedWindow)) ||
        !p->d_win0.alloc(std::max<size_t>(tseeds, 1) * sizeof(apa2::GcshSeedWindow)) ||
        !p->d_lrec.alloc((tm + 2 * std::max<size_t>(P, 1)) * sizeof(apa2::GcshCell)) || !p->d_cell.alloc(std::max<size_t>(tm, 1) * sizeof(apa2::GcshCell)) ||
        !p->d_probe.alloc(128 + 8 * std::max<size_t>(P, 1)))  // (16 counters, then per pair: HW_ID / XCC_ID and the ticks of its band search; PA_APA2_PROBE_STATS)
        return false;
    cmark("device buffers");
    if (tsh && !hip_ok(hipMemcpy(p->d_sh.ptr, shv.data(), tsh * 4, hipMemcpyHostToDevice), "H2D sh")) return false;
    if (tm && !p->device_build) {  // (the GPU's builder writes d_mi / d_mj itself: nothing to upload -- until round 4 this sent 2 x 4 tm bytes of zeros)
        std::vector<int32_t> mi(tm), mj(tm);
        for (size_t i = 0; i < P; ++i) {
            std::copy(pmi[i].begin(), pmi[i].end(), mi.begin() + (long)match_off[i]);
            std::copy(pmj[i].begin(), pmj[i].end(), mj.begin() + (long)match_off[i]);
        }
        if (!hip_ok(hipMemcpy(p->d_mi.ptr, mi.data(), tm * 4, hipMemcpyHostToDevice), "H2D matches") ||
            !hip_ok(hipMemcpy(p->d_mj.ptr, mj.data(), tm * 4, hipMemcpyHostToDevice), "H2D matches"))
            return false;
    }
    if (tseeds && !p->device_build && !hip_ok(hipMemcpy(p->d_win0.ptr, win.data(), tseeds * sizeof(apa2::GcshSeedWindow), hipMemcpyHostToDevice), "H2D seed windows")) return false;
    std::vector<apa2::GcshBuildJob> bj(p->device_build ? P : 0);
    if (p->device_build) {
        // scratch of the build kernel, one slice per pair: u32 keys / next_same / cnt / fill per seed, the table, five ints and two bytes
        // per candidate slot
        const size_t words = 4 * tseeds + P + ttab + 4 * tm, bytes = words * 4 + 2 * tm + 64;
        if (!p->d_bscratch.alloc(bytes) || !p->d_bjobs.alloc(std::max<size_t>(P, 1) * sizeof(apa2::GcshBuildJob)) || !p->d_bstatus.alloc(std::max<size_t>(P, 1) * 4) ||
            !p->d_bticket.alloc(64) || !hip_ok(hipEventCreate(&p->evB0), "event") || !hip_ok(hipEventCreate(&p->evB1), "event"))
            return false;
        cmark("  build: buffers");
        int32_t* w32 = p->d_bscratch.as<int32_t>();
        uint8_t* w8 = (uint8_t*)(w32 + words);
        size_t o32 = 0, o8 = 0;
        for (size_t i = 0; i < P; ++i) {
            const size_t ns = p->n[i] >= (size_t)hk ? (p->n[i] - hk) / hk + 1 : 0;
            apa2::GcshBuildJob& x = bj[i];
            std::memset(&x, 0, sizeof x);
            x.a = p->d_a.as<uint8_t>() + p->a_off[i];
            x.b = p->d_b.as<uint8_t>() + p->b_off[i];
            x.keys = (uint32_t*)(w32 + o32);
            o32 += ns;
            x.next_same = w32 + o32;
            o32 += ns;
            x.cnt = w32 + o32;
            o32 += ns + 1;
            x.fill = w32 + o32;
            o32 += ns;
            x.slot = w32 + o32;
            o32 += tsz[i];
            x.tmp_s = w32 + o32;
            o32 += cap[i];
            x.tmp_j = w32 + o32;
            o32 += cap[i];
            x.gpos = w32 + o32;
            o32 += cap[i];
            x.cj = w32 + o32;
            o32 += cap[i];
            x.flag = w8 + o8;
            o8 += cap[i];
            x.keptg = w8 + o8;
            o8 += cap[i];
            x.mi = p->d_mi.as<int32_t>() + match_off[i];
            x.mj = p->d_mj.as<int32_t>() + match_off[i];
            x.win0 = p->d_win0.as<apa2::GcshSeedWindow>() + seed_off[i];
            x.nmatch_out = &p->d_fjobs.as<apa2::FullJob>()[i].g.nmatch;
            x.status = p->d_bstatus.as<uint32_t>() + i;
            x.n = (int32_t)p->n[i];
            x.m = (int32_t)p->m[i];
            x.k = hk;
            x.p = (int32_t)ap.heuristic_p;
            x.nseeds = (int32_t)ns;
            x.tsize = (int32_t)tsz[i];
            x.cap = (int32_t)cap[i];
        }
        cmark("  build: descriptors");
        if (P && !hip_ok(hipMemcpy(p->d_bjobs.ptr, bj.data(), P * sizeof(apa2::GcshBuildJob), hipMemcpyHostToDevice), "H2D build jobs")) return false;
        cmark("  build: H2D");
    }
    cmark("build scratch + jobs");
    const bool launch_build = p->device_build && P;
    std::vector<apa2::FullJob> fj(P);
    std::vector<int32_t> order(P);
    for (size_t i = 0; i < P; ++i) {
        const size_t w = (p->m[i] + 63) / 64, nblk = (p->n[i] + 255) / 256;
        apa2::FullJob& j = fj[i];
        std::memset(&j, 0, sizeof j);
        j.a_codes = p->d_codes.as<uint32_t>() + p->code_off[i];
        j.b_prof = p->d_prof.as<uint32_t>() + p->prof_off[i] * 4;
        j.rec = p->d_rec.as<sweep::BlockRec>() + rec_off[i];
        j.jh = p->d_jh.as<int32_t>() + rec_off[i];
        j.col = p->d_ckpt.as<uint32_t>() + p->ckpt_off[i];
        j.col_stride = (int64_t)p->win_words[i];
        j.slot_ratio = p->slot_ratio[i];
        j.hrow = p->d_hrow.as<uint8_t>() + col_off[i];
        j.sh_h = sh ? p->d_sh.as<int32_t>() + sh_off[i] : nullptr;
        j.gran = p->d_scratch_gran.as<uint64_t>() + i * 16;
        j.sum = p->d_sums.as<int32_t>() + i;
        j.result = p->d_results.as<apa2::PairResult>() + i;
        j.n = (int32_t)p->n[i];
        j.m = (int32_t)p->m[i];
        j.heur = (int32_t)ap.heuristic;
        if (gcsh) {
            apa2::GcshDev& g = j.g;
            g.mi = p->d_mi.as<int32_t>() + match_off[i];
            g.mj = p->d_mj.as<int32_t>() + match_off[i];
            g.active = p->d_active.as<uint8_t>() + match_off[i];
            g.win = p->d_win.as<apa2::GcshSeedWindow>() + seed_off[i];
            g.lrec = p->d_lrec.as<apa2::GcshCell>() + match_off[i] + 2 * i;
            g.cell = p->d_cell.as<apa2::GcshCell>() + match_off[i];
            g.nmatch = (int32_t)(match_off[i + 1] - match_off[i]);
            g.nlayers = 1;
            g.n = j.n;
            g.m = j.m;
            g.k = hk;
            g.nseeds = j.n >= hk ? (j.n - hk) / hk + 1 : 0;
            g.prune = ap.prune ? 1 : 0;
        }
        TraceJob& t = tjobs[i];
        t.rec = j.rec;
        t.res = j.result;
        t.tstats = p->d_tstats.as<uint32_t>() + 8 * i;
        t.final_v = p->d_ckpt.as<uint32_t>() + p->ckpt_off[i] + nblk * (size_t)p->win_words[i] * 4;  // (unused: banded blocks go through the slots)
        t.win = (int32_t)p->win_words[i];
        t.slot_ratio = p->slot_ratio[i];
        (void)w;
        order[i] = (int32_t)i;
    }
    if (!astar_start_order(p, order)) return false;  // the most expensive pairs first
    p->order_host = order;
    if (P && (!hip_ok(hipMemcpy(p->d_fjobs.ptr, fj.data(), P * sizeof(apa2::FullJob), hipMemcpyHostToDevice), "H2D pair jobs") ||
              !hip_ok(hipMemcpy(p->d_order.ptr, order.data(), P * 4, hipMemcpyHostToDevice), "H2D order")))
        return false;
    cmark("pair jobs + order");
    if (launch_build) {
        // The matches of GCSH are part of the batch like the sequences they are derived from: found here, once, by the GPU (one wavefront
        // per pair, 12.8 KB of LDS each: twelve to a CU), on the batch's stream -- the first alignment call queues behind it.
        const int cus = g_device_props_cus > 0 ? g_device_props_cus : 256;
        static const int per_cu = getenv("PA_BUILD_WAVES_PER_CU") ? std::max(1, atoi(getenv("PA_BUILD_WAVES_PER_CU"))) : 12;
        const int grid = (int)std::min<size_t>(P, (size_t)cus * (size_t)per_cu);
        if (!hip_ok(hipMemsetAsync(p->d_bticket.ptr, 0, 64, p->stream), "memset") || !hip_ok(hipEventRecord(p->evB0, p->stream), "event")) return false;
        if (!hip_ok(apa2::launch_gcsh_build_kernel(grid, p->stream, p->d_bjobs.as<apa2::GcshBuildJob>(), (int)P, p->d_bticket.as<uint32_t>()), "gcsh_build_kernel launch") || !hip_ok(hipEventRecord(p->evB1, p->stream), "event")) return false;
    }
    return true;
}

static pa_batch* batch_create(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len, size_t pairs,
                              bool trace, float band_hint = -1.f, int dt_max_g = 0, int dt_fr_drop = 0, const pa_astarpa2_params* astar = nullptr,
                              int window_override = -1) {
    if (!ensure_device()) return nullptr;
    static const bool cprof = getenv("PA_ALIGN_PROFILE") != nullptr;  // diagnostics: where the creation time goes
    auto cnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double c_mark = cnow();
    auto cmark = [&](const char* what) {
        if (!cprof) return;
        const double t = cnow();
        std::fprintf(stderr, "[pa_batch_create] %-28s %8.3f ms\n", what, t - c_mark);
        c_mark = t;
    };
    auto p = std::make_unique<pa_batch>();
    p->pairs = pairs;
    p->trace = trace;
    int slice_rows = 0;  // > 0: the batch runs bit-sliced with that many rows per lane (slice_plan.hpp)
    if (astar) {
        p->astar = true;
        p->aparams_c = *astar;
        p->astar_full = !apa2_supported(engine::params_from_c(*astar));  // GCSH / pruning / incremental doubling: apa2_full_kernel.hpp
        p->window_override = window_override;
        const bool gcsh = engine::params_from_c(*astar).heuristic == engine::HeuristicKind::GCSH;
        for (size_t i = 0; i < pairs; ++i) {
            p->win_words.push_back((uint32_t)window_words(a_len[i], b_len[i], gcsh, window_override));
            p->slot_ratio.push_back(a_len[i] ? (uint32_t)std::min<uint64_t>(((uint64_t)b_len[i] << 20) / (uint64_t)a_len[i], 0xFFFFFFFFull) : 0u);
        }
    }
    p->dt_max_g = dt_max_g;
    p->dt_fr_drop = dt_fr_drop;
    p->banded = band_hint >= 0.f;
    if (p->banded) {
        for (size_t i = 0; i < pairs; ++i) {
            p->n.push_back(a_len[i]);
            p->m.push_back(b_len[i]);
            const double len = (double)std::max(a_len[i], b_len[i]);
            const long d = std::labs((long)a_len[i] - (long)b_len[i]);
            p->band_t.push_back((int32_t)std::min<double>(d + std::ceil(band_hint * len) + 32, (double)a_len[i] + (double)b_len[i] + 64));
        }
        choose_band_shape(p.get());
        p->n.clear();
        p->m.clear();
    } else {
        const BatchShape sh = astar ? BatchShape() : choose_batch_shape(a_len, b_len, pairs);
        p->k = sh.k;
        p->sequential = sh.sequential;
        p->block_waves = sh.block_waves;
        if (!astar && !trace) {  // a cost-only batch big enough for groups of 32 pairs: the bit-sliced kernel, when its estimate is the lower one
            double est = -1;
            const double simds = (double)(g_device_props_cus > 0 ? g_device_props_cus : 256) * 4.0;
            const int R = slice::choose_rows_per_lane(a_len, b_len, pairs, simds, &est);
            const bool forced = getenv("PA_SLICE") && atoi(getenv("PA_SLICE")) > 0;
            if (R > 0 && (forced || sh.est_ns < 0 || est < sh.est_ns)) slice_rows = R;
        }
    }
    size_t ta = 0, tb = 0, tc = 0, tp = 0, tg = 0;
    for (size_t i = 0; i < pairs; ++i) {
        if (a_len[i] > (size_t)(1u << 30) || b_len[i] > (size_t)(1u << 30)) {
            set_error("sequence too long");
            return nullptr;
        }
        p->n.push_back(a_len[i]);
        p->m.push_back(b_len[i]);
        p->a_off.push_back(ta);
        p->b_off.push_back(tb);
        p->code_off.push_back(tc);
        p->prof_off.push_back(tp);
        p->gran_off.push_back(tg);
        const size_t w = (b_len[i] + 63) / 64;
        ta += (a_len[i] + 15) & ~size_t(15);
        tb += (b_len[i] + 15) & ~size_t(15);
        tc += (a_len[i] + 15) / 16;
        tp += w;
        tg += (astar || slice_rows) ? 0
              : p->banded ? (size_t)(p->sequential ? 2 : std::max(1, strip_plan((int)w, p->k, false).strips() - 1)) * (a_len[i] / 32 + 2)
                          : rect_granules((int)a_len[i], (int)w, p->k, p->sequential);
        p->cells += (double)a_len[i] * (double)b_len[i];
        p->word_updates += (double)a_len[i] * (double)w;
        // algorithmic HBM bytes, cost-only rectangle: 0.75 B/column + 48 B/word
        p->algo_bytes += 0.75 * (double)a_len[i] + 48.0 * (double)w;
    }
    p->total_gran = tg;
    if (trace) {
        size_t tck = 0, tcg = 0, tw = 0;
        for (size_t i = 0; i < pairs; ++i) {
            const size_t w = (b_len[i] + 63) / 64;
            p->ckpt_off.push_back(tck);
            p->cigar_off.push_back(tcg);
            p->word_off.push_back(tw);
            tw += std::min<size_t>(std::max<size_t>(w, 1), (size_t)kTraceScratchWords);
            // u32: one V column per 256 columns of a (slot 0 unused); A*PA2 mode: slots 0 .. ceil(n / 256)
            tck += astar ? ((a_len[i] + 255) / 256 + 1) * (size_t)p->win_words[i] * 4 : (a_len[i] / 256 + 1) * w * 4;
            tcg += a_len[i] + b_len[i] + 2;
        }
        if (tcg >= (size_t(1) << 62) || !p->d_ckpt.alloc(tck * 4) || !p->d_cigar.alloc(tcg * 4) || !p->d_packed.alloc(tcg) ||
            !p->d_tlen_pos.alloc(std::max<size_t>(pairs * 4, 16)) || !p->d_dst_pos.alloc(std::max<size_t>(pairs * 8, 16)) || !p->d_cmeta.alloc(256) ||
            !p->d_cigar_len.alloc(std::max<size_t>(pairs * 4, 16)) || !p->d_costs.alloc(std::max<size_t>(pairs * 4, 16)) ||
            // re-fill scratch: 256 columns x min(w, kTraceScratchWords) words of V per pair
            !p->d_scratch_v.alloc(std::max<size_t>(tw, 1) * 16) || !p->d_scratch_vals.alloc(std::max<size_t>(tw, 1) * 256 * 16) ||
            !p->d_scratch_gran.alloc(std::max<size_t>(pairs, 1) * 16 * 8) ||
            !p->d_tjobs.alloc(std::max<size_t>(pairs, 1) * sizeof(TraceJob)) || !p->d_cig_src_off.alloc(std::max<size_t>(pairs, 1) * 8))
            return nullptr;
    }
    if (!p->d_a.alloc(ta) || !p->d_b.alloc(tb) || !p->d_codes.alloc(tc * 4) || !p->d_prof.alloc(tp * 16) ||
        !p->d_v.alloc(tp * 16) || !p->d_gran.alloc(tg * 8) || !p->d_sums.alloc(std::max<size_t>(pairs * 4, 16)) || !p->d_misc.alloc(32))
        return nullptr;
    cmark("host layout + hipMalloc");
    if (!(p->stream = bstream_take()) || !hip_ok(hipEventCreate(&p->ev0), "event") ||
        !hip_ok(hipEventCreate(&p->ev1), "event") || !hip_ok(hipEventCreate(&p->ev2), "event"))
        return nullptr;
    cmark("stream + events");
    // Upload: the sequences are gathered into the device layout through two pinned staging buffers, so that the copy of one
    // chunk overlaps the gathering of the next and runs at link speed (a pageable H2D of 800 MB costs 5x as much).
    {
        const size_t kChunk = size_t(32) << 20;
        // the two pinned buffers are kept for the life of the process (pinning 64 MB costs more than uploading 200 MB);
        // one creation at a time uses them
        static std::mutex stage_mutex;
        static uint8_t* stage_cache[2] = {nullptr, nullptr};
        std::lock_guard<std::mutex> stage_lock(stage_mutex);
        uint8_t* stage[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        bool ok = true;
        for (int k = 0; k < 2 && ok; ++k) {
            if (!stage_cache[k]) {
                void* hp = nullptr;
                ok = hip_ok(hipHostMalloc(&hp, kChunk, hipHostMallocDefault), "hipHostMalloc(upload staging)");
                stage_cache[k] = (uint8_t*)hp;
            }
            ok = ok && hip_ok(hipEventCreate(&done[k]), "event");
            stage[k] = stage_cache[k];
        }
        bool used[2] = {false, false};  // a staging buffer is reused only after its previous copy has finished
        int buf = 0;
        auto upload = [&](uint8_t* dev, const std::vector<size_t>& off, const uint8_t* const* src, const size_t* len, size_t total) {
            // walk the device image [0, total) in chunks; every chunk is assembled from the pairs that intersect it
            size_t pair = 0;
            for (size_t base = 0; base < total && ok; base += kChunk, buf ^= 1) {
                const size_t end = std::min(total, base + kChunk);
                if (used[buf]) ok = hip_ok(hipEventSynchronize(done[buf]), "event sync");
                while (pair < pairs && off[pair] + len[pair] <= base) ++pair;
                // the pieces of this chunk: (pair, first byte, end, end of the piece before) -- only the padding between two sequences
                // needs zeroing.  A big chunk is gathered by several threads (round 5: one thread copies 12-16 GB/s, less than the link
                // takes; the C4 batch's 200 MB: 12.5 ms of its 16 ms creation)
                struct Piece {
                    size_t q, lo, hi, prev;
                };
                std::vector<Piece> pieces;
                size_t cur = base;
                for (size_t q = pair; q < pairs && off[q] < end; ++q) {
                    const size_t lo = std::max(off[q], base), hi = std::min(off[q] + len[q], end);
                    if (lo >= hi) continue;
                    pieces.push_back(Piece{q, lo, hi, cur});
                    cur = hi;
                }
                uint8_t* const dst = stage[buf];
                auto gather = [&, dst, base](size_t p0, size_t p1) {
                    for (size_t t = p0; t < p1; ++t) {
                        const Piece& pc = pieces[t];
                        if (pc.lo > pc.prev) std::memset(dst + (pc.prev - base), 0, pc.lo - pc.prev);
                        std::memcpy(dst + (pc.lo - base), src[pc.q] + (pc.lo - off[pc.q]), pc.hi - pc.lo);
                    }
                };
                const size_t nthreads = (end - base >= (size_t(8) << 20) && pieces.size() >= 8) ? std::min<size_t>(4, host_threads()) : 1;
                if (nthreads > 1) {
                    std::vector<std::thread> th;
                    for (size_t t = 1; t < nthreads; ++t) th.emplace_back(gather, pieces.size() * t / nthreads, pieces.size() * (t + 1) / nthreads);
                    gather(0, pieces.size() / nthreads);
                    for (auto& x : th) x.join();
                } else {
                    gather(0, pieces.size());
                }
                if (end > cur) std::memset(stage[buf] + (cur - base), 0, end - cur);
                ok = ok && hip_ok(hipMemcpyAsync(dev + base, stage[buf], end - base, hipMemcpyHostToDevice, p->stream), "H2D sequences") &&
                     hip_ok(hipEventRecord(done[buf], p->stream), "event");
                used[buf] = true;
            }
        };
        if (ok) upload(p->d_a.as<uint8_t>(), p->a_off, a, a_len, ta);
        if (ok) upload(p->d_b.as<uint8_t>(), p->b_off, b, b_len, tb);
        ok = ok && hip_ok(hipStreamSynchronize(p->stream), "sync");
        for (int k = 0; k < 2; ++k)
            if (done[k]) (void)hipEventDestroy(done[k]);
        if (!ok) return nullptr;
    }
    cmark("upload of the sequences");
    // Jobs: pair-major, strips of a pair consecutive (ticket order == dependency order).
    p->last_job.assign(pairs, -1);
    std::vector<int32_t> first(pairs + 1, 0);
    for (size_t i = 0; i < pairs; ++i) {
        first[i] = (int32_t)p->jobs.size();
        first[i + 1] = first[i];
        const int w = (int)((b_len[i] + 63) / 64);
        if (w == 0 || a_len[i] == 0 || astar || slice_rows) continue;
        if (p->banded) {
            plan_banded_pair(p.get(), i, p->band_t[i], p->jobs);
            p->last_job[i] = (int)p->jobs.size() - 1;
            first[i + 1] = (int32_t)p->jobs.size();
            continue;
        }
        RectPlan r;
        r.a_codes = p->d_codes.as<uint32_t>() + p->code_off[i];
        r.b_prof = p->d_prof.as<uint32_t>() + p->prof_off[i] * 4;
        r.v = p->d_v.as<uint32_t>() + p->prof_off[i] * 4;
        r.n = (int)a_len[i];
        r.w0 = 0;
        r.w1 = w;
        r.gran = p->d_gran.as<uint64_t>() + p->gran_off[i];
        r.gran_stride = (a_len[i] + 31) / 32;
        r.sum_out = p->d_sums.as<int32_t>() + i;
        r.exact_end = false;
        r.v_init_one = true;
        r.tail_rows = (int)b_len[i];
        r.k = p->k;
        r.pingpong = p->sequential;
        if (trace) {
            r.ckpt = p->d_ckpt.as<uint32_t>() + p->ckpt_off[i];
            r.ckpt_stride = w;
        }
        plan_rect(p->jobs, r);
        p->last_job[i] = (int)p->jobs.size() - 1;
        first[i + 1] = (int32_t)p->jobs.size();
    }
    if (p->sequential) {
        if (!p->d_first.alloc(first.size() * 4)) return nullptr;
        if (!hip_ok(hipMemcpyAsync(p->d_first.ptr, first.data(), first.size() * 4, hipMemcpyHostToDevice, p->stream), "H2D first")) return nullptr;
        if (!hip_ok(hipStreamSynchronize(p->stream), "sync")) return nullptr;  // `first` is a local
    }
    {
        std::vector<PairDesc> desc(pairs);
        for (size_t i = 0; i < pairs; ++i) {
            desc[i] = PairDesc{p->a_off[i], p->b_off[i], p->code_off[i], p->prof_off[i], (int)a_len[i], (int)b_len[i]};
            p->max_n = std::max(p->max_n, a_len[i]);
            p->max_m = std::max(p->max_m, b_len[i]);
        }
        if (!p->d_desc.alloc(pairs * sizeof(PairDesc))) return nullptr;
        if (pairs && !hip_ok(hipMemcpyAsync(p->d_desc.ptr, desc.data(), pairs * sizeof(PairDesc), hipMemcpyHostToDevice, p->stream), "H2D desc"))
            return nullptr;
        if (!hip_ok(hipStreamSynchronize(p->stream), "sync")) return nullptr;  // desc is a local
    }
    if (trace && pairs) {
        std::vector<TraceJob> tjobs(pairs);
        std::vector<uint64_t> src_off(pairs);
        for (size_t i = 0; i < pairs; ++i) {
            TraceJob& t = tjobs[i];
            t.a = p->d_a.as<uint8_t>() + p->a_off[i];
            t.b = p->d_b.as<uint8_t>() + p->b_off[i];
            t.a_codes = p->d_codes.as<uint32_t>() + p->code_off[i];
            t.b_prof = p->d_prof.as<uint32_t>() + p->prof_off[i] * 4;
            t.ckpt = p->d_ckpt.as<uint32_t>() + p->ckpt_off[i];
            t.final_v = p->d_v.as<uint32_t>() + p->prof_off[i] * 4;
            t.sum = p->d_sums.as<int32_t>() + i;
            t.cigar = p->d_cigar.as<uint32_t>() + p->cigar_off[i];
            t.cigar_len = p->d_cigar_len.as<uint32_t>() + i;
            t.cost_out = p->d_costs.as<int32_t>() + i;
            t.scratch_v = p->d_scratch_v.as<uint32_t>() + p->word_off[i] * 4;
            t.scratch_vals = p->d_scratch_vals.as<uint32_t>() + p->word_off[i] * 256 * 4;
            t.scratch_gran = p->d_scratch_gran.as<uint64_t>() + i * 16;
            t.scratch_words = (int32_t)std::min<size_t>(std::max<size_t>((b_len[i] + 63) / 64, 1), (size_t)kTraceScratchWords);
            t.n = (int32_t)a_len[i];
            t.m = (int32_t)b_len[i];
            t.w = (int32_t)((b_len[i] + 63) / 64);
            t.cigar_cap = (uint32_t)std::min<size_t>(a_len[i] + b_len[i] + 2, 0xFFFFFFF0u);
            t.dt_max_g = dt_max_g;
            t.dt_fr_drop = dt_fr_drop;
            t.win = t.w;
            t.slot_ratio = 0;
            src_off[i] = p->cigar_off[i];
        }
        if (astar && !(p->astar_full ? astar_full_jobs(p.get(), a, b, tjobs) : astar_jobs(p.get(), a, b, tjobs))) return nullptr;
        if (!astar) {  // (the plain traced batch: chunks of the pairs as they come)
            p->order_host.resize(pairs);
            for (size_t i = 0; i < pairs; ++i) p->order_host[i] = (int32_t)i;
            if (!p->d_order.alloc(pairs * 4) || !hip_ok(hipMemcpy(p->d_order.ptr, p->order_host.data(), pairs * 4, hipMemcpyHostToDevice), "H2D order")) return nullptr;
        }
        {
            // chunks of pa_batch_align.  ONE by default: measured in round 4 (profiles/README.md), chunks on streams of their own do not
            // shorten the call -- band search and traceback are both bound by instruction issue, so running the traceback of one chunk
            // beside the band search of the next gains nothing (C4: 24.65 against 25.0 ms at the C ABI with four chunks), and chunks too
            // small to fill the chip lose (4096 x 100 kbp in three chunks: 135 against 120 ms).  PA_ALIGN_CHUNKS=n for experiments.
            static const int env_chunks = getenv("PA_ALIGN_CHUNKS") ? atoi(getenv("PA_ALIGN_CHUNKS")) : 0;
            // (Until the traceback started its expensive pairs first, four chunks paid for many SHORT pairs -- C4: 24.1 against 26.0 ms --
            //  by cutting the traceback's tail; with the ordering one chunk is ahead there too: 22.3 against 22.9 ms.)
            int C = 1;
            if (env_chunks > 0) C = std::min<int>(env_chunks, pa_batch::kMaxChunks);
            C = (int)std::max<size_t>(1, std::min<size_t>((size_t)C, pairs));
            p->chunk_lo.assign((size_t)C + 1, 0);
            p->chunk_base.assign((size_t)C + 1, 0);
            for (int c = 0; c <= C; ++c) p->chunk_lo[(size_t)c] = pairs * (size_t)c / (size_t)C;
            uint64_t acc = 0;
            for (int c = 0; c < C; ++c) {
                p->chunk_base[(size_t)c] = acc;
                for (size_t q = p->chunk_lo[(size_t)c]; q < p->chunk_lo[(size_t)c + 1]; ++q) {
                    const size_t i = (size_t)p->order_host[q];
                    acc += a_len[i] + b_len[i] + 2;
                }
            }
            p->chunk_base[(size_t)C] = acc;
            p->torder_host = p->order_host;
            for (int c = 0; c < C; ++c) std::sort(p->torder_host.begin() + (long)p->chunk_lo[(size_t)c], p->torder_host.begin() + (long)p->chunk_lo[(size_t)c + 1]);
            if (!p->d_torder.alloc(std::max<size_t>(pairs, 1) * 4) || !p->d_tlist.alloc(std::max<size_t>(pairs, 1) * 4) ||
                !hip_ok(hipMemcpy(p->d_torder.ptr, p->torder_host.data(), pairs * 4, hipMemcpyHostToDevice), "H2D trace order"))
                return nullptr;
            for (size_t i = 0; i < pairs; ++i) p->max_nm = (uint32_t)std::max<size_t>(p->max_nm, std::min<size_t>(a_len[i] + b_len[i], 0x7FFFFFFFu));
            p->h_meta = (uint8_t*)pinned_take(64 + pairs * 12 + 64, &p->h_meta_size);
            if (!p->h_meta) return nullptr;
            if (!hip_ok(hipEventCreate(&p->ev_pre), "event")) return nullptr;
            for (int c = 0; c < C; ++c)
                if (!(p->cstream[c] = stream_take()) || !hip_ok(hipEventCreate(&p->evF0[c]), "event") ||
                    !hip_ok(hipEventCreate(&p->evF1[c]), "event") || !hip_ok(hipEventCreate(&p->evT1[c]), "event"))
                    return nullptr;
        }
        if (!hip_ok(hipMemsetAsync(p->d_scratch_gran.ptr, 0, pairs * 16 * 8, p->stream), "memset trace granules") ||
            !hip_ok(hipMemcpyAsync(p->d_tjobs.ptr, tjobs.data(), pairs * sizeof(TraceJob), hipMemcpyHostToDevice, p->stream), "H2D trace jobs") ||
            !hip_ok(hipMemcpyAsync(p->d_cig_src_off.ptr, src_off.data(), pairs * 8, hipMemcpyHostToDevice, p->stream), "H2D offsets") ||
            !hip_ok(hipStreamSynchronize(p->stream), "sync"))
            return nullptr;
    }
    if (!p->sequential && !p->trace && !p->banded && !p->jobs.empty()) {
        // Chained batches beyond one wavefront per SIMD: a SIMD serves its OLDEST wavefront first, the younger ones get what is
        // left, and a chain advances at the pace of its most starved strip -- so pairs finish staggered by wave slot and the
        // tail of the launch runs on a mostly idle chip (PA_STRIP_WAVELOG shows it).  The top strip of every pair therefore
        // paces itself against the average progress of all pairs (strip_kernel.hpp kJobPace).
        // Both need every strip resident (at most four wavefronts per SIMD at <= 128 VGPRs); beyond that strips queue in ticket
        // order and only the priority rotation is kept.
        static const bool no_pace = getenv("PA_STRIP_NO_PACE") != nullptr;
        static const bool no_rotate = getenv("PA_STRIP_NO_ROTATE") != nullptr;
        const size_t simds = (size_t)(g_device_props_cus > 0 ? g_device_props_cus : 256) * 4;
        int tops = 0;
        for (const StripJob& j : p->jobs) tops += j.hin_gran == nullptr;
        for (StripJob& j : p->jobs) {
            if (p->jobs.size() > simds && !no_rotate) j.flags |= kJobRotatePrio;
            if (!no_pace && p->jobs.size() > simds && p->jobs.size() <= 4 * simds && j.hin_gran == nullptr && tops > 1) {
                j.flags |= kJobPace;
                j.ckpt = p->d_misc.as<uint32_t>() + 4;  // the u64 progress counter (zeroed with the ticket before every pass)
                j.ckpt_stride = tops;
            }
        }
    }
    if (getenv("PA_STRIP_WAVELOG") && !p->trace && !p->jobs.empty()) {
        // diagnostics: every strip wavefront leaves {HW_ID, XCC_ID, start, end (100 MHz), chunks that had to poll} behind
        if (!p->d_wavelog.alloc(p->jobs.size() * 32)) return nullptr;
        for (size_t j = 0; j < p->jobs.size(); ++j) {
            p->jobs[j].values = p->d_wavelog.as<uint32_t>() + 8 * j;
            p->jobs[j].flags |= kJobLog;
        }
    }
    cmark("jobs + descriptors");
    if (slice_rows) {
        p->sequential = false;
        p->sliced = slice::create(p->n.data(), p->m.data(), pairs, p->code_off.data(), p->prof_off.data(), slice_rows);
        if (!p->sliced) return nullptr;
        cmark("bit-sliced plan");
    }
    if (!p->d_jobs.alloc(p->jobs.size() * sizeof(StripJob))) return nullptr;
    if (!p->jobs.empty() &&
        !hip_ok(hipMemcpyAsync(p->d_jobs.ptr, p->jobs.data(), p->jobs.size() * sizeof(StripJob), hipMemcpyHostToDevice, p->stream), "H2D jobs"))
        return nullptr;
    if (!hip_ok(hipStreamSynchronize(p->stream), "sync")) return nullptr;
    return p.release();
}

extern "C" pa_batch* pa_batch_create(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b,
                                     const size_t* b_len, size_t pairs) {
    return batch_create(a, a_len, b, b_len, pairs, false);
}

extern "C" pa_batch* pa_batch_create_banded(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b,
                                            const size_t* b_len, size_t pairs, float divergence_hint) {
    if (!(divergence_hint >= 0.f)) {
        set_error("pa_batch_create_banded: divergence_hint must be >= 0");
        return nullptr;
    }
    return batch_create(a, a_len, b, b_len, pairs, false, divergence_hint);
}

extern "C" pa_batch* pa_batch_create_trace(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b,
                                           const size_t* b_len, size_t pairs) {
    return batch_create(a, a_len, b, b_len, pairs, true);
}

// ... with the traceback options of `trace_params->front` (dt_trace, max_g, fr_drop): DT-trace through every block first, the
// re-fill only where it gives up (blocks/trace.rs:51-125), e.g. the `simple` preset's { dt_trace: true, max_g: 40, fr_drop: 10 }.
extern "C" pa_batch* pa_batch_create_trace_params(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b,
                                                  const size_t* b_len, size_t pairs, const pa_astarpa2_params* trace_params) {
    if (!trace_params) return batch_create(a, a_len, b, b_len, pairs, true);
    const engine::AstarPa2Params tp = engine::params_from_c(*trace_params);
    if (!tp.front.sparse || tp.block_width != 256) {
        set_error("pa_batch_create_trace_params: the batched traceback walks sparse 256-column blocks");
        return nullptr;
    }
    if (tp.front.dt_trace && (tp.front.max_g < 1 || tp.front.max_g > kDtMaxG)) {
        set_error("pa_batch_create_trace_params: max_g must be in 1..%d", kDtMaxG);
        return nullptr;
    }
    return batch_create(a, a_len, b, b_len, pairs, true, -1.f, tp.front.dt_trace ? (int)tp.front.max_g : 0, tp.front.dt_trace ? (int)tp.front.fr_drop : 0);
}

// 1 if pa_batch_create_params takes these parameters, 0 if they belong to pa_align (or are invalid).
extern "C" int pa_batch_params_supported(const pa_astarpa2_params* params) {
    return params && engine::params_valid(*params) && apa2_full_supported(engine::params_from_c(*params)) ? 1 : 0;
}

// A*PA2 for many pairs (the `simple` preset and its relatives): what a loop over pa_align(a, b, params, trace = 1) returns --
// cost, CIGAR and statistics -- with every pair's whole band search run by one wavefront on the GPU.
extern "C" pa_batch* pa_batch_create_params(const uint8_t* const* a, const size_t* a_len, const uint8_t* const* b, const size_t* b_len,
                                            size_t pairs, const pa_astarpa2_params* params) {
    if (!params || !engine::params_valid(*params)) {
        set_error("pa_batch_create_params: invalid A*PA2 parameters");
        return nullptr;
    }
    const engine::AstarPa2Params ap = engine::params_from_c(*params);
    if (!apa2_full_supported(ap)) {
        set_error("pa_batch_create_params: the batched band search runs Domain::Astar (NoCost / GapCost / SH / GCSH, with or without pruning and "
                  "incremental doubling) over sparse 256-column blocks with band doubling or a linear search -- the `simple` and `full` presets and "
                  "their relatives; use pa_align for other parameters");
        return nullptr;
    }
    return batch_create(a, a_len, b, b_len, pairs, true, -1.f, ap.front.dt_trace ? (int)ap.front.max_g : 0, ap.front.dt_trace ? (int)ap.front.fr_drop : 0, params);
}

// Profiles -> (granule clear) -> DP kernel, all queued on the batch's stream; ev0/ev1 bracket the DP kernel.
// The band-search kernel (apa2_kernel / apa2_full_kernel) for pairs order[lo .. lo + cnt) on stream s; `ticket` is the launch's own
// ticket word (zeroed by the caller).
static int launch_astar(pa_batch* p, hipStream_t s, size_t lo, size_t cnt, uint32_t* ticket, uint32_t* dbg) {
    if (cnt == 0) return 0;
    // a persistent grid: wavefronts pull pairs by ticket; at most kApa2BlocksPerCu blocks of four wavefronts per CU
    // (apa2_full_kernel fits five wavefronts per SIMD, apa2_kernel -- four strip heights, 128 VGPRs -- four)
    const int per_cu = getenv("PA_APA2_BLOCKS_PER_CU") ? std::max(1, atoi(getenv("PA_APA2_BLOCKS_PER_CU"))) : (p->astar_full ? 5 : 4);
    static const bool probe_stats = getenv("PA_APA2_PROBE_STATS") != nullptr;
    const int cus = g_device_props_cus > 0 ? g_device_props_cus : 256;
    const int grid = (int)std::min<size_t>((cnt + kStripBlockWaves - 1) / kStripBlockWaves, (size_t)cus * per_cu);
    const int32_t* ord = p->d_order.as<int32_t>() + lo;
    // Two half-wave blocks of one workgroup run as one strip (strip2_kernel.hpp).  PA_APA2_RDV=0 turns the rendezvous off (every strip alone,
    // as before round 5: same results -- tests compare the two); PA_APA2_RDV_PATIENCE_US: how long a posted block waits for a partner.
    // (read at every launch: tests switch it inside one process)
    const char* rdv_env = getenv("PA_APA2_RDV");
    const double rdv_us = getenv("PA_APA2_RDV_PATIENCE_US") ? std::max(0.0, atof(getenv("PA_APA2_RDV_PATIENCE_US"))) : 20.0;
    // A block that waits for a partner is a wavefront that does nothing: worth it when the SIMDs have other wavefronts to run, not when a
    // batch leaves most of them with one or none (512 x 100 kbp: 38.3 against 36.4 ms).  PA_APA2_RDV=0: never; =2: whatever the batch size.
    const size_t simds = (size_t)(g_device_props_cus > 0 ? g_device_props_cus : 256) * 4;
    RdvParams rp;
    rp.enabled = cnt >= 2 * simds ? 1u : 0u;
    if (rdv_env && rdv_env[0] == '0') rp.enabled = 0u;
    if (rdv_env && rdv_env[0] == '2') rp.enabled = cnt > 1 ? 1u : 0u;
    rp.patience = (uint32_t)(rdv_us * 100.0);  // ticks of the 100 MHz clock
    // (measured on C4, profiles/r05_runs/prio_probe.log: `full` 11.08 -> 10.45 ms, `simple` 8.70 -> 9.12 ms: on for the first only)
    rp.prio = getenv("PA_APA2_PRIO") ? (getenv("PA_APA2_PRIO")[0] == '0' ? 0u : 1u) : (p->astar_full ? 1u : 0u);
    rp.search_windows = (getenv("PA_APA2_SEARCH_WINDOWS") && getenv("PA_APA2_SEARCH_WINDOWS")[0] == '0') ? 0u : 1u;  // (experiments)
    unsigned long long* rdv_stats = p->d_rdv.ptr ? p->d_rdv.as<unsigned long long>() : nullptr;
    const hipError_t e = p->astar_full ? apa2::launch_apa2_full_kernel(grid, s, p->d_fjobs.as<apa2::FullJob>(), ord, (int)cnt, p->fsp, ticket, p->d_misc.as<uint32_t>() + 1, dbg,
                                                                       probe_stats ? p->d_probe.as<unsigned long long>() : nullptr, rp, rdv_stats)
                                       : apa2::launch_apa2_kernel(grid, s, p->d_pjobs.as<apa2::PairJob>(), ord, (int)cnt, p->sp, ticket, p->d_misc.as<uint32_t>() + 1, dbg,
                                                                  getenv("PA_APA2_K1") ? 1 : 0, rp, rdv_stats);
    return hip_ok(e, "apa2_kernel launch") ? 0 : PA_E_HIP;
}

// Profiles -> (granule clear) -> DP kernel, all queued on the batch's stream; ev0/ev1 bracket the DP kernel.
// launch = false (batched A*PA2 through pa_batch_align): everything BEFORE the band-search kernel only; the caller launches it chunk by
// chunk on streams of their own.
static int batch_forward(pa_batch* p, bool launch = true) {
    hipStream_t s = p->stream;
    // (1) profiles (BitProfile::build, once per pair: blocks.rs:112)
    if (!hip_ok(hipMemsetAsync(p->d_misc.ptr, 0, 32, s), "memset")) return PA_E_HIP;  // (+ the pace counter of chained batches)
    if (p->astar && !p->d_rdv.ptr && !p->d_rdv.alloc(64)) return PA_E_HIP;
    if (p->d_rdv.ptr && !hip_ok(hipMemsetAsync(p->d_rdv.ptr, 0, 64, s), "memset rendezvous counters")) return PA_E_HIP;
    for (size_t base = 0; base < p->pairs; base += 32768) {  // gridDim.y limit
        const unsigned ny = (unsigned)std::min<size_t>(32768, p->pairs - base);
        const PairDesc* dd = p->d_desc.as<PairDesc>() + base;
        if (p->max_n) {
            const unsigned nx = (unsigned)(((p->max_n + 15) / 16 + 255) / 256);
            hipLaunchKernelGGL(encode_a_batch_kernel, dim3(nx, ny), dim3(256), 0, s, p->d_a.as<uint8_t>(), p->d_codes.as<uint32_t>(), dd,
                               p->d_misc.as<uint32_t>() + 3);
        }
        if (p->max_m) {
            const unsigned nx = (unsigned)(((p->max_m + 63) / 64 + 3) / 4);
            hipLaunchKernelGGL(build_b_batch_kernel, dim3(nx, ny), dim3(256), 0, s, p->d_b.as<uint8_t>(), p->d_prof.as<uint64_t>(), dd,
                               p->d_misc.as<uint32_t>() + 3);
        }
        if (!hip_ok(hipGetLastError(), "profile kernels")) return PA_E_HIP;
    }
    // (2) clear hand-off granules, (3) strips
    // every strip hands the granules it consumed back zeroed, so the buffer is cleared only before the first pass (and
    // after a pass that did not finish)
    if (p->total_gran && p->gran_dirty && !hip_ok(hipMemsetAsync(p->d_gran.ptr, 0, p->total_gran * 8, s), "memset gran")) return PA_E_HIP;
    p->gran_dirty = true;  // (banded chained strips skip part of every row: it stays dirty, cleared before every pass)
    if (!hip_ok(hipMemsetAsync(p->d_sums.ptr, 0, std::max<size_t>(p->pairs * 4, 16), s), "memset sums")) return PA_E_HIP;
    // d_misc (ticket, err, -, bad-base flag) was zeroed above; the events bracket the strip kernel alone
    if (p->astar && p->astar_full) {
        // a batch can be aligned again: the pruning state starts from scratch (every match active, the windows as built)
        if (!hip_ok(hipMemsetAsync(p->d_active.ptr, 1, std::max<size_t>(p->full_matches, 64), s), "memset active") ||
            (p->full_seeds && !hip_ok(hipMemcpyAsync(p->d_win.ptr, p->d_win0.ptr, p->full_seeds * sizeof(apa2::GcshSeedWindow), hipMemcpyDeviceToDevice, s), "D2D windows")) ||
            !hip_ok(hipMemsetAsync(p->d_probe.ptr, 0, 128, s), "memset probe stats"))
            return PA_E_HIP;
    }
    if (!launch) return 0;
    if (!hip_ok(hipEventRecord(p->ev0, s), "event")) return PA_E_HIP;
    if (p->astar) {
        if (p->pairs) {
            uint32_t* dbg = nullptr;
            if (getenv("PA_APA2_DEBUG")) {
                void* hp = nullptr;
                if (!hip_ok(hipHostMalloc(&hp, 256, hipHostMallocMapped), "hipHostMalloc(debug)")) return PA_E_HIP;
                std::memset(hp, 0, 256);
                dbg = (uint32_t*)hp;
            }
            if (const int rc = launch_astar(p, s, 0, p->pairs, p->d_misc.as<uint32_t>(), dbg)) return rc;
            if (dbg) {  // diagnostics: the forward pass alone, progress markers and first results on stderr
                std::fprintf(stderr, "[apa2] forward launched: pairs %zu\n", p->pairs);
                for (int sec = 0; sec < 8 && hipStreamQuery(s) == hipErrorNotReady; ++sec) {
                    const volatile uint32_t* d = dbg;
                    std::fprintf(stderr, "[apa2] t=%ds stage %u f_max %d tries %u block %u js %d je %d strip %u pair %u\n", sec, d[0], (int)d[1], d[2], d[3], (int)d[4], (int)d[5], d[6], d[7]);
                    std::this_thread::sleep_for(std::chrono::milliseconds(1000));
                }
                if (hipStreamQuery(s) == hipErrorNotReady) {
                    std::fprintf(stderr, "[apa2] the forward kernel does not finish: giving up\n");
                    std::_Exit(3);
                }
                if (!hip_ok(hipStreamSynchronize(s), "sync")) return PA_E_HIP;
                std::vector<apa2::PairResult> r(std::min<size_t>(p->pairs, 8));
                if (!hip_ok(hipMemcpy(r.data(), p->d_results.ptr, r.size() * sizeof(apa2::PairResult), hipMemcpyDeviceToHost), "D2H")) return PA_E_HIP;
                for (size_t i = 0; i < r.size(); ++i)
                    std::fprintf(stderr, "[apa2] pair %zu: status %d cost %d f_max %d tries %u blocks %u lanes %llu last %d len %d\n", i, r[i].status, r[i].cost, r[i].f_max,
                                 r[i].f_max_tries, r[i].num_blocks, (unsigned long long)r[i].computed_lanes, r[i].last_block_idx, r[i].blocks_len);
            }
        }
    } else if (p->sequential) {
        if (!launch_pairs(p->d_jobs.as<StripJob>(), p->d_first.as<int32_t>(), (int)p->pairs, p->d_misc.as<uint32_t>(), s, p->k, p->trace))
            return PA_E_HIP;
    } else if (!launch_strips(p->d_jobs.as<StripJob>(), (int)p->jobs.size(), false, p->d_misc.as<uint32_t>(), s, false, false, p->k,
                              p->block_waves, p->trace)) {
        return PA_E_HIP;
    }
    if (!hip_ok(hipEventRecord(p->ev1, s), "event")) return PA_E_HIP;
    return 0;
}

// Banded pass: cost = n + (sum over the pair's strips of their right-edge vertical deltas).  A cost above the band's
// threshold is only an upper bound: those pairs run again with a wider band (at most up to the bound itself, which is then
// certainly wide enough), and the thresholds that worked are kept for the next pass over the same batch.
static int banded_finish(pa_batch* p, std::vector<int32_t>& sums, int32_t* cost_out, float* kernel_ms) {
    hipStream_t s = p->stream;
    std::vector<size_t> todo;
    for (size_t i = 0; i < p->pairs; ++i) {
        const size_t n = p->n[i], m = p->m[i];
        if (n == 0 || m == 0) {
            cost_out[i] = (int32_t)(n + m);
            continue;
        }
        cost_out[i] = (int32_t)n + sums[i];
        if (cost_out[i] > p->band_t[i]) todo.push_back(i);
    }
    bool replanned = !todo.empty();
    while (!todo.empty()) {
        p->band_retries += todo.size();
        std::vector<StripJob> jobs;
        std::vector<int32_t> first(todo.size() + 1, 0);
        for (size_t k = 0; k < todo.size(); ++k) {
            const size_t i = todo[k];
            p->band_t[i] = (int32_t)std::min<long>((long)cost_out[i], std::max<long>(2L * p->band_t[i], 64));
            first[k] = (int32_t)jobs.size();
            plan_banded_pair(p, i, p->band_t[i], jobs);
            first[k + 1] = (int32_t)jobs.size();
        }
        if (!p->d_rjobs.alloc(jobs.size() * sizeof(StripJob)) || !p->d_rfirst.alloc(first.size() * 4)) return PA_E_HIP;
        hipEvent_t e0 = p->ev0, e1 = p->ev2;
        for (size_t i : todo)
            if (!hip_ok(hipMemsetAsync(p->d_sums.as<int32_t>() + i, 0, 4, s), "memset sum")) return PA_E_HIP;
        uint32_t misc[4] = {0, 0, 0, 0};
        float ms = 0.f;
        if (!hip_ok(hipMemcpyAsync(p->d_rjobs.ptr, jobs.data(), jobs.size() * sizeof(StripJob), hipMemcpyHostToDevice, s), "H2D jobs") ||
            !hip_ok(hipMemcpyAsync(p->d_rfirst.ptr, first.data(), first.size() * 4, hipMemcpyHostToDevice, s), "H2D first") ||
            (!p->sequential && p->total_gran && !hip_ok(hipMemsetAsync(p->d_gran.ptr, 0, p->total_gran * 8, s), "memset gran")) ||
            !hip_ok(hipEventRecord(e0, s), "event") ||
            !(p->sequential ? launch_pairs(p->d_rjobs.as<StripJob>(), p->d_rfirst.as<int32_t>(), (int)todo.size(), p->d_misc.as<uint32_t>(), s, p->k, false)
                            : launch_strips(p->d_rjobs.as<StripJob>(), (int)jobs.size(), false, p->d_misc.as<uint32_t>(), s, true, false, p->k, 1, false)) ||
            !hip_ok(hipEventRecord(e1, s), "event") ||
            !hip_ok(hipMemcpyAsync(sums.data(), p->d_sums.ptr, p->pairs * 4, hipMemcpyDeviceToHost, s), "D2H") ||
            !hip_ok(hipMemcpyAsync(misc, p->d_misc.ptr, 16, hipMemcpyDeviceToHost, s), "D2H") || !hip_ok(hipStreamSynchronize(s), "sync"))
            return PA_E_HIP;
        if (misc[1] != PA_ERR_NONE) {
            set_error("device spin timeout (err=%u)", misc[1]);
            return PA_E_TIMEOUT;
        }
        if (kernel_ms && hip_ok(hipEventElapsedTime(&ms, e0, e1), "elapsed")) *kernel_ms += ms;
        std::vector<size_t> next;
        for (size_t i : todo) {
            cost_out[i] = (int32_t)p->n[i] + sums[i];
            if (cost_out[i] > p->band_t[i]) next.push_back(i);
        }
        todo.swap(next);
    }
    if (replanned) {  // keep what worked: the next pass over this batch starts from these bands
        p->jobs.clear();
        std::vector<int32_t> first(p->pairs + 1, 0);
        for (size_t i = 0; i < p->pairs; ++i) {
            first[i] = (int32_t)p->jobs.size();
            plan_banded_pair(p, i, p->band_t[i], p->jobs);
            first[i + 1] = (int32_t)p->jobs.size();
        }
        if (!p->d_jobs.alloc(p->jobs.size() * sizeof(StripJob)) ||
            !hip_ok(hipMemcpyAsync(p->d_jobs.ptr, p->jobs.data(), p->jobs.size() * sizeof(StripJob), hipMemcpyHostToDevice, s), "H2D jobs") ||
            (p->sequential && !hip_ok(hipMemcpyAsync(p->d_first.ptr, first.data(), first.size() * 4, hipMemcpyHostToDevice, s), "H2D first")) ||
            !hip_ok(hipStreamSynchronize(s), "sync"))
            return PA_E_HIP;
    }
    return 0;
}

extern "C" int pa_batch_run(pa_batch* p, int32_t* cost_out, float* kernel_ms) {
    if (!p) return PA_E_ARG;
    if (p->astar) {  // batched A*PA2, costs only: the band search without the traceback kernels (the distance over the traced band)
        float fwd = 0.f;
        const int rc = pa_batch_align(p, cost_out, nullptr, &fwd, nullptr);
        if (kernel_ms) *kernel_ms = fwd;
        return rc;
    }
    hipStream_t s = p->stream;
    if (p->sliced) {  // groups of 32 pairs, bit-sliced: profiles as always, then slice_unit.hip; d_sums receives the distances themselves
        if (const int rc = batch_forward(p, false)) return rc;
        if (const int rc = slice::run(p->sliced, s, p->d_codes.as<uint32_t>(), p->d_prof.as<uint64_t>(), p->d_sums.as<int32_t>(), p->d_misc.as<uint32_t>() + 4,
                                      p->ev0, p->ev1))
            return rc;
        uint32_t misc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p->pairs && !hip_ok(hipMemcpyAsync(cost_out, p->d_sums.ptr, p->pairs * 4, hipMemcpyDeviceToHost, s), "D2H")) return PA_E_HIP;
        if (!hip_ok(hipMemcpyAsync(misc, p->d_misc.ptr, 32, hipMemcpyDeviceToHost, s), "D2H")) return PA_E_HIP;
        if (!hip_ok(hipStreamSynchronize(s), "sync")) return PA_E_HIP;
        if (misc[3]) {
            set_error("sequence contains a base outside ACGT");
            return PA_E_INVALID_BASE;
        }
        if (misc[5] != 0) {
            set_error("device spin timeout in the bit-sliced kernel (err=%u)", misc[5]);
            return PA_E_TIMEOUT;
        }
        if (kernel_ms && !hip_ok(hipEventElapsedTime(kernel_ms, p->ev0, p->ev1), "elapsed")) return PA_E_HIP;
        for (size_t i = 0; i < p->pairs; ++i)  // (a pair with an empty sequence is in no group)
            if (p->n[i] == 0 || p->m[i] == 0) cost_out[i] = (int32_t)(p->n[i] + p->m[i]);
        return 0;
    }
    if (const int rc = batch_forward(p)) return rc;
    // (4) read back: bottom sums and each pair's last v word (for the rows beyond |b| in the last word)
    std::vector<int32_t> sums(p->pairs, 0);
    uint32_t misc[4] = {0, 0, 0, 0};
    if (p->pairs && !hip_ok(hipMemcpyAsync(sums.data(), p->d_sums.ptr, p->pairs * 4, hipMemcpyDeviceToHost, s), "D2H")) return PA_E_HIP;
    if (!hip_ok(hipMemcpyAsync(misc, p->d_misc.ptr, 16, hipMemcpyDeviceToHost, s), "D2H")) return PA_E_HIP;
    if (!hip_ok(hipStreamSynchronize(s), "sync")) return PA_E_HIP;
    if (misc[3]) {
        set_error("sequence contains a base outside ACGT");
        return PA_E_INVALID_BASE;
    }
    if (misc[1] != PA_ERR_NONE) {
        set_error("device spin timeout (err=%u)", misc[1]);
        return PA_E_TIMEOUT;
    }
    p->gran_dirty = p->banded && !p->sequential;  // clean finish (banded chained strips leave unconsumed granules behind)
    if (p->d_wavelog.ptr) {
        std::vector<uint32_t> log(p->jobs.size() * 8);
        if (hip_ok(hipMemcpy(log.data(), p->d_wavelog.ptr, log.size() * 4, hipMemcpyDeviceToHost), "D2H wavelog")) {
            if (FILE* f = std::fopen(getenv("PA_STRIP_WAVELOG"), "w")) {
                std::fprintf(f, "job k word0 xcc se cu simd wave t0 t1 polled\n");
                for (size_t j = 0; j < p->jobs.size(); ++j) {
                    const uint32_t* r = &log[8 * j];
                    const uint32_t hw = r[0];
                    std::fprintf(f, "%zu %d %d %u %u %u %u %u %llu %llu %u\n", j, p->sequential ? p->jobs[j].k : p->k, p->jobs[j].word0, r[1] & 15u,
                                 (hw >> 13) & 7u, (hw >> 8) & 15u, (hw >> 4) & 3u, hw & 15u, (unsigned long long)(r[2] | ((uint64_t)r[3] << 32)),
                                 (unsigned long long)(r[4] | ((uint64_t)r[5] << 32)), r[6]);
                }
                std::fclose(f);
            }
        }
    }
    if (kernel_ms) {
        *kernel_ms = 0.f;
        if (!p->jobs.empty() && !hip_ok(hipEventElapsedTime(kernel_ms, p->ev0, p->ev1), "elapsed")) return PA_E_HIP;
    }
    if (p->banded) return banded_finish(p, sums, cost_out, kernel_ms);
    for (size_t i = 0; i < p->pairs; ++i) {
        const size_t n = p->n[i], m = p->m[i], w = (m + 63) / 64;
        if (n == 0) { cost_out[i] = (int32_t)m; continue; }
        if (w == 0) { cost_out[i] = (int32_t)n; continue; }
        // bot_val = rounded |b| + sum of bottom deltas (blocks.rs:171,255-267); cost = get(|b|) (domain.rs:520)
        // the strip kernel already removed the rows beyond |b| of the last word (StripJob::tail_rows)
        cost_out[i] = (int32_t)(w * 64) + sums[i];
    }
    return 0;
}

// The parameter set whose traceback pa_batch_align reproduces: nw (Full domain, no doubling) with sparse 256-column
// blocks and no DT-trace (params.rs:46-68 with front.sparse = true).
static pa_astarpa2_params traced_batch_params() {
    engine::AstarPa2Params p = engine::AstarPa2Params::nw();
    p.front.sparse = true;
    p.front.dt_trace = false;
    p.front.incremental_doubling = false;
    pa_astarpa2_params c;
    engine::params_to_c(p, &c);
    return c;
}

extern "C" void pa_params_batch_align(pa_astarpa2_params* out) {
    if (out) *out = traced_batch_params();
}

// A batched A*PA2 of a handful of LONG pairs is one lone wavefront per pair for the whole band search (about 50 ms for 100 kbp at 5 %),
// while the single-pair engine spreads one pair's pass over many wavefronts (14.4 ms; two pairs 29 vs 47 ms, three 44 vs 48): up to kSmallRoutePairs pairs of at least
// kSmallRouteLen bases go through that engine one after another.  Same parameter set, same host logic: cost, CIGAR string and statistics
// are the ones the batch kernels produce (tests/test_gpu_apa2_batch.py compares both routes).  PA_BATCH_SMALL_ROUTE=0 switches it off.
static constexpr size_t kSmallRoutePairs = 2, kSmallRouteLen = 32768;

static bool small_route(const pa_batch* p, const char* const* cigar_out) {
    static const char* env = getenv("PA_BATCH_SMALL_ROUTE");
    if (env && env[0] == '0') return false;
    if (!p->astar || !cigar_out || p->pairs == 0 || p->pairs > kSmallRoutePairs) return false;
    for (size_t i = 0; i < p->pairs; ++i)
        if (std::min(p->n[i], p->m[i]) < kSmallRouteLen) return false;
    return true;
}

static int batch_align_small(pa_batch* p, int32_t* cost_out, char** cigar_out, float* forward_ms, float* trace_ms) {
    const size_t P = p->pairs;
    const auto t0 = std::chrono::steady_clock::now();
    p->pair_stats.assign(P, pa_astarpa2_stats{});
    p->apa2_strip_instr = 0;
    for (size_t i = 0; i < P; ++i) cigar_out[i] = nullptr;
    auto fail_out = [&](int code) {
        for (size_t k = 0; k < P; ++k) {
            std::free(cigar_out[k]);
            cigar_out[k] = nullptr;
        }
        return code;
    };
    // one after another: two sweeps of long pairs at once get in each other's way (measured: 2 pairs 69 ms side by side, 29 ms in a row)
    for (size_t i = 0; i < P; ++i) {
        std::vector<uint8_t> ba(p->n[i]), bb(p->m[i]);
        if (!hip_ok(hipMemcpy(ba.data(), p->d_a.as<uint8_t>() + p->a_off[i], p->n[i], hipMemcpyDeviceToHost), "D2H a") ||
            !hip_ok(hipMemcpy(bb.data(), p->d_b.as<uint8_t>() + p->b_off[i], p->m[i], hipMemcpyDeviceToHost), "D2H b"))
            return fail_out(PA_E_HIP);
        std::string text;
        int32_t c = 0;
        int rc = align_hip(ba.data(), p->n[i], bb.data(), p->m[i], p->aparams_c, true, false, &c, &text, &p->pair_stats[i]);
        if (rc == PA_E_TIMEOUT) rc = align_hip(ba.data(), p->n[i], bb.data(), p->m[i], p->aparams_c, true, false, &c, &text, &p->pair_stats[i]);
        if (rc != 0) return fail_out(rc);
        cost_out[i] = c;
        cigar_out[i] = (char*)std::malloc(text.size() + 1);
        if (!cigar_out[i]) {
            set_error("out of memory");
            return fail_out(PA_E_NOMEM);
        }
        std::memcpy(cigar_out[i], text.c_str(), text.size() + 1);
    }
    if (forward_ms) *forward_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (trace_ms) *trace_ms = 0.f;  // (the engine's traceback is inside the figure above)
    return 0;
}

extern "C" int pa_batch_align(pa_batch* p, int32_t* cost_out, char** cigar_out, float* forward_ms, float* trace_ms) {
    if (!p || !p->trace) {
        set_error("pa_batch_align needs a batch made by pa_batch_create_trace");
        return PA_E_ARG;
    }
    if (small_route(p, cigar_out)) return batch_align_small(p, cost_out, cigar_out, forward_ms, trace_ms);
    const size_t P = p->pairs;
    if (P == 0) {  // an empty batch
        if (forward_ms) *forward_ms = 0.f;
        if (trace_ms) *trace_ms = 0.f;
        return 0;
    }
    static const bool prof = getenv("PA_ALIGN_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_mark = t_begin;
    auto mark = [&](const char* what) {
        if (!prof) return;
        const double t = now();
        std::fprintf(stderr, "[pa_batch_align] %-28s %8.3f ms\n", what, t - t_mark);
        t_mark = t;
    };
    if (cigar_out)
        for (size_t i = 0; i < P; ++i) cigar_out[i] = nullptr;
    // a failure after the first string has been handed out: free them all again, the caller owns outputs only on success
    auto fail_all = [&](int code) {
        if (cigar_out)
            for (size_t k = 0; k < P; ++k) {
                if (!(p->view_mode && p->in_text(cigar_out[k]))) std::free(cigar_out[k]);
                cigar_out[k] = nullptr;
            }
        return code;
    };
    // ---- everything before the chunks, on the batch's stream: profiles, clears, (full DP) the checkpointing forward pass ----
    const bool cost_only_astar = p->astar && !cigar_out;  // batched A*PA2 without CIGARs asked for: no traceback, the costs come from the forward pass
    if (const int rc = batch_forward(p, !p->astar)) return rc;
    const size_t C = p->chunk_lo.empty() ? 0 : p->chunk_lo.size() - 1;
    unsigned long long* d_total = p->d_cmeta.as<unsigned long long>();
    uint32_t* d_ticket = p->d_cmeta.as<uint32_t>() + 2 * pa_batch::kMaxChunks;
    unsigned long long* h_total = (unsigned long long*)p->h_meta;
    uint32_t* h_tlen = (uint32_t*)(p->h_meta + 64);
    uint64_t* h_dst = (uint64_t*)(p->h_meta + 64 + ((P * 4 + 7) & ~size_t(7)));
    if (!hip_ok(hipMemsetAsync(p->d_cmeta.ptr, 0, 256, p->stream), "memset chunk meta") || !hip_ok(hipEventRecord(p->ev_pre, p->stream), "event")) return PA_E_HIP;
    // ---- per chunk, on its own stream: [band search] -> traceback -> CIGAR text into the chunk's packed region -> its lengths to the host ----
    for (size_t c = 0; c < C; ++c) {
        hipStream_t s = p->cstream[c];
        const size_t lo = p->chunk_lo[c], cnt = p->chunk_lo[c + 1] - lo;
        if (!hip_ok(hipStreamWaitEvent(s, p->ev_pre, 0), "wait") || !hip_ok(hipEventRecord(p->evF0[c], s), "event")) return PA_E_HIP;
        if (p->astar)
            if (const int rc = launch_astar(p, s, lo, cnt, d_ticket + c, nullptr)) return rc;
        if (!hip_ok(hipEventRecord(p->evF1[c], s), "event")) return PA_E_HIP;
        if (cnt && !cost_only_astar) {
            static const int tbw = [] { const char* e = getenv("PA_TRACE_BLOCK_WAVES"); const int v = e ? atoi(e) : kStripBlockWaves; return v >= 1 && v <= kStripBlockWaves ? v : kStripBlockWaves; }();
            const dim3 tg((unsigned)((cnt + tbw - 1) / tbw)), tb(64 * tbw);
            const TraceJob* tjp = p->d_tjobs.as<TraceJob>();
            const int32_t* list = p->d_torder.as<int32_t>() + lo;
            uint32_t* terr = p->d_misc.as<uint32_t>() + 1;
            // the traceback starts its most expensive pairs first (PA_TRACE_ORDER=0: index order, for comparison)
            static const bool by_cost = [] { const char* e = getenv("PA_TRACE_ORDER"); return !(e && atoi(e) == 0); }();
            const int32_t* tlist = list;
            if (by_cost && cnt > (size_t)tbw) {
                hipLaunchKernelGGL(trace_order_kernel, dim3(1), dim3(1024), 0, s, tjp, list, (int)cnt, p->d_tlist.as<int32_t>() + lo, p->max_nm);
                if (!hip_ok(hipGetLastError(), "trace_order_kernel launch")) return PA_E_HIP;
                tlist = p->d_tlist.as<int32_t>() + lo;
            }
            if (p->dt_max_g > 0 && p->astar) hipLaunchKernelGGL((trace_kernel<true, true>), tg, tb, tbw * sizeof(DtLds), s, tjp, tlist, (int)cnt, terr);
            else if (p->dt_max_g > 0) hipLaunchKernelGGL((trace_kernel<true, false>), tg, tb, tbw * sizeof(DtLds), s, tjp, tlist, (int)cnt, terr);
            else if (p->astar) hipLaunchKernelGGL((trace_kernel<false, true>), tg, tb, 0, s, tjp, tlist, (int)cnt, terr);
            else hipLaunchKernelGGL((trace_kernel<false, false>), tg, tb, 0, s, tjp, tlist, (int)cnt, terr);
            if (!hip_ok(hipGetLastError(), "trace_kernel launch") || !hip_ok(hipEventRecord(p->evT1[c], s), "event")) return PA_E_HIP;
            hipLaunchKernelGGL(format_pack_kernel, dim3((unsigned)cnt), dim3(64), 0, s, p->d_cigar.as<uint32_t>(), p->d_cig_src_off.as<uint64_t>(), p->d_cigar_len.as<uint32_t>(),
                               list, p->d_packed.as<uint8_t>() + p->chunk_base[c], d_total + c, p->d_tlen_pos.as<uint32_t>() + lo, p->d_dst_pos.as<uint64_t>() + lo);
            if (!hip_ok(hipGetLastError(), "format_pack_kernel") ||
                !hip_ok(hipMemcpyAsync(h_total + c, d_total + c, 8, hipMemcpyDeviceToHost, s), "D2H total") ||
                !hip_ok(hipMemcpyAsync(h_tlen + lo, p->d_tlen_pos.as<uint32_t>() + lo, cnt * 4, hipMemcpyDeviceToHost, s), "D2H text lens") ||
                !hip_ok(hipMemcpyAsync(h_dst + lo, p->d_dst_pos.as<uint64_t>() + lo, cnt * 8, hipMemcpyDeviceToHost, s), "D2H text offsets"))
                return PA_E_HIP;
        } else if (!hip_ok(hipEventRecord(p->evT1[c], s), "event")) {
            return PA_E_HIP;
        }
    }
    mark("launches");
    // ---- per chunk, as it completes: its packed text to the host, strings to the caller (the later chunks are still on the GPU) ----
    std::vector<size_t> handed_back;  // pairs the traceback handed back (a state the reference would panic on): the host engine redoes them
    for (size_t c = 0; c < C; ++c) {
        hipStream_t s = p->cstream[c];
        const size_t lo = p->chunk_lo[c], cnt = p->chunk_lo[c + 1] - lo;
        if (!hip_ok(hipStreamSynchronize(s), "sync")) return fail_all(PA_E_HIP);
        if (!cnt || cost_only_astar || !cigar_out) continue;
        const uint64_t total = h_total[c];
        if (total > p->h_text_size) {  // pinned, so the copy runs at link speed; from the process-wide pool
            pinned_give(p->h_text, p->h_text_size);
            p->h_text = (uint8_t*)pinned_take(total + total / 4 + 4096, &p->h_text_size);
            if (!p->h_text) {
                p->h_text_size = 0;
                return fail_all(PA_E_HIP);
            }
        }
        if (total && (!hip_ok(hipMemcpyAsync(p->h_text, p->d_packed.as<uint8_t>() + p->chunk_base[c], total, hipMemcpyDeviceToHost, s), "D2H cigars") ||
                      !hip_ok(hipStreamSynchronize(s), "sync")))
            return fail_all(PA_E_HIP);
        // strings to the caller: one malloc + one copy per pair; for tens of megabytes of text (4096 x 100 kbp: 70 MB) on several threads
        std::atomic<bool> oom{false};
        auto make = [&](size_t q) {
            const size_t i = (size_t)p->torder_host[q];
            if (h_tlen[q] == kTextFailed) return;
            if (p->view_mode && C == 1) {  // pa_batch_align_view: the text stays where the copy from the GPU put it
                cigar_out[i] = (char*)(p->h_text + h_dst[q]);
                p->view_len[i] = h_tlen[q];
                return;
            }
            char* out = (char*)std::malloc((size_t)h_tlen[q] + 1);
            if (!out) {
                oom = true;
                return;
            }
            if (h_tlen[q]) std::memcpy(out, p->h_text + h_dst[q], h_tlen[q]);
            out[h_tlen[q]] = 0;
            cigar_out[i] = out;
        };
        if (total >= (size_t(8) << 20) && cnt >= 64 && host_threads() > 1 && !(p->view_mode && C == 1)) {  // (a view copies nothing)
            const unsigned nt = std::min<unsigned>(host_threads(), 8);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t)
                th.emplace_back([&, t] {
                    for (size_t q = lo + t; q < lo + cnt; q += nt) make(q);
                });
            for (auto& t : th) t.join();
        } else {
            for (size_t q = lo; q < lo + cnt; ++q) make(q);
        }
        if (oom) {
            set_error("out of memory");
            return fail_all(PA_E_NOMEM);
        }
        for (size_t q = lo; q < lo + cnt; ++q)
            if (h_tlen[q] == kTextFailed) handed_back.push_back((size_t)p->torder_host[q]);
    }
    mark("chunks: text D2H + strings");
    // ---- the small per-pair arrays, once ----
    std::vector<uint32_t> lens(P, 0);
    std::vector<int32_t> costs(P, 0);
    uint32_t misc[4] = {0, 0, 0, 0};
    if (P && !cost_only_astar &&
        (!hip_ok(hipMemcpy(lens.data(), p->d_cigar_len.ptr, P * 4, hipMemcpyDeviceToHost), "D2H lens") ||
         !hip_ok(hipMemcpy(costs.data(), p->d_costs.ptr, P * 4, hipMemcpyDeviceToHost), "D2H costs")))
        return fail_all(PA_E_HIP);
    if (!hip_ok(hipMemcpy(misc, p->d_misc.ptr, 16, hipMemcpyDeviceToHost), "D2H")) return fail_all(PA_E_HIP);
    if (misc[3]) {
        set_error("sequence contains a base outside ACGT");
        return fail_all(PA_E_INVALID_BASE);
    }
    if (misc[1] != PA_ERR_NONE) {
        set_error("device spin timeout (err=%u)", misc[1]);
        return fail_all(PA_E_TIMEOUT);
    }
    p->gran_dirty = false;
    // kernel times.  With several chunks the kernels of different chunks run side by side: the figures are the SPANS from the first start to
    // the last end of each phase (equal to the kernel times when there is one chunk), and the two spans overlap.
    auto span = [&](hipEvent_t* from, hipEvent_t* to, float* out) -> bool {
        float best = 0.f;
        for (size_t c0 = 0; c0 < C; ++c0)
            for (size_t c1 = 0; c1 < C; ++c1) {
                float ms = 0.f;
                if (!hip_ok(hipEventElapsedTime(&ms, from[c0], to[c1]), "elapsed")) return false;
                if (ms > best) best = ms;
            }
        *out = best;
        return true;
    };
    if (forward_ms) {
        *forward_ms = 0.f;
        if (p->astar) {
            if (!span(p->evF0, p->evF1, forward_ms)) return fail_all(PA_E_HIP);
        } else if (!p->jobs.empty() && !hip_ok(hipEventElapsedTime(forward_ms, p->ev0, p->ev1), "elapsed")) {
            return fail_all(PA_E_HIP);
        }
    }
    if (trace_ms) {
        *trace_ms = 0.f;
        if (!span(p->evF1, p->evT1, trace_ms)) return fail_all(PA_E_HIP);
    }
    std::vector<apa2::PairResult> results;
    if (p->astar) {  // per-pair statistics (domain.rs:31-43) of the band search and the traceback
        results.resize(P);
        std::vector<uint32_t> ts(P * 8, 0);
        if (P && (!hip_ok(hipMemcpy(results.data(), p->d_results.ptr, P * sizeof(apa2::PairResult), hipMemcpyDeviceToHost), "D2H results") ||
                  !hip_ok(hipMemcpy(ts.data(), p->d_tstats.ptr, P * 32, hipMemcpyDeviceToHost), "D2H trace stats")))
            return fail_all(PA_E_HIP);
        if (prof && !p->astar_full && P) {  // diagnostics: the spread of the pairs' band-search times (what ends the launch: the work, or a few chains?)
            std::vector<double> ms;
            for (const apa2::PairResult& r : results)
                if (r.pad0) ms.push_back((double)r.pad0 / 1e5);
            std::sort(ms.begin(), ms.end());
            auto q = [&](double f) { return ms.empty() ? 0.0 : ms[(size_t)(f * (double)(ms.size() - 1))]; };
            double sum = 0;
            for (double x : ms) sum += x;
            std::fprintf(stderr, "[pa_batch_align] per-pair band search ms: min %.2f  median %.2f  p90 %.2f  p99 %.2f  p99.9 %.2f  max %.2f  sum %.1f  (%zu pairs)\n", q(0), q(0.5), q(0.9),
                         q(0.99), q(0.999), q(1.0), sum, ms.size());
        }
        p->pair_stats.assign(P, pa_astarpa2_stats{});
        p->apa2_strip_instr = 0;
        for (size_t i = 0; i < P; ++i) {
            pa_astarpa2_stats& st = p->pair_stats[i];
            const apa2::PairResult& r = results[i];
            if (p->sp.doubling == apa2::kDoublingBand) {  // (the reference reports block counters after a band doubling only, lib.rs:158)
                st.num_blocks = r.num_blocks;
                st.num_incremental_blocks = r.num_incremental_blocks;
                st.computed_lanes = r.computed_lanes;
                st.unique_lanes = r.unique_lanes;
            }
            st.f_max_tries = r.f_max_tries;
            st.sanity_violations = r.sanity_violations;
            p->apa2_strip_instr += (double)r.strip_instr;
            if (cost_only_astar) {  // no traceback ran: the cost is the forward pass's, a pair it handed back goes to the host engine
                costs[i] = r.cost;
                lens[i] = r.status != apa2::kOk ? kTraceFailed : 0u;
                if (r.status != apa2::kOk) handed_back.push_back(i);
                continue;
            }
            st.dt_trace_tries = ts[8 * i + 0];
            st.dt_trace_success = ts[8 * i + 1];
            st.dt_trace_fallback = ts[8 * i + 2];
            st.fill_tries = ts[8 * i + 3];
            st.fill_success = ts[8 * i + 4];
            st.fill_fallback = ts[8 * i + 5];
        }
    }
    // ---- second round: pairs whose band left their window of the column store, again with full-height slots ----
    if (p->astar) {
        std::vector<size_t> redo;
        for (size_t i = 0; i < P; ++i)
            if (results[i].status == apa2::kErrWindow) redo.push_back(i);
        // The second round's memory is bounded: the pairs go in sub-batches whose full-height stores stay below ~24 GB each (one pair
        // alone may exceed it: 9.8 MB per 100 kbp pair, 1 GB per 1 Mbp pair), one sub-batch at a time; pa_batch_window_retry_bytes reports
        // the largest.  PA_WINDOW_RETRY_BYTES overrides the bound (tests).
        double retry_cap = 24e9;
        if (const char* e = getenv("PA_WINDOW_RETRY_BYTES")) retry_cap = std::max(1.0, atof(e));
        for (size_t r0 = 0; r0 < redo.size();) {
            size_t r1 = r0;
            double bytes = 0;
            while (r1 < redo.size()) {
                const size_t i = redo[r1];
                const double need = ((double)p->n[i] / 256.0 + 2.0) * (double)((p->m[i] + 63) / 64) * 16.0;
                if (r1 > r0 && bytes + need > retry_cap) break;
                bytes += need;
                r1 += 1;
            }
            p->window_retry_peak_bytes = std::max(p->window_retry_peak_bytes, bytes);
            const size_t R = r1 - r0;
            std::vector<std::vector<uint8_t>> ra(R), rb(R);
            std::vector<const uint8_t*> ap(R), bp(R);
            std::vector<size_t> al(R), bl(R);
            for (size_t q = 0; q < R; ++q) {
                const size_t i = redo[r0 + q];
                ra[q].resize(p->n[i]);
                rb[q].resize(p->m[i]);
                if ((p->n[i] && !hip_ok(hipMemcpy(ra[q].data(), p->d_a.as<uint8_t>() + p->a_off[i], p->n[i], hipMemcpyDeviceToHost), "D2H a")) ||
                    (p->m[i] && !hip_ok(hipMemcpy(rb[q].data(), p->d_b.as<uint8_t>() + p->b_off[i], p->m[i], hipMemcpyDeviceToHost), "D2H b")))
                    return fail_all(PA_E_HIP);
                ap[q] = ra[q].data();
                bp[q] = rb[q].data();
                al[q] = p->n[i];
                bl[q] = p->m[i];
            }
            std::unique_ptr<pa_batch> sub(batch_create(ap.data(), al.data(), bp.data(), bl.data(), R, true, -1.f, p->dt_max_g, p->dt_fr_drop, &p->aparams_c, 0));
            if (!sub) return fail_all(PA_E_HIP);
            std::vector<int32_t> c2(R, 0);
            std::vector<char*> g2(R, nullptr);
            const int rc2 = pa_batch_align(sub.get(), c2.data(), cigar_out ? g2.data() : nullptr, nullptr, nullptr);
            if (rc2 != 0) return fail_all(rc2);  // (pa_batch_align hands out no strings when it fails)
            for (size_t q = 0; q < R; ++q) {
                const size_t i = redo[r0 + q];
                costs[i] = c2[q];
                lens[i] = 0;
                results[i].status = apa2::kOk;
                if (q < sub->pair_stats.size()) p->pair_stats[i] = sub->pair_stats[q];
                if (cigar_out) {
                    if (!(p->view_mode && p->in_text(cigar_out[i]))) std::free(cigar_out[i]);
                    cigar_out[i] = g2[q];
                }
            }
            p->trace_fallbacks += sub->trace_fallbacks;
            p->window_retries += R;
            r0 = r1;
        }
        if (!redo.empty()) {  // (they are not the host engine's)
            std::vector<size_t> keep;
            for (const size_t i : handed_back)
                if (results[i].status != apa2::kOk || lens[i] == kTraceFailed) keep.push_back(i);
            handed_back.swap(keep);
        }
    }
    mark("second round (windows)");
    if (!cigar_out && !cost_only_astar)  // (costs alone of a traced full-DP batch: the pairs the traceback handed back are not redone)
        handed_back.clear();
    if (cigar_out && !cost_only_astar) {  // (without cigar_out the loop over the chunks above did not look at the lengths)
        handed_back.clear();
        for (size_t i = 0; i < P; ++i)
            if (lens[i] == kTraceFailed) handed_back.push_back(i);
    }
    for (size_t i = 0; i < P; ++i) cost_out[i] = costs[i];
    mark("small arrays + statistics");
    pa_astarpa2_params fallback = p->astar ? p->aparams_c : traced_batch_params();
    if (!p->astar && p->dt_max_g > 0) {
        fallback.front.dt_trace = 1;
        fallback.front.max_g = p->dt_max_g;
        fallback.front.fr_drop = p->dt_fr_drop;
    }
    for (const size_t i : handed_back) {
        // a state the reference would panic on (or one the kernels leave alone): the host engine redoes this pair
        std::string text;
        p->trace_fallbacks += 1;
        std::vector<uint8_t> ba(p->n[i]), bb(p->m[i]);
        if ((p->n[i] && !hip_ok(hipMemcpy(ba.data(), p->d_a.as<uint8_t>() + p->a_off[i], p->n[i], hipMemcpyDeviceToHost), "D2H a")) ||
            (p->m[i] && !hip_ok(hipMemcpy(bb.data(), p->d_b.as<uint8_t>() + p->b_off[i], p->m[i], hipMemcpyDeviceToHost), "D2H b")))
            return fail_all(PA_E_HIP);
        int32_t c = 0;
        pa_astarpa2_stats fst{};
        const int rc = align_hip(ba.data(), p->n[i], bb.data(), p->m[i], fallback, !cost_only_astar, false, &c, &text, &fst);
        if (rc != 0) return fail_all(rc);
        if (p->astar) {
            p->pair_stats[i] = fst;
            if (results[i].status != apa2::kOk) costs[i] = c;  // the forward pass itself handed the pair back
            cost_out[i] = c;
        }
        if (c != costs[i]) {
            set_error("traceback fallback disagrees with the batched cost (pair %zu: %d vs %d)", i, c, costs[i]);
            return fail_all(PA_E_INTERNAL);
        }
        if (!cigar_out) continue;
        if (!(p->view_mode && p->in_text(cigar_out[i]))) std::free(cigar_out[i]);
        cigar_out[i] = (char*)std::malloc(text.size() + 1);
        if (!cigar_out[i]) {
            set_error("out of memory");
            return fail_all(PA_E_NOMEM);
        }
        std::memcpy(cigar_out[i], text.c_str(), text.size() + 1);
    }
    mark("pairs handed back");
    return 0;
}

// Diagnostics / tests: the matches of GCSH (seed length k, local pruning p_local) of one pair AS THE GPU FINDS THEM (gcsh_build_kernel.hpp),
// by start: out_ij[2 t], out_ij[2 t + 1] for t < min(count, cap_out).  Returns the count, or -(100 + status) when the kernel gave up.
extern "C" long pa_debug_gcsh_matches(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, int32_t k, int32_t p_local, int32_t* out_ij, size_t cap_out) {
    if (!ensure_device()) return PA_E_HIP;
    if (!a || !b || a_len == 0 || b_len == 0 || k < 1 || k > 31 || p_local < 0 || p_local > apa2::kBuildMaxP) return PA_E_ARG;
    const size_t ns = a_len >= (size_t)k ? (a_len - k) / k + 1 : 0, cap = ns + ns / 2 + 2048;
    size_t tsz = 64;
    while (tsz < 2 * ns + 1) tsz *= 2;
    DeviceBuf d_a, d_b, d_w, d_mi, d_mj, d_win, d_job, d_out;
    const size_t words = 4 * ns + 1 + tsz + 4 * cap;
    if (!d_a.alloc(a_len + 64) || !d_b.alloc(b_len + 64) || !d_w.alloc(words * 4 + 2 * cap + 64) || !d_mi.alloc(cap * 4) || !d_mj.alloc(cap * 4) ||
        !d_win.alloc(std::max<size_t>(ns, 1) * sizeof(apa2::GcshSeedWindow)) || !d_job.alloc(sizeof(apa2::GcshBuildJob)) || !d_out.alloc(64))
        return PA_E_NOMEM;
    apa2::GcshBuildJob x;
    std::memset(&x, 0, sizeof x);
    int32_t* w32 = d_w.as<int32_t>();
    size_t o = 0;
    x.a = d_a.as<uint8_t>();
    x.b = d_b.as<uint8_t>();
    x.keys = (uint32_t*)(w32 + o), o += ns;
    x.next_same = w32 + o, o += ns;
    x.cnt = w32 + o, o += ns + 1;
    x.fill = w32 + o, o += ns;
    x.slot = w32 + o, o += tsz;
    x.tmp_s = w32 + o, o += cap;
    x.tmp_j = w32 + o, o += cap;
    x.gpos = w32 + o, o += cap;
    x.cj = w32 + o, o += cap;
    x.flag = (uint8_t*)(w32 + words);
    x.keptg = x.flag + cap;
    x.mi = d_mi.as<int32_t>();
    x.mj = d_mj.as<int32_t>();
    x.win0 = d_win.as<apa2::GcshSeedWindow>();
    x.nmatch_out = d_out.as<int32_t>();
    x.status = d_out.as<uint32_t>() + 1;
    x.n = (int32_t)a_len;
    x.m = (int32_t)b_len;
    x.k = k;
    x.p = p_local;
    x.nseeds = (int32_t)ns;
    x.tsize = (int32_t)tsz;
    x.cap = (int32_t)cap;
    DeviceBuf d_clk;
    static const bool clocks = getenv("PA_BUILD_CLOCKS") != nullptr;
    if (clocks) {
        if (!d_clk.alloc(128) || !hip_ok(hipMemset(d_clk.ptr, 0, 128), "memset")) return PA_E_HIP;
        x.clocks = d_clk.as<unsigned long long>();
    }
    int32_t res[4] = {0, 0, 0, 0};
    if (!hip_ok(hipMemcpy(d_a.ptr, a, a_len, hipMemcpyHostToDevice), "H2D") || !hip_ok(hipMemcpy(d_b.ptr, b, b_len, hipMemcpyHostToDevice), "H2D") ||
        !hip_ok(hipMemset(d_out.ptr, 0, 64), "memset") || !hip_ok(hipMemcpy(d_job.ptr, &x, sizeof x, hipMemcpyHostToDevice), "H2D"))
        return PA_E_HIP;
    if (!hip_ok(apa2::launch_gcsh_build_kernel(1, 0, d_job.as<apa2::GcshBuildJob>(), 1, d_out.as<uint32_t>() + 8), "gcsh_build_kernel") || !hip_ok(hipDeviceSynchronize(), "sync") || !hip_ok(hipMemcpy(res, d_out.ptr, 16, hipMemcpyDeviceToHost), "D2H"))
        return PA_E_HIP;
    if (clocks) {
        unsigned long long c[16] = {0};
        (void)hipMemcpy(c, d_clk.ptr, 128, hipMemcpyDeviceToHost);
        std::fprintf(stderr, "[gcsh build] n %zu m %zu k %d p %d: A %.3f  B %.3f  C %.3f  D %.3f  E %.3f  F %.3f ms; %llu candidates, %llu kept alone, %llu searches in E; D: %llu search levels of %llu that its rounds last (deepest lane x lanes); status %d\n", a_len, b_len,
                     k, p_local, c[0] * 1e-5, c[1] * 1e-5, c[2] * 1e-5, c[3] * 1e-5, c[4] * 1e-5, c[5] * 1e-5, c[6], c[7], c[8], c[9], c[10], res[1]);
    }
    if (res[1] != 0) return -(100 + (long)res[1]);
    const size_t cnt = (size_t)std::max(res[0], 0), take = std::min(cnt, cap_out);
    if (take && out_ij) {
        std::vector<int32_t> mi(take), mj(take);
        if (!hip_ok(hipMemcpy(mi.data(), d_mi.ptr, take * 4, hipMemcpyDeviceToHost), "D2H") || !hip_ok(hipMemcpy(mj.data(), d_mj.ptr, take * 4, hipMemcpyDeviceToHost), "D2H"))
            return PA_E_HIP;
        for (size_t t = 0; t < take; ++t) {
            out_ij[2 * t] = mi[t];
            out_ij[2 * t + 1] = mj[t];
        }
    }
    return (long)cnt;
}

// Diagnostics / tests: the DEVICE form of GCSH alone.  The matches are found on the host (csrc/gcsh.hpp), one wavefront derives the contours
// and evaluates h at nq positions (queries[2 t], queries[2 t + 1]); out[t] = h, out[nq] = number of contour layers (incl. layer 0).
extern "C" int pa_debug_gcsh_probe(const uint8_t* a, size_t a_len, const uint8_t* b, size_t b_len, int32_t k, int32_t p_local, const int32_t* queries,
                                   size_t nq, int32_t* out) {
    if (!ensure_device()) return PA_E_HIP;
    if (!a || !b || a_len == 0 || b_len == 0 || k < 1 || k > 31 || (!queries && nq) || !out) return PA_E_ARG;
    engine::GcshHeuristic gh(a, (engine::I)a_len, b, (engine::I)b_len, k, p_local, false, false);
    const size_t M = gh.by_start.size();
    std::vector<int32_t> mi(M), mj(M);
    for (size_t t = 0; t < M; ++t) {
        mi[t] = gh.by_start[t].i;
        mj[t] = gh.by_start[t].j;
    }
    DeviceBuf d_mi, d_mj, d_act, d_lrec, d_cell, d_job, d_q, d_out, d_err;
    if (!d_mi.alloc(std::max<size_t>(M, 1) * 4) || !d_mj.alloc(std::max<size_t>(M, 1) * 4) || !d_act.alloc(std::max<size_t>(M, 64)) ||
        !d_lrec.alloc((M + 2) * sizeof(apa2::GcshCell)) || !d_cell.alloc(std::max<size_t>(M, 1) * sizeof(apa2::GcshCell)) || !d_job.alloc(sizeof(apa2::FullJob)) ||
        !d_q.alloc(std::max<size_t>(nq, 1) * 8) || !d_out.alloc((nq + 1) * 4) || !d_err.alloc(64))
        return PA_E_NOMEM;
    apa2::FullJob j;
    std::memset(&j, 0, sizeof j);
    j.n = (int32_t)a_len;
    j.m = (int32_t)b_len;
    j.heur = apa2::kFullHeurGcsh;
    j.g.mi = d_mi.as<int32_t>();
    j.g.mj = d_mj.as<int32_t>();
    j.g.active = d_act.as<uint8_t>();
    j.g.lrec = d_lrec.as<apa2::GcshCell>();
    j.g.cell = d_cell.as<apa2::GcshCell>();
    j.g.nmatch = (int32_t)M;
    j.g.nlayers = 1;
    j.g.n = j.n;
    j.g.m = j.m;
    j.g.k = k;
    j.g.nseeds = gh.nseeds;
    if ((M && (!hip_ok(hipMemcpy(d_mi.ptr, mi.data(), M * 4, hipMemcpyHostToDevice), "H2D") || !hip_ok(hipMemcpy(d_mj.ptr, mj.data(), M * 4, hipMemcpyHostToDevice), "H2D"))) ||
        !hip_ok(hipMemset(d_act.ptr, 1, std::max<size_t>(M, 64)), "memset") || !hip_ok(hipMemset(d_err.ptr, 0, 64), "memset") ||
        !hip_ok(hipMemcpy(d_job.ptr, &j, sizeof j, hipMemcpyHostToDevice), "H2D") ||
        (nq && !hip_ok(hipMemcpy(d_q.ptr, queries, nq * 8, hipMemcpyHostToDevice), "H2D")))
        return PA_E_HIP;
    if (!hip_ok(apa2::launch_gcsh_probe_kernel(0, d_job.as<apa2::FullJob>(), d_q.as<int32_t>(), (int)nq, d_out.as<int32_t>(), d_err.as<uint32_t>()), "gcsh_probe_kernel") || !hip_ok(hipDeviceSynchronize(), "sync") ||
        !hip_ok(hipMemcpy(out, d_out.ptr, (nq + 1) * 4, hipMemcpyDeviceToHost), "D2H"))
        return PA_E_HIP;
    return 0;
}

// Reporting (whole-family batches, pa_batch_create_params with GCSH / pruning / incremental doubling): host milliseconds spent finding
// the matches of the heuristic at creation, their number, and -- with PA_APA2_PROBE_STATS set -- the h probes of the last forward pass
// and the load rounds (64 layers each) they took.
// phase_wave_ms[0..7) (PA_APA2_PROBE_STATS): wavefront-milliseconds (summed over all wavefronts; 100 MHz clock) spent deriving contours,
// in the DP strips, in h probes, in Block::index, in prune_block, initialising block columns, and in total.
#ifdef PA_TRACE_CLOCKS
// Experiments (-DPA_TRACE_CLOCKS): the traceback kernels' clocks since the last call, then zeroed.
extern "C" int pa_debug_trace_clocks(double* out10) {
    unsigned long long v[10] = {0};
    if (!hip_ok(hipDeviceSynchronize(), "sync") || !hip_ok(hipMemcpyFromSymbol(v, HIP_SYMBOL(pa::g_trace_clk), sizeof(v)), "trace clocks")) return PA_E_HIP;
    for (int i = 0; i < 10; ++i) out10[i] = (double)v[i];
    unsigned long long z[10] = {0};
    return hip_ok(hipMemcpyToSymbol(HIP_SYMBOL(pa::g_trace_clk), z, sizeof(z)), "trace clocks") ? 0 : PA_E_HIP;
}
#endif
// Diagnostics: the rendezvous of half-wave blocks in the last forward pass of a batched A*PA2 plan -- out4[0] strips that ran fused with a
// partner's (counted at the wavefront that ran both), [1] strips a partner ran, [2] strips that ran alone, [3] of those: posted, then withdrawn.
extern "C" int pa_batch_rdv_stats(const pa_batch* p, uint64_t* out4) {
    if (!p || !out4) return PA_E_ARG;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (!p->d_rdv.ptr) return 0;
    unsigned long long v[4] = {0, 0, 0, 0};
    if (!hip_ok(hipMemcpy(v, p->d_rdv.ptr, sizeof(v), hipMemcpyDeviceToHost), "D2H rendezvous counters")) return PA_E_HIP;
    for (int i = 0; i < 4; ++i) out4[i] = v[i];
    return 0;
}
extern "C" void pa_batch_full_info(const pa_batch* p, double* build_ms, double* matches, double* probes, double* rounds, double* phase_wave_ms) {
    if (build_ms) *build_ms = p ? p->full_build_ms : 0;
    if (matches) *matches = p ? (double)p->full_matches : 0;
    if (p && p->device_build && p->pairs) {  // the GPU found them: the build kernel's time at creation (negative = on the device), their number
        float ms = 0.f;
        if (build_ms && p->evB0 && hipEventElapsedTime(&ms, p->evB0, p->evB1) == hipSuccess) *build_ms = -(double)ms;
        std::vector<apa2::FullJob> fj(p->pairs);
        if (matches && hipMemcpy(fj.data(), p->d_fjobs.ptr, p->pairs * sizeof(apa2::FullJob), hipMemcpyDeviceToHost) == hipSuccess) {
            double tot = 0;
            for (const auto& j : fj) tot += j.g.nmatch > 0 ? j.g.nmatch : 0;
            *matches = tot;
        }
    }
    unsigned long long pr[16] = {0};
    if (p && p->astar_full && p->d_probe.ptr) (void)hipMemcpy(pr, p->d_probe.ptr, 128, hipMemcpyDeviceToHost);
    if (p && p->astar_full && p->d_probe.ptr && getenv("PA_APA2_PROBE_STATS") && p->pairs) {
        // diagnostics: how long every pair's band search took its wavefront, by XCD (is the launch's length the work or the placement?)
        std::vector<unsigned long long> pp(p->pairs);
        if (hipMemcpy(pp.data(), (const uint8_t*)p->d_probe.ptr + 128, p->pairs * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            std::vector<double> all;
            std::vector<std::vector<double>> by_xcc(8);
            for (unsigned long long v : pp) {
                const double ms = (double)(uint32_t)v / 1e5;
                if (ms <= 0) continue;
                all.push_back(ms);
                by_xcc[(size_t)((v >> 32) & 7)].push_back(ms);
            }
            auto q = [](std::vector<double>& x, double f) { return x.empty() ? 0.0 : x[(size_t)(f * (double)(x.size() - 1))]; };
            std::sort(all.begin(), all.end());
            std::fprintf(stderr, "[apa2_full] h probes %llu, answered from a register window %llu, load rounds %llu\n", pr[0], pr[9], pr[1]);
            std::fprintf(stderr, "[apa2_full] per-pair band search ms: min %.2f  p10 %.2f  median %.2f  p90 %.2f  p99 %.2f  max %.2f  (%zu pairs)\n", q(all, 0), q(all, 0.1), q(all, 0.5),
                         q(all, 0.9), q(all, 0.99), q(all, 1.0), all.size());
            for (size_t x = 0; x < 8; ++x) {
                std::sort(by_xcc[x].begin(), by_xcc[x].end());
                std::fprintf(stderr, "[apa2_full]   XCD %zu: %5zu pairs  median %.2f  max %.2f\n", x, by_xcc[x].size(), q(by_xcc[x], 0.5), q(by_xcc[x], 1.0));
            }
        }
    }
    if (probes) *probes = (double)pr[0];
    if (rounds) *rounds = (double)pr[1];
    if (phase_wave_ms)
        for (int t = 0; t < 7; ++t) phase_wave_ms[t] = (double)pr[2 + t] * 1e-5;
}

// pa_batch_align without the per-pair strings: text_out[i] points at text_len_out[i] characters of pair i's CIGAR (NOT NUL-terminated)
// inside memory the plan owns -- valid until the next alignment call on this plan or its destruction, nothing to free.  For callers that
// copy the text somewhere of their own anyway (a language binding building its string objects, a writer of the pa-bin CSV): 10 000
// malloc + copy + free less per C4 batch.
extern "C" int pa_batch_align_view(pa_batch* p, int32_t* cost_out, const char** text_out, uint32_t* text_len_out, float* forward_ms, float* trace_ms) {
    if (!p || !text_out || !text_len_out) {
        set_error("pa_batch_align_view: plan, text_out and text_len_out must not be NULL");
        return PA_E_ARG;
    }
    p->free_view_owned();
    p->view_len.assign(p->pairs, 0);
    p->view_mode = true;
    const int rc = pa_batch_align(p, cost_out, const_cast<char**>(text_out), forward_ms, trace_ms);
    p->view_mode = false;
    if (rc != 0) return rc;
    for (size_t i = 0; i < p->pairs; ++i) {
        const char* q = text_out[i];
        if (!q) {
            text_len_out[i] = 0;
        } else if (p->in_text(q)) {
            text_len_out[i] = p->view_len[i];
        } else {  // a string of its own (host engine, second round, ...): the plan keeps it until the next call
            text_len_out[i] = (uint32_t)std::strlen(q);
            p->view_owned.push_back(const_cast<char*>(q));
        }
    }
    return 0;
}

extern "C" size_t pa_batch_trace_fallbacks(const pa_batch* p) { return p ? p->trace_fallbacks : 0; }
// Pairs (summed over all pa_batch_align calls) whose band left their window of the block-column store and that were aligned again with
// full-height slots.
extern "C" size_t pa_batch_window_retries(const pa_batch* p) { return p ? p->window_retries : 0; }
extern "C" double pa_batch_window_retry_bytes(const pa_batch* p) { return p ? p->window_retry_peak_bytes : 0.0; }

extern "C" int pa_batch_pair_stats(const pa_batch* p, pa_astarpa2_stats* stats_out) {
    if (!p || !p->astar || !stats_out || p->pair_stats.size() != p->pairs) {
        set_error("pa_batch_pair_stats needs a batch made by pa_batch_create_params after pa_batch_align");
        return PA_E_ARG;
    }
    for (size_t i = 0; i < p->pairs; ++i) stats_out[i] = p->pair_stats[i];
    return 0;
}

extern "C" void pa_batch_stats(const pa_batch* p, double* cells, double* word_updates, double* strips, double* algo_bytes) {
    if (cells) *cells = p->cells;
    if (word_updates) *word_updates = p->word_updates;
    if (strips) *strips = p->sliced ? (double)slice::info(p->sliced).jobs : (double)p->jobs.size();
    if (algo_bytes) *algo_bytes = p->algo_bytes;
}

extern "C" int pa_batch_slice_info(const pa_batch* p, double* groups, double* jobs, double* computed_cells, double* device_bytes, double* boundary_bytes) {
    if (!p || !p->sliced) return 0;
    const slice::Info i = slice::info(p->sliced);
    if (groups) *groups = (double)i.groups;
    if (jobs) *jobs = (double)i.jobs;
    if (computed_cells) *computed_cells = i.computed_rows_cells;
    if (device_bytes) *device_bytes = i.device_bytes;
    if (boundary_bytes) *boundary_bytes = i.boundary_bytes;
    return i.rows_per_lane;
}

extern "C" void pa_batch_shape(const pa_batch* p, int* k, int* sequential, double* valu_instructions) {
    if (p->sliced) {
        if (k) *k = 0;
        if (sequential) *sequential = 0;
        if (valu_instructions) *valu_instructions = slice::info(p->sliced).valu_instructions;
        return;
    }
    if (k) *k = p->k;
    if (sequential) *sequential = p->sequential ? 1 : 0;
    if (valu_instructions && p->astar) {
        *valu_instructions = p->apa2_strip_instr;  // of the last pa_batch_align: the DP strips alone (the band logic comes on top)
    } else if (valu_instructions) {
        double t = 0;
        for (const StripJob& j : p->jobs) {
            const int kj = p->sequential ? j.k : p->k;
            // run_strip: (C + 2) chunks of 32 steps; tall strips read their eq words from LDS (strip_kernel.hpp LdsEq)
            const bool lds_eq = kj >= 4 && !getenv(p->sequential ? "PA_PAIR_NO_LDSEQ" : "PA_STRIP_NO_LDSEQ");
            t += 32.0 * (double)((j.n + 31) / 32 + 2) * (lds_eq ? 10.0 + 10.0 * kj : 11.0 + 12.0 * kj);
        }
        *valu_instructions = t;
    }
}

extern "C" void pa_batch_destroy(pa_batch* p) { delete p; }
