// The device form of the GCSH heuristic (csrc/gcsh_flat.hpp) ON the device: h(i, j) for random positions of a 100 kbp pair from a
// kernel against csrc/gcsh.hpp on the host, and what one probe costs a lone wavefront -- the layers read from global memory and from a
// copy in LDS (DESIGN.md 9, item 3 estimated ~6 us per 64 probes from L2 and 1-2 us from LDS; this measures it).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/gcsh_flat_probe tools/gcsh_flat_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../astar-pairwise-aligner_amd/csrc/engine.hpp"
#include "../astar-pairwise-aligner_amd/csrc/gcsh_flat.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

using namespace pa::apa2;

__global__ void probe_global(GcshFlat g, const int32_t* qi, const int32_t* qj, int32_t nq, int32_t* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nq) out[t] = gcsh_h(g, qi[t], qj[t]);
}

// one wavefront, `rounds` dependent rounds of 64 probes (lane l probes row j0 + l of column i; the next round starts from the answers)
template <bool LDS>
__global__ __launch_bounds__(64) void probe_latency(GcshFlat g, int32_t npts, int32_t rounds, int32_t i, int32_t j0, int32_t* out, unsigned long long* clocks) {
    extern __shared__ int32_t sh[];
    GcshFlat gl = g;
    if (LDS) {  // layer offsets, x, y into LDS
        int32_t* off = sh;
        int32_t* px = off + g.nlayers + 1;
        int32_t* py = px + npts;
        for (int t = threadIdx.x; t <= g.nlayers; t += 64) off[t] = g.layer_off[t];
        for (int t = threadIdx.x; t < npts; t += 64) {
            px[t] = g.px[t];
            py[t] = g.py[t];
        }
        __syncthreads();
        gl.layer_off = off;
        gl.px = px;
        gl.py = py;
    }
    int32_t j = j0 + (int32_t)threadIdx.x, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        const int32_t v = gcsh_h(gl, i, j);
        acc += v;
        j = j0 + (int32_t)threadIdx.x + (v & 1);  // (a dependency from one round to the next, as between the probes of a scan)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) clocks[0] = t1 - t0;
}

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 11);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 100000;
    const double e = argc > 2 ? atof(argv[2]) : 0.05;
    std::vector<uint8_t> a(n), b;
    for (auto& c : a) c = "ACGT"[rnd() & 3];
    for (int i = 0; i < n; ++i) {  // substitutions, insertions, deletions at rate e
        const double u = (rnd() & 0xFFFFFF) / (double)0x1000000;
        if (u < e / 3) continue;                                  // deletion
        if (u < 2 * e / 3) b.push_back("ACGT"[rnd() & 3]);        // insertion before
        b.push_back(u < e ? "ACGT"[rnd() & 3] : a[i]);            // (substitution or) copy
    }
    const int m = (int)b.size();
    pa::engine::GcshHeuristic gh(a.data(), n, b.data(), m, 12, 14, true);
    GcshFlatStorage flat;
    flat.build(gh.layers);
    const int npts = (int)flat.px.size(), nl = (int)flat.layer_off.size() - 1;
    printf("pair %d x %d at %.2f: %zu matches kept, %d layers, %d front points (%.1f KB as three arrays)\n", n, m, e, gh.by_start.size(), nl, npts,
           (4.0 * (nl + 1) + 8.0 * npts) / 1024);
    int32_t *d_off, *d_px, *d_py, *d_qi, *d_qj, *d_out;
    unsigned long long* d_clk;
    CK(hipMalloc(&d_off, (nl + 1) * 4)); CK(hipMalloc(&d_px, (npts + 1) * 4)); CK(hipMalloc(&d_py, (npts + 1) * 4));
    CK(hipMemcpy(d_off, flat.layer_off.data(), (nl + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_px, flat.px.data(), npts * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_py, flat.py.data(), npts * 4, hipMemcpyHostToDevice));
    GcshFlat g = flat.view(n, m, 12, gh.nseeds);
    g.layer_off = d_off; g.px = d_px; g.py = d_py;
    // ---- correctness: 200 000 random positions, plus positions near the diagonal ----
    const int nq = 200000;
    std::vector<int32_t> qi(nq), qj(nq), want(nq), got(nq);
    for (int t = 0; t < nq; ++t) {
        qi[t] = (int32_t)(rnd() % (uint32_t)(n + 1));
        qj[t] = (t & 1) ? (int32_t)(rnd() % (uint32_t)(m + 1)) : std::min(m, std::max(0, qi[t] + (int32_t)(rnd() % 2001) - 1000));
        want[t] = gh.h(qi[t], qj[t]);
    }
    CK(hipMalloc(&d_qi, nq * 4)); CK(hipMalloc(&d_qj, nq * 4)); CK(hipMalloc(&d_out, nq * 4)); CK(hipMalloc(&d_clk, 8));
    CK(hipMemcpy(d_qi, qi.data(), nq * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_qj, qj.data(), nq * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_global, dim3((nq + 255) / 256), dim3(256), 0, 0, g, d_qi, d_qj, nq, d_out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), d_out, nq * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < nq; ++t) bad += got[t] != want[t];
    printf("h on the device against gcsh.hpp on the host: %d positions, %d mismatches\n", nq, bad);
    // ---- latency of a lone wavefront: 64 probes per round, dependent rounds ----
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    const int rounds = 2000;
    for (int lds = 0; lds < 2; ++lds) {
        const size_t shb = lds ? (size_t)(nl + 1 + 2 * npts) * 4 : 0;
        if (lds && shb > 160 * 1024) {
            printf("LDS copy: %zu bytes do not fit\n", shb);
            continue;
        }
        if (lds) CK(hipFuncSetAttribute((const void*)probe_latency<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            if (lds) hipLaunchKernelGGL(probe_latency<true>, dim3(1), dim3(64), shb, 0, g, npts, rounds, n / 2, n / 2 - 32, d_out, d_clk);
            else hipLaunchKernelGGL(probe_latency<false>, dim3(1), dim3(64), 0, 0, g, npts, rounds, n / 2, n / 2 - 32, d_out, d_clk);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) printf("%s: %d rounds of 64 probes by one wavefront: %.3f ms = %.2f us per round (kernel incl. %s)\n", lds ? "layers in LDS   " : "layers in global", rounds, ms,
                                 ms * 1e3 / rounds, lds ? "the copy into LDS" : "nothing else");
        }
    }
    return bad ? 1 : 0;
}
