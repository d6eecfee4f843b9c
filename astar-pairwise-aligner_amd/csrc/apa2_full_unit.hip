// apa2_full_unit.hip -- translation unit of pa::apa2::apa2_full_kernel and gcsh_probe_kernel (apa2_full_kernel.hpp): the batched band
// search of the whole A*PA2 family (GCSH, pruning, incremental doubling) and the diagnostics kernel of its heuristic.
#define PA_UNIT_APA2_FULL 1
#include "apa2_units.hpp"
#include "apa2_full_kernel.hpp"

namespace pa {
namespace apa2 {

hipError_t launch_apa2_full_kernel(int grid, hipStream_t s, const FullJob* jobs, const int32_t* order, int npairs, const FullParams& sp, uint32_t* ticket,
                                   uint32_t* err, uint32_t* dbg, unsigned long long* probe_stats, const RdvParams& rp, unsigned long long* rdv_stats) {
    hipLaunchKernelGGL(apa2_full_kernel, dim3(grid), dim3(64 * kStripBlockWaves), 0, s, jobs, order, npairs, sp, ticket, err, dbg, probe_stats, rp, rdv_stats);
    return hipGetLastError();
}

hipError_t launch_gcsh_probe_kernel(hipStream_t s, const FullJob* jobs, const int32_t* q, int nq, int32_t* out, uint32_t* err) {
    hipLaunchKernelGGL(gcsh_probe_kernel, dim3(1), dim3(64), 0, s, jobs, q, nq, out, err);
    return hipGetLastError();
}

}  // namespace apa2
}  // namespace pa
