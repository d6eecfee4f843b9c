// apa2_simple_unit.hip -- translation unit of pa::apa2::apa2_kernel (apa2_kernel.hpp): the batched band search of the `simple` family.
#define PA_UNIT_APA2_SIMPLE 1
#include "apa2_units.hpp"
#include "apa2_kernel.hpp"

namespace pa {
namespace apa2 {

hipError_t launch_apa2_kernel(int grid, hipStream_t s, const PairJob* jobs, const int32_t* order, int npairs, const SearchParams& sp, uint32_t* ticket,
                              uint32_t* err, uint32_t* dbg, int k1_only, const RdvParams& rp, unsigned long long* rdv_stats) {
    hipLaunchKernelGGL(apa2_kernel, dim3(grid), dim3(64 * kStripBlockWaves), 0, s, jobs, order, npairs, sp, ticket, err, dbg, k1_only, rp, rdv_stats);
    return hipGetLastError();
}

}  // namespace apa2
}  // namespace pa
