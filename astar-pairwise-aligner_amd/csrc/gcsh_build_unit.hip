// gcsh_build_unit.hip -- translation unit of pa::apa2::gcsh_build_kernel (gcsh_build_kernel.hpp): the matches of GCSH found on the GPU.
#define PA_UNIT_GCSH_BUILD 1
#include "apa2_units.hpp"
#include "gcsh_build_kernel.hpp"

namespace pa {
namespace apa2 {

hipError_t launch_gcsh_build_kernel(int grid, hipStream_t s, const GcshBuildJob* jobs, int npairs, uint32_t* ticket) {
    hipLaunchKernelGGL(gcsh_build_kernel, dim3(grid), dim3(64), 0, s, jobs, npairs, ticket);
    return hipGetLastError();
}

}  // namespace apa2
}  // namespace pa
