// apa2_units.hpp -- the band-search kernels of the batched A*PA2 are compiled in translation units of their own (apa2_simple_unit.hip,
// apa2_full_unit.hip, gcsh_build_unit.hip: hipcc spends a minute on each, side by side instead of one after the other inside pa_hip.hip);
// pa_hip.hip sees their descriptors (apa2_jobs.hpp) and
// launches them through these functions.  Each returns hipGetLastError() of its launch.
#pragma once
#include <hip/hip_runtime.h>

#include "apa2_jobs.hpp"

namespace pa {
namespace apa2 {

hipError_t launch_apa2_kernel(int grid, hipStream_t s, const PairJob* jobs, const int32_t* order, int npairs, const SearchParams& sp, uint32_t* ticket,
                              uint32_t* err, uint32_t* dbg, int k1_only);
hipError_t launch_apa2_full_kernel(int grid, hipStream_t s, const FullJob* jobs, const int32_t* order, int npairs, const FullParams& sp, uint32_t* ticket,
                                   uint32_t* err, uint32_t* dbg, unsigned long long* probe_stats);
hipError_t launch_gcsh_probe_kernel(hipStream_t s, const FullJob* jobs, const int32_t* q, int nq, int32_t* out, uint32_t* err);
hipError_t launch_gcsh_build_kernel(int grid, hipStream_t s, const GcshBuildJob* jobs, int npairs, uint32_t* ticket);

}  // namespace apa2
}  // namespace pa
