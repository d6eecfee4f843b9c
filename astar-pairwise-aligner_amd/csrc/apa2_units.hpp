// apa2_units.hpp -- the band-search kernels of the batched A*PA2 are compiled in translation units of their own (apa2_simple_unit.hip,
// apa2_full_unit.hip, gcsh_build_unit.hip: hipcc spends a minute on each, side by side instead of one after the other inside pa_hip.hip);
// pa_hip.hip sees their descriptors (apa2_jobs.hpp) and
// launches them through these functions.  Each returns hipGetLastError() of its launch.
#pragma once
#include <hip/hip_runtime.h>

#include "apa2_jobs.hpp"
#include "rdv_params.hpp"

namespace pa {
namespace apa2 {

hipError_t launch_apa2_kernel(int grid, hipStream_t s, const PairJob* jobs, const int32_t* order, int npairs, const SearchParams& sp, uint32_t* ticket,
                              uint32_t* err, uint32_t* dbg, int k1_only, const RdvParams& rp, unsigned long long* rdv_stats);
hipError_t launch_apa2_full_kernel(int grid, hipStream_t s, const FullJob* jobs, const int32_t* order, int npairs, const FullParams& sp, uint32_t* ticket,
                                   uint32_t* err, uint32_t* dbg, unsigned long long* probe_stats, const RdvParams& rp, unsigned long long* rdv_stats);
hipError_t launch_gcsh_probe_kernel(hipStream_t s, const FullJob* jobs, const int32_t* q, int nq, int32_t* out, uint32_t* err);
hipError_t launch_gcsh_build_kernel(int grid, hipStream_t s, const GcshBuildJob* jobs, int npairs, uint32_t* ticket);

// sketch_unit.hip: found_out[pair] = how many of 64 sampled 16-mers of a occur in b near the diagonal (an estimate of (1 - e)^16 in 64ths);
// the same layout as pa_hip.hip's PairDesc (element offsets into the concatenated sequences)
struct SketchDesc {
    unsigned long long a_off, b_off, code_off, prof_off;
    int n, m;
};
hipError_t launch_sketch_kernel(hipStream_t s, const uint8_t* a_cat, const uint8_t* b_cat, const SketchDesc* desc, int npairs, uint8_t* found_out);

}  // namespace apa2
}  // namespace pa
