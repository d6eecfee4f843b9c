// slice_plan.hpp -- host side of the bit-sliced full-DP kernel (slice_kernel.hpp): the plan of a cost-only batch as GROUPS of 32 pairs.
// Compiled in slice_unit.hip (a translation unit of its own, like the band-search kernels: apa2_units.hpp); pa_hip.hip decides when a
// batch runs this way (choose_batch_shape) and calls these functions from pa_batch_create / pa_batch_run.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace pa {
namespace slice {

struct Plan;

// Rows per lane the kernel is instantiated for (two wavefronts per SIMD: 4 R + ~40 VGPRs <= 256; even: the rows are stepped in asm blocks of four, then
// one of two).
constexpr int kRowsPerLane[] = {52, 50, 48, 46, 44, 42, 40, 36, 32, 28};

// Which R (0 = do not use the sliced kernel) and the estimated time of one pass in ns, for ranking against the other batch shapes.
// A batch qualifies when it has enough pairs to fill the chip with (group, strip) jobs.
int choose_rows_per_lane(const size_t* a_len, const size_t* b_len, size_t pairs, double simds, double* est_ns);

// code_off / prof_off: element offsets of every pair into the batch's packed codes (u32) and profile (pairs of u64), as PairDesc has them.
Plan* create(const size_t* a_len, const size_t* b_len, size_t pairs, const size_t* code_off, const size_t* prof_off, int rows_per_lane);
void destroy(Plan* p);

// Queues one pass on `s`: transposes (from the codes / profile the batch's encode kernels have just written), boundary rows reset, the
// kernel (bracketed by ev0 / ev1), the score kernel.  d_costs[pair] -- ZERO before the call -- receives the distance of every pair with two non-empty sequences;
// d_ticket_err: two u32, zeroed here; [1] != 0 afterwards = a bounded poll expired (slice::kErrSpin).
int run(Plan* p, hipStream_t s, const uint32_t* d_codes, const uint64_t* d_prof, int32_t* d_costs, uint32_t* d_ticket_err, hipEvent_t ev0, hipEvent_t ev1);

struct Info {
    int rows_per_lane;
    size_t groups, jobs;          // jobs = (group, strip) units
    double valu_instructions;     // wavefront VALU instructions of the DP kernel per pass (ISA model: (8 R + kStepOverhead) per strip step)
    double computed_rows_cells;   // cells actually computed (rows padded to whole strips, columns to the group's longest a)
    double device_bytes;          // the plan's own device memory
    double boundary_bytes;        // of those, the boundary rows that are reset before every pass
};
Info info(const Plan* p);
constexpr int kStepOverheadInstr = 24;  // VALU instructions of a strip step outside the rows (ISA count: DPP, predicates, addresses, the poll test)

}  // namespace slice
}  // namespace pa
