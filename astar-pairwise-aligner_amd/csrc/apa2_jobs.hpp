// apa2_jobs.hpp -- the per-pair descriptors the host fills in and the batched A*PA2 kernels read (apa2_kernel.hpp, apa2_full_kernel.hpp,
// gcsh_build_kernel.hpp).  Plain structs in a header of their own so that the host side (pa_hip.hip) does not depend on the kernels'
// source: the kernels live in translation units of their own (apa2_units.hpp).
#pragma once
#include <stdint.h>

#include "apa2_full_logic.hpp"
#include "apa2_logic.hpp"
#include "gcsh_dev.hpp"
#include "sweep_logic.hpp"

namespace pa {
namespace apa2 {

// ---- apa2_kernel.hpp -----------------------------------------------------------------------------------------------------------
struct PairJob {
    const uint32_t* a_codes;  // packed 2-bit codes of a
    const uint32_t* b_prof;   // BitProfile words of b, u32 view
    BlockRec* rec;            // [nblk + 1] persistent block records
    uint32_t* col;            // column store: slot k (block k's right-edge column) = col + k * col_stride * 4, indexed by absolute word
    int64_t col_stride;       // words per slot: the pair's window (sweep_logic.hpp SlotGeom), ceil(m / 64) = the full column
    uint32_t slot_ratio;      // SlotGeom::ratio
    uint32_t pad0;
    const int32_t* sh_h;      // SH: h(i) for i = 0..n, else nullptr
    uint64_t* gran;           // 2 rows x 8 granules, zero between uses
    int32_t* sum;             // scratch: bottom-row sum of the last strip
    PairResult* result;
    int32_t n, m;
};

// ---- apa2_full_kernel.hpp -----------------------------------------------------------------------------------------------------
enum : int32_t { kFullHeurNone = 0, kFullHeurGap = 1, kFullHeurSH = 2, kFullHeurGcsh = 3 };  // engine.hpp HeuristicKind

struct FullJob {
    const uint32_t* a_codes;  // packed 2-bit codes of a
    const uint32_t* b_prof;   // BitProfile words of b, u32 view
    BlockRec* rec;            // [nblk + 2] persistent block records (the traceback reads them: trace_kernel.hpp)
    int32_t* jh;              // [nblk + 2] row of the stored horizontal differences per block (Block::j_h), kNone: none
    uint32_t* col;            // column store: slot k = col + k * col_stride * 4, indexed by absolute word
    int64_t col_stride;       // words per slot: the pair's window (sweep_logic.hpp SlotGeom)
    uint8_t* hrow;            // [n] the stored row: one byte per column, bit0 = +1, bit1 = -1 (blocks.rs:103-105)
    const int32_t* sh_h;      // SH: h(i) for i = 0..n
    uint64_t* gran;           // 2 rows x 8 granules, zero between uses
    int32_t* sum;             // scratch: bottom-row sum of the last strip
    PairResult* result;
    GcshDev g;                // GCSH
    int32_t n, m, heur;
    uint32_t slot_ratio;      // SlotGeom::ratio
};

// ---- gcsh_build_kernel.hpp ----------------------------------------------------------------------------------------------------
constexpr int kBuildMaxP = 14;                    // local-pruning look-ahead the LDS arrays are sized for (the `full` preset's)
constexpr int kBuildFr = 2 * kBuildMaxP + 3;      // diagonals of a search + one sentinel either side
constexpr int32_t kBuildNeg = INT32_MIN;          // "no column yet" (prepruning.rs uses I::MIN)

struct GcshBuildJob {
    const uint8_t* a;   // ASCII, device
    const uint8_t* b;
    uint32_t* keys;     // [nseeds]   the k-mer of every seed (low 32 bits of the 2-bit packing, first character highest: qgrams.rs:30-43)
    int32_t* slot;      // [tsize]    hash table: the newest seed of a k-mer's chain, -1 empty
    int32_t* next_same; // [nseeds]   next seed with the same k-mer, -1
    int32_t* cnt;       // [nseeds + 1] candidates per seed, then their exclusive prefix
    int32_t* fill;      // [nseeds]
    int32_t* tmp_s;     // [cap] candidates in push order read backwards: rows ascending, seeds descending within a row
    int32_t* tmp_j;     // [cap]
    int32_t* gpos;      // [cap] position of candidate t in the by-start order
    int32_t* cj;        // [cap] by-start order: rows (the seed of position q is the one whose prefix range holds q)
    uint8_t* flag;      // [cap] by candidate t: 1 = kept
    uint8_t* keptg;     // [cap] by by-start position: 1 = kept
    int32_t* mi;        // out [cap] kept matches by start: columns
    int32_t* mj;        // out: rows
    GcshSeedWindow* win0;  // out [nseeds]
    int32_t* nmatch_out;   // out: &FullJob::g.nmatch of this pair
    uint32_t* status;      // out: 0 built, else why not (kBuild*)
    int32_t n, m, k, p, nseeds, tsize, cap, pad;
    unsigned long long* clocks;  // diagnostics (optional): 100 MHz ticks of phases A .. F, then the candidates, those kept alone, the searches of E
};
enum : uint32_t { kBuildOk = 0, kBuildOverflow = 1, kBuildRing = 2 };

}  // namespace apa2
}  // namespace pa
