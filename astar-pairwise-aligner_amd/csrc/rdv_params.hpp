// rdv_params.hpp -- launch parameters of the half-wave rendezvous (rdv_logic.hpp, strip2_kernel.hpp), shared by host and kernels.
#pragma once
#include <stdint.h>

namespace pa {

struct RdvParams {
    uint32_t enabled;   // 0: every strip runs alone
    uint32_t patience;  // ticks of the 100 MHz clock a posted strip waits for a partner before it runs alone
    uint32_t prio;      // 1: the issue priority of a wavefront follows its pair's rank in the start order (the most expensive pairs -- the
                        // launch's critical path -- are served first by their SIMDs; the cheap pairs that fill the rest have the slack)
    uint32_t search_windows;  // apa2_full_kernel: 1 = the h probes of the search go through two register windows of 64 layer records
};

// s_setprio takes an immediate
#define PA_PRIO_OF_RANK(t, npairs) ((uint32_t)(t) < (uint32_t)(npairs) / 8u ? 3 : ((uint32_t)(t) < (uint32_t)(npairs) / 4u ? 2 : ((uint32_t)(t) < (uint32_t)(npairs) / 2u ? 1 : 0)))
#define PA_SETPRIO_BY_RANK(t, npairs)                                               \
    do {                                                                            \
        const uint32_t t_ = (uint32_t)(t), n_ = (uint32_t)(npairs);                 \
        if (t_ < n_ / 8u) __builtin_amdgcn_s_setprio(3);                            \
        else if (t_ < n_ / 4u) __builtin_amdgcn_s_setprio(2);                       \
        else if (t_ < n_ / 2u) __builtin_amdgcn_s_setprio(1);                       \
        else __builtin_amdgcn_s_setprio(0);                                         \
    } while (0)

}  // namespace pa
