// rdv_params.hpp -- launch parameters of the half-wave rendezvous (rdv_logic.hpp, strip2_kernel.hpp), shared by host and kernels.
#pragma once
#include <stdint.h>

namespace pa {

struct RdvParams {
    uint32_t enabled;   // 0: every strip runs alone
    uint32_t patience;  // ticks of the 100 MHz clock a posted strip waits for a partner before it runs alone
};

}  // namespace pa
