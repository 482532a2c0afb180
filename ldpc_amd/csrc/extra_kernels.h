// Launchers of the kernels beside the belief-propagation decoders (bit flipping, encoder, fused
// sweep); kept out of bp_common.h so that adding one does not rebuild every decoder.
#pragma once
#include "bp_common.h"

namespace ldpc {

// bitflip.hip: ldpc.py:82-135
hipError_t launch_bitflip(const CsrDev &code, int max_iters, int num_active_checks, const Launch &io, int num_cus,
                          const char **kernel_name, bool *supported);

}  // namespace ldpc
