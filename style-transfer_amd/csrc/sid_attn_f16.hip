// Instantiation unit: attention kernels with half-precision (f16) query / key operands and bf16
// probability / value operands (SID_COMPUTE_F16QK), every head-dim class.
#include "sid_attn_kernel.hpp"

namespace sid_attn {
int launch_flash_f16qk(const FlashParams& fp, const Geometry& g, hipStream_t stream) {
    return launch_flash<false, true>(fp, g, stream);
}
}  // namespace sid_attn
