// Instantiation unit: attention kernels with bf16 MFMA operands (SID_COMPUTE_BF16).
#include "sid_attn_kernel.hpp"

namespace sid_attn {
int launch_flash_bf16(const FlashParams& fp, const Geometry& g, hipStream_t stream) {
    return launch_flash<false>(fp, g, stream);
}
}  // namespace sid_attn
