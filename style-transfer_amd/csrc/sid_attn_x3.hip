// Instantiation unit: attention kernels with split-bf16 (hi + lo) operands, three MFMA products
// per matrix product, f32-grade accuracy (SID_COMPUTE_F32X3).
#include "sid_attn_kernel.hpp"

namespace sid_attn {
int launch_flash_x3(const FlashParams& fp, const Geometry& g, hipStream_t stream) {
    return launch_flash<true>(fp, g, stream);
}
}  // namespace sid_attn
