// Masked AdaIN initial latent:  out = m * adain(cnt, char) + (1 - m) * adain(cnt, back),
//   adain(c, s) = ((c - mean_c) / std_c) * std_s + mean_s     per channel over (batch, H, W), unbiased std,
//   m = bilinear resize (align_corners=False) of the raw [0,1] content mask to (H, W).
//
// One workgroup per channel: the latents are (1,4,64,64), so the whole problem is 3 x 64 KiB of
// input.  Two passes over the channel for each tensor (mean, then centred sum of squares), then the
// element-wise blend in the operation order of the reference expressions.
#include "sid_common.h"

#include <cstdint>

namespace {

template <bool BF16>
__device__ __forceinline__ float ld(const void* p, size_t i) {
    if (BF16) return (float)((const __bf16*)p)[i];
    return ((const float*)p)[i];
}

__device__ __forceinline__ void taps(int dst, int in_size, float scale, int* i0, int* i1, float* w0, float* w1) {
    float src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
    if (src < 0.f) src = 0.f;
    int a = (int)floorf(src);
    if (a > in_size - 1) a = in_size - 1;
    float l1 = __fsub_rn(src, (float)a);
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    *i0 = a;
    *i1 = a + (a < in_size - 1 ? 1 : 0);
    *w1 = l1;
    *w0 = __fsub_rn(1.0f, l1);
}

// sum of three values over the 256 threads of the workgroup, result in every thread
__device__ __forceinline__ void block_sum3(float (&v)[3], float (*sm)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sm[wave][0] = v[0]; sm[wave][1] = v[1]; sm[wave][2] = v[2]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = (sm[0][k] + sm[1][k]) + (sm[2][k] + sm[3][k]);
}

template <bool BF16>
__global__ void __launch_bounds__(256) adain_kernel(const void* __restrict__ cnt, const void* __restrict__ s1,
                                                    const void* __restrict__ s2, const float* __restrict__ mask,
                                                    int Hin, int Win, float sh, float sw, void* __restrict__ out,
                                                    int B, int C, int H, int W) {
    __shared__ float sm[4][3];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int HW = H * W, n = B * HW;
    const void* src[3] = {cnt, s1, s2};
    auto idx = [&](int e) { return ((size_t)(e / HW) * C + c) * HW + (e % HW); };

    float mean[3], sd[3], acc[3] = {0.f, 0.f, 0.f};
    for (int e = tid; e < n; e += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += ld<BF16>(src[k], idx(e));
    block_sum3(acc, sm);
#pragma unroll
    for (int k = 0; k < 3; ++k) { mean[k] = acc[k] / (float)n; acc[k] = 0.f; }
    for (int e = tid; e < n; e += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float dlt = ld<BF16>(src[k], idx(e)) - mean[k]; acc[k] += dlt * dlt; }
    block_sum3(acc, sm);
#pragma unroll
    for (int k = 0; k < 3; ++k) sd[k] = sqrtf(acc[k] / (float)(n - 1));      // torch.std: unbiased

    for (int e = tid; e < n; e += 256) {
        const int p = e % HW, oy = p / W, ox = p % W;
        float m = 1.0f;
        if (mask) {
            if (Hin == H && Win == W) {
                m = mask[p];
            } else {
                int y0, y1, x0, x1;
                float wy0, wy1, wx0, wx1;
                taps(oy, Hin, sh, &y0, &y1, &wy0, &wy1);
                taps(ox, Win, sw, &x0, &x1, &wx0, &wx1);
                const float a = mask[(size_t)y0 * Win + x0], b = mask[(size_t)y0 * Win + x1];
                const float cc = mask[(size_t)y1 * Win + x0], d = mask[(size_t)y1 * Win + x1];
                const float top = __fadd_rn(__fmul_rn(wx0, a), __fmul_rn(wx1, b));
                const float bot = __fadd_rn(__fmul_rn(wx0, cc), __fmul_rn(wx1, d));
                m = __fadd_rn(__fmul_rn(wy0, top), __fmul_rn(wy1, bot));
            }
        }
        const size_t i = idx(e);
        const float z = __fdiv_rn(__fsub_rn(ld<BF16>(cnt, i), mean[0]), sd[0]);
        const float a1 = __fadd_rn(__fmul_rn(z, sd[1]), mean[1]);
        float r = a1;
        if (mask) {
            const float a2 = __fadd_rn(__fmul_rn(z, sd[2]), mean[2]);
            r = __fadd_rn(__fmul_rn(m, a1), __fmul_rn(__fsub_rn(1.0f, m), a2));
        }
        if (BF16) ((__bf16*)out)[i] = (__bf16)r;
        else ((float*)out)[i] = r;
    }
}

}  // namespace

extern "C" int sid_masked_adain(const void* cnt, const void* sty_char, const void* sty_back, const float* mask,
                                int Hin, int Win, void* out, int B, int C, int H, int W, int dtype, void* stream) {
    SID_CHECK_ARG(cnt && sty_char && out, "NULL tensor pointer");
    SID_CHECK_ARG((mask == nullptr) == (sty_back == nullptr), "mask and sty_back must both be given (blend) or both be NULL (plain AdaIN)");
    SID_CHECK_ARG(B >= 1 && C >= 1 && H >= 1 && W >= 1, "B, C, H, W must be >= 1");
    SID_CHECK_ARG((long long)B * H * W >= 2, "the unbiased std needs at least two elements per channel");
    SID_CHECK_ARG((long long)B * C * H * W < (1LL << 31), "tensor too large");
    SID_CHECK_ARG(mask == nullptr || (Hin >= 1 && Win >= 1), "mask size must be >= 1");
    SID_CHECK_ARG(dtype == SID_DTYPE_F32 || dtype == SID_DTYPE_BF16, "unknown dtype %d", dtype);
    const float sh = mask ? (float)Hin / (float)H : 1.f, sw = mask ? (float)Win / (float)W : 1.f;
    const void* s2 = sty_back ? sty_back : sty_char;
    if (dtype == SID_DTYPE_BF16)
        hipLaunchKernelGGL(adain_kernel<true>, dim3(C), dim3(256), 0, (hipStream_t)stream, cnt, sty_char, s2, mask, Hin, Win, sh, sw, out, B, C, H, W);
    else
        hipLaunchKernelGGL(adain_kernel<false>, dim3(C), dim3(256), 0, (hipStream_t)stream, cnt, sty_char, s2, mask, Hin, Win, sh, sw, out, B, C, H, W);
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}
