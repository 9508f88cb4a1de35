// Content mask -> signed per-token mask: threshold, sign, bilinear resize.
//
// One thread per output pixel.  The arithmetic is float32 with every product and sum rounded
// separately (no FMA contraction), nested as h0*(w0*a + w1*b) + h1*(w0*c + w1*d), so that for the
// masks this path is fed (binary, power-of-two reduction: taps weigh 0.5/0.5) the result is exact
// and identical to any other evaluation order.
#include "sid_common.h"

namespace {

__device__ __forceinline__ float thresh(float v, float ch) {
    // <0.5 -> -1, >0.5 -> +1, exactly 0.5 (and NaN) unchanged; then the sign channel
    float t = v;
    if (v < 0.5f) t = -1.0f;
    if (v > 0.5f) t = 1.0f;
    return __fmul_rn(t, ch);
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

__global__ void __launch_bounds__(256) mask_downsample_kernel(const float* __restrict__ mask, int Hin, int Win,
                                                              int Hout, int Wout, float ch, float sh, float sw,
                                                              float* __restrict__ out) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= Hout * Wout) return;
    const int oy = gid / Wout, ox = gid % Wout;
    float r;
    if (Hin == Hout && Win == Wout) {
        r = thresh(mask[gid], ch);
    } else {
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        taps(oy, Hin, sh, &y0, &y1, &wy0, &wy1);
        taps(ox, Win, sw, &x0, &x1, &wx0, &wx1);
        const float a = thresh(mask[(size_t)y0 * Win + x0], ch), b = thresh(mask[(size_t)y0 * Win + x1], ch);
        const float c = thresh(mask[(size_t)y1 * Win + x0], ch), d = thresh(mask[(size_t)y1 * Win + x1], ch);
        const float top = __fadd_rn(__fmul_rn(wx0, a), __fmul_rn(wx1, b));
        const float bot = __fadd_rn(__fmul_rn(wx0, c), __fmul_rn(wx1, d));
        r = __fadd_rn(__fmul_rn(wy0, top), __fmul_rn(wy1, bot));
    }
    out[gid] = r;
}

}  // namespace

extern "C" int sid_mask_downsample(const float* mask, int Hin, int Win, int Hout, int Wout, float ch, float* out,
                                   void* stream) {
    SID_CHECK_ARG(mask && out, "NULL pointer");
    SID_CHECK_ARG(Hin >= 1 && Win >= 1 && Hout >= 1 && Wout >= 1, "sizes must be >= 1");
    SID_CHECK_ARG((long long)Hin * Win < (1LL << 31) && (long long)Hout * Wout < (1LL << 31), "mask too large");
    SID_CHECK_ARG(ch == 1.0f || ch == -1.0f, "ch must be +1 or -1 (got %g)", (double)ch);
    const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
    const int n = Hout * Wout;
    hipLaunchKernelGGL(mask_downsample_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mask, Hin,
                       Win, Hout, Wout, ch, sh, sw, out);
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}
