// Cycles per MFMA, back-to-back issue, one wave per SIMD: v_mfma_f32_32x32x16_bf16 against the K = 8 form
// v_mfma_f32_32x32x8_bf16_1k (would a head_dim of 40 = 16 + 16 + 8 pay for 2.5 instead of 3 k-steps?).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma8.hip -o ab/ubench_mfma8 && ab/ubench_mfma8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int MODE>
__global__ void __launch_bounds__(256) bench(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    u32x4 a = {0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
    u32x2 a2 = {a[0], a[1]}, b2 = a2;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x8bf16_1k %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a2), "v"(b2));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    (void)hipMallocManaged(&cyc, 8);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        bench<0><<<256, 256>>>(out, iters, cyc); (void)hipDeviceSynchronize();
        printf("v_mfma_f32_32x32x16_bf16   : %.1f cycles per MFMA\n", (double)*cyc / (4.0 * iters));
        bench<1><<<256, 256>>>(out, iters, cyc); (void)hipDeviceSynchronize();
        printf("v_mfma_f32_32x32x8_bf16_1k : %.1f cycles per MFMA\n", (double)*cyc / (4.0 * iters));
    }
    return 0;
}
