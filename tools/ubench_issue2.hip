// Second issue-rate microbenchmark for gfx950: one group = [one MFMA] + K vector ops of one kind on
// independent registers; how many cycles does a group take as K grows, with one and two waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue2.hip -o exp/ubench_issue2 && exp/ubench_issue2
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define MFMA32(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA16(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

enum { ADD = 0, EXP = 1, CVT = 2, MAX3 = 3, PKADD = 4, MOV = 5 };

template <int OP>
__device__ __forceinline__ void vop(float& x, float y, float z) {
    if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (OP == CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
    if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
}

// MF: 0 none, 1 = 32x32x16 (8 passes), 2 = 16x16x32 (4 passes... per its own shape)
template <int MF, int K, int OP>
__global__ void __launch_bounds__(512) bench(float* out, int iters) {
    f32x16 acc[4];
    f32x4 acc4[4];
    for (int c = 0; c < 4; ++c) {
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        for (int i = 0; i < 4; ++i) acc4[c][i] = 0.f;
    }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)(i * 0.01f); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = -0.001f * (threadIdx.x + i);
    float y = 0.5f, z = 0.25f;
    asm volatile("" : "+v"(y), "+v"(z));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (MF == 1) MFMA32(acc[g & 3]);
            if (MF == 2) MFMA16(acc4[g & 3]);
#pragma unroll
            for (int k = 0; k < K; ++k) vop<OP>(x[(g * K + k) & 15], y, z);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0.f;
    for (int c = 0; c < 4; ++c) {
        for (int i = 0; i < 16; ++i) s += acc[c][i];
        for (int i = 0; i < 4; ++i) s += acc4[c][i];
    }
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* g_out;
static double g_ghz = 2.3;

template <int MF, int K, int OP>
double run(int threads) {
    const int grid = 256, iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) bench<MF, K, OP><<<grid, threads>>>(g_out, iters);
    (void)hipEventRecord(e0);
    bench<MF, K, OP><<<grid, threads>>>(g_out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (iters * 16.0);    // ns per group (all waves of a SIMD together)
}

template <int MF, int OP>
void sweep(const char* name) {
    for (int threads : {256, 512}) {
        const int w = threads / 256;
        double r[9];
        r[0] = run<MF, 0, OP>(threads); r[1] = run<MF, 1, OP>(threads); r[2] = run<MF, 2, OP>(threads);
        r[3] = run<MF, 3, OP>(threads); r[4] = run<MF, 4, OP>(threads); r[5] = run<MF, 5, OP>(threads);
        r[6] = run<MF, 6, OP>(threads); r[7] = run<MF, 8, OP>(threads); r[8] = run<MF, 12, OP>(threads);
        printf("%-22s waves/SIMD %d | cycles per group per wave, K=0,1,2,3,4,5,6,8,12:", name, w);
        for (int i = 0; i < 9; ++i) printf(" %6.1f", r[i] * g_ghz / w);
        printf("\n");
    }
}

int main() {
    (void)hipMalloc(&g_out, 256 * 512 * sizeof(float));
    // calibrate the clock: back-to-back 32x32x16 MFMAs take 32 cycles each
    for (int i = 0; i < 3; ++i) run<1, 0, ADD>(256);
    g_ghz = 32.0 / run<1, 0, ADD>(256);
    printf("clock from MFMA calibration: %.2f GHz\n", g_ghz);
    sweep<0, ADD>("no mfma + add");
    sweep<0, EXP>("no mfma + exp");
    sweep<0, CVT>("no mfma + cvt_pk");
    sweep<0, MAX3>("no mfma + max3");
    sweep<0, MOV>("no mfma + mov");
    sweep<1, ADD>("mfma32 + add");
    sweep<1, EXP>("mfma32 + exp");
    sweep<1, CVT>("mfma32 + cvt_pk");
    sweep<1, MAX3>("mfma32 + max3");
    sweep<2, ADD>("mfma16 + add");
    sweep<2, EXP>("mfma16 + exp");
    return 0;
}
