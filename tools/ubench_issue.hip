// Issue-rate microbenchmark for gfx950: how well do the matrix pipe (v_mfma_f32_32x32x16_bf16) and the
// vector ALU (v_exp_f32, v_max3_f32, v_cvt_pk_bf16_f32) overlap, within one wave and across the two
// waves of a SIMD?  The instruction mix per iteration is that of one steady-state stage of the
// head_dim-40 flash kernel: 28 MFMA, 64 exponentials, 40 max, 32 packed conversions.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o exp/ubench_issue && exp/ubench_issue
//
// Every instruction is volatile asm, so the order in the binary is the order written here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define MAX3(x, y, z) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))

// PATTERN 0: MFMA only   1: VALU only   2: MFMA block, then VALU block   3: 1 MFMA : 5 VALU interleaved
// 4: exp only (64)   5: max only (40)   6: cvt only (32)
template <int PATTERN>
__global__ void __launch_bounds__(512) bench(float* out, long long* cycles, int iters) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)(i * 0.01f); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = -0.001f * (threadIdx.x + i);
    float m = 0.f;
    unsigned pk[8];
    for (int i = 0; i < 8; ++i) pk[i] = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (PATTERN == 0 || PATTERN == 2) {
#pragma unroll
            for (int j = 0; j < 28; ++j) MFMA(acc[j & 3]);
        }
        if (PATTERN == 1 || PATTERN == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 16; ++i) EXP(x[i]);
#pragma unroll
                for (int i = 0; i < 16; i += 2) MAX3(m, x[i], x[i + 1]);
                MAX3(m, x[0], x[1]); MAX3(m, x[2], x[3]);
#pragma unroll
                for (int i = 0; i < 8; ++i) CVT(pk[i], x[2 * i], x[2 * i + 1]);
            }
        }
        if (PATTERN == 3) {
            // 28 groups: MFMA then ~5 VALU (64 exp + 40 max + 32 cvt = 136 = 28 * 5 - 4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                MFMA(acc[0]); EXP(x[0]); EXP(x[1]); EXP(x[2]); EXP(x[3]); MAX3(m, x[0], x[1]);
                MFMA(acc[1]); EXP(x[4]); EXP(x[5]); EXP(x[6]); EXP(x[7]); MAX3(m, x[2], x[3]);
                MFMA(acc[2]); EXP(x[8]); EXP(x[9]); EXP(x[10]); EXP(x[11]); MAX3(m, x[4], x[5]);
                MFMA(acc[3]); EXP(x[12]); EXP(x[13]); EXP(x[14]); EXP(x[15]); MAX3(m, x[6], x[7]);
                MFMA(acc[0]); MAX3(m, x[8], x[9]); MAX3(m, x[10], x[11]); MAX3(m, x[12], x[13]); MAX3(m, x[14], x[15]); MAX3(m, x[0], x[1]);
                MFMA(acc[1]); MAX3(m, x[2], x[3]); CVT(pk[0], x[0], x[1]); CVT(pk[1], x[2], x[3]); CVT(pk[2], x[4], x[5]); CVT(pk[3], x[6], x[7]);
                MFMA(acc[2]); CVT(pk[4], x[8], x[9]); CVT(pk[5], x[10], x[11]); CVT(pk[6], x[12], x[13]); CVT(pk[7], x[14], x[15]);
            }
        }
        if (PATTERN == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) EXP(x[i]);
        }
        if (PATTERN == 5) {
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int i = 0; i < 16; i += 2) MAX3(m, x[i], x[i + 1]);
        }
        if (PATTERN == 6) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) CVT(pk[i], x[2 * i], x[2 * i + 1]);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = m;
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += (float)pk[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int PATTERN>
void run(const char* name, int threads, float* out, long long* cyc, int iters) {
    const int grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) bench<PATTERN><<<grid, threads>>>(out, cyc, iters);   // clock ramp
    hipEventRecord(e0);
    bench<PATTERN><<<grid, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < grid; ++i) avg += h[i];
    avg /= grid;
    // s_memtime counts at a constant 100 MHz on this part; derive shader cycles from the event time too
    printf("%-28s waves/SIMD %d  %8.1f ns/iter  (counter %.1f ticks/iter)\n", name, threads / 256, ms * 1e6 / iters,
           avg / iters);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, 256 * sizeof(long long));
    const int iters = 20000;
    for (int threads : {256, 512}) {
        run<0>("mfma only (28)", threads, out, cyc, iters);
        run<4>("exp only (64)", threads, out, cyc, iters);
        run<5>("max3 only (40)", threads, out, cyc, iters);
        run<6>("cvt_pk only (32)", threads, out, cyc, iters);
        run<1>("valu only (64+40+32)", threads, out, cyc, iters);
        run<2>("mfma block + valu block", threads, out, cyc, iters);
        run<3>("interleaved 1:5", threads, out, cyc, iters);
    }
    return 0;
}
