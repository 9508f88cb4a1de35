// Second issue-rate microbenchmark for gfx950: one group = [one MFMA] + K vector ops of one kind on
// independent registers; how many cycles does a group take as K grows, with one and two waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue2.hip -o exp/ubench_issue2 && exp/ubench_issue2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define MFMA32(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA16(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

enum { ADD = 0, EXP = 1, CVT = 2, MAX3 = 3, PKADD = 4, MOV = 5, DSREAD = 6, DSREAD64 = 7,
       EXP16 = 8, EXP16HI = 9, EXPLEG = 10, CVTRTZ = 11, LDEXP = 12, PKMUL16 = 13 };
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int OP>
__device__ __forceinline__ void vop(float& x, float y, float z) {
    if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (OP == CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
    if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
    // half-precision exponentials (low half; high half in place through SDWA), the legacy f32 form, and what a
    // half-precision softmax would need around them
    if (OP == EXP16) asm volatile("v_exp_f16 %0, %0" : "+v"(x));
    if (OP == EXP16HI) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(x));
    if (OP == EXPLEG) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(x));
    if (OP == CVTRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
    if (OP == LDEXP) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == PKMUL16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x) : "v"(y));
}

// MF: 0 none, 1 = 32x32x16 (8 passes), 2 = 16x16x32 (4 passes... per its own shape)
template <int MF, int K, int OP>
__global__ void __launch_bounds__(512) bench(float* out, int iters, int rnd) {
    f32x16 acc[4];
    f32x4 acc4[4];
    for (int c = 0; c < 4; ++c) {
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        for (int i = 0; i < 4; ++i) acc4[c][i] = 0.f;
    }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)(i * 0.01f); }
    if (rnd) {      // random sign and mantissa, magnitude in [0.5, 2): full switching in the multipliers
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        u4 ua, ub;
        for (int i = 0; i < 4; ++i) {
            unsigned h = (threadIdx.x * 2654435761u) ^ ((i + 1) * 40503u + 0x9e3779b9u);
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
            ua[i] = (h & 0x80ff80ffu) | 0x3f003f00u;
            h *= 2654435761u; h ^= h >> 15;
            ub[i] = (h & 0x80ff80ffu) | 0x3f003f00u;
        }
        a = __builtin_bit_cast(bf16x8, ua); b = __builtin_bit_cast(bf16x8, ub);
    }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = -0.001f * (threadIdx.x + i);
    float y = 0.5f, z = 0.25f;
    asm volatile("" : "+v"(y), "+v"(z));
    __shared__ u32x4 smem[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) smem[i] = u32x4{1u, 2u, 3u, 4u};
    __syncthreads();
    const unsigned laddr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    u32x4 dsr[4];
    u32x2 dsr2[4];
    for (int i = 0; i < 4; ++i) { dsr[i] = u32x4{0u, 0u, 0u, 0u}; dsr2[i] = u32x2{0u, 0u}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (MF == 1) MFMA32(acc[g & 3]);
            if (MF == 2) MFMA16(acc4[g & 3]);
            if (MF == 3) {   // four exponentials (about 28 issue cycles of vector work) instead of an MFMA
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[0])); asm volatile("v_exp_f32 %0, %0" : "+v"(x[1]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[2])); asm volatile("v_exp_f32 %0, %0" : "+v"(x[3]));
            }
            if (MF == 4) {   // eight max3 (about 32 issue cycles)
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[q]) : "v"(y), "v"(z));
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (OP == DSREAD) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(dsr[(g * K + k) & 3]) : "v"(laddr));
                else if (OP == DSREAD64) asm volatile("ds_read_b64 %0, %1 offset:8192" : "=v"(dsr2[(g * K + k) & 3]) : "v"(laddr));
                else vop<OP>(x[(g * K + k) & 15], y, z);
            }
            if (OP == DSREAD || OP == DSREAD64) { if ((g & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0.f;
    for (int c = 0; c < 4; ++c) {
        for (int i = 0; i < 16; ++i) s += acc[c][i];
        for (int i = 0; i < 4; ++i) s += acc4[c][i];
    }
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += (float)(dsr[i][0] + dsr[i][3] + dsr2[i][0]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* g_out;
static double g_ghz = 2.3;
static int g_rnd = 0;

template <int MF, int K, int OP>
double run(int threads) {
    const int grid = 256, iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) bench<MF, K, OP><<<grid, threads>>>(g_out, iters, g_rnd);
    (void)hipEventRecord(e0);
    bench<MF, K, OP><<<grid, threads>>>(g_out, iters, g_rnd);
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

int main(int argc, char** argv) {
    (void)hipMalloc(&g_out, 256 * 512 * sizeof(float));
    // calibrate the clock: back-to-back 32x32x16 MFMAs take 32 cycles each
    for (int i = 0; i < 3; ++i) run<1, 0, ADD>(256);
    g_ghz = 32.0 / run<1, 0, ADD>(256);
    printf("clock from MFMA calibration: %.2f GHz\n", g_ghz);
    if (argc > 1 && argv[1][0] == 'p') {   // "peak": back-to-back MFMAs alone, near-constant against random operand bits
        for (int rnd = 0; rnd < 2; ++rnd) {
            g_rnd = rnd;
            for (int threads : {256, 512}) {
                // 4 SIMDs x 256 CUs; a 32x32x16 MFMA is 32768 flop, a 16x16x32 one 16384; ns per MFMA per SIMD
                const double n32 = run<1, 0, ADD>(threads), n16 = run<2, 0, ADD>(threads);
                printf("%s operands, %d wave(s) per SIMD: 32x32x16 %.2f ns per MFMA = %.0f TFLOP/s | 16x16x32 %.2f ns = %.0f TFLOP/s\n",
                       rnd ? "random" : "near-constant", threads / 256, n32 / (threads / 256), (threads / 256) * 1024 * 32768.0 / n32 / 1e3,
                       n16 / (threads / 256), (threads / 256) * 1024 * 16384.0 / n16 / 1e3);
            }
        }
        return 0;
    }
    if (argc > 1) {   // any other argument: only the half-precision exponential survey
        sweep<0, EXP>("no mfma + exp_f32");
        sweep<0, EXP16>("no mfma + exp_f16");
        sweep<0, EXP16HI>("no mfma + exp_f16 hi");
        sweep<0, EXPLEG>("no mfma + exp_legacy");
        sweep<0, CVTRTZ>("no mfma + cvt_pkrtz");
        sweep<0, LDEXP>("no mfma + ldexp");
        sweep<0, PKMUL16>("no mfma + pk_mul_f16");
        sweep<1, EXP>("mfma32 + exp_f32");
        sweep<1, EXP16>("mfma32 + exp_f16");
        sweep<1, EXP16HI>("mfma32 + exp_f16 hi");
        sweep<1, EXPLEG>("mfma32 + exp_legacy");
        return 0;
    }
    sweep<0, DSREAD>("no mfma + ds_read_b128");
    sweep<1, DSREAD>("mfma32 + ds_read_b128");
    sweep<0, DSREAD64>("no mfma + ds_read_b64");
    sweep<1, DSREAD64>("mfma32 + ds_read_b64");
    sweep<3, DSREAD>("4 exp + ds_read_b128");
    sweep<4, DSREAD>("8 max3 + ds_read_b128");
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
