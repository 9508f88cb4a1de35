// Microbenchmark of the flash kernel's pipelined slot on gfx950, with its real dependency structure:
//   7 MFMAs (2 P.V, 3 chained K.Q', 2 P.V) each followed by a share of the softmax vector ops
//   (16 exp, 8 max3, 8 cvt_pk, reading the logits the previous slot produced, writing the packed
//   probabilities the next slot's P.V reads).  Variants switch single dependencies off to see which
//   one costs issue cycles.  One wave per SIMD (256 threads) or two (512).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_slot.hip -o exp/ubench_slot && exp/ubench_slot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define MFMA_ACC(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_ZERO(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
#define EXP(d, s) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(s))
#define MAX3(x, y, z) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))

// VARIANT bits: 1 = K.Q' chain on three different accumulators (no MFMA->MFMA dependency)
//               2 = vector ops read private registers instead of the logits (no MFMA->VALU dependency)
//               4 = cvt writes private registers instead of P (no VALU->MFMA dependency)
//               8 = no vector ops at all          16 = no MFMAs at all
//               32 = plus the kernel's LDS fragment reads (14 ds_read_b128 per stage, waited for one slot later)
//               64 = plus a workgroup barrier per stage
//               128 = plus about 60 scalar instructions per stage (address arithmetic, loop control)
template <int VARIANT>
__global__ void __launch_bounds__(512) bench(float* out, int iters) {
    f32x16 O0, O1, SA, SB, SX, SY;
    for (int i = 0; i < 16; ++i) { O0[i] = O1[i] = SA[i] = SB[i] = SX[i] = SY[i] = 0.f; }
    u32x4 kf[3], qf[3], vf[4], PA[2], PB[2], PX[2];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 3; ++j) { kf[j][i] = 0x3c003c00u + threadIdx.x + j; qf[j][i] = 0x3c003c00u + i + j; }
        for (int j = 0; j < 4; ++j) vf[j][i] = 0x3c003c00u + i * j;
        for (int j = 0; j < 2; ++j) { PA[j][i] = PB[j][i] = PX[j][i] = 0x3c003c00u; }
    }
    float priv[16], e[16], m = 0.f;
    for (int i = 0; i < 16; ++i) { priv[i] = -0.01f * i; e[i] = 0.f; }
    __shared__ u32x4 smem[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) smem[i] = u32x4{1u, 2u, 3u, 4u};
    __syncthreads();
    const unsigned laddr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + (threadIdx.x & 63) * 16;
    u32x4 fr[7];
    for (int i = 0; i < 7; ++i) fr[i] = u32x4{0u, 0u, 0u, 0u};
    unsigned sc0 = blockIdx.x, sc1 = 3;
    auto ldsread = [&](int i, int off) {
        if (VARIANT & 32) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[i]) : "v"(laddr), "n"(0) : "memory");
        (void)off;
    };
    auto salu = [&](int n) {
        if (VARIANT & 128)
            for (int i = 0; i < n; ++i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sc0) : "s"(sc1) : "scc");
    };

    auto slot = [&](f32x16& Sc, f32x16& Sn, u32x4 (&Pc)[2], u32x4 (&Pp)[2], int u) {
        if (VARIANT & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        auto src = [&](int i) -> float { return (VARIANT & 2) ? priv[i] : Sc[i]; };
        auto chunk = [&](int c) {
            if (VARIANT & 8) return;
            // 32 ops over 7 chunks: chunk c does exps 2c.. etc.
            switch (c) {
                case 0: EXP(e[0], src(0)); EXP(e[1], src(1)); MAX3(m, src(0), src(1)); EXP(e[2], src(2)); EXP(e[3], src(3)); break;
                case 1: MAX3(m, src(2), src(3)); CVT(((VARIANT & 4) ? PX : Pc)[0][0], e[0], e[1]); EXP(e[4], src(4)); EXP(e[5], src(5)); MAX3(m, src(4), src(5)); break;
                case 2: CVT(((VARIANT & 4) ? PX : Pc)[0][1], e[2], e[3]); EXP(e[6], src(6)); EXP(e[7], src(7)); MAX3(m, src(6), src(7)); CVT(((VARIANT & 4) ? PX : Pc)[0][2], e[4], e[5]); break;
                case 3: EXP(e[8], src(8)); EXP(e[9], src(9)); MAX3(m, src(8), src(9)); CVT(((VARIANT & 4) ? PX : Pc)[0][3], e[6], e[7]); EXP(e[10], src(10)); break;
                case 4: EXP(e[11], src(11)); MAX3(m, src(10), src(11)); CVT(((VARIANT & 4) ? PX : Pc)[1][0], e[8], e[9]); EXP(e[12], src(12)); EXP(e[13], src(13)); break;
                case 5: MAX3(m, src(12), src(13)); CVT(((VARIANT & 4) ? PX : Pc)[1][1], e[10], e[11]); EXP(e[14], src(14)); EXP(e[15], src(15)); MAX3(m, src(14), src(15)); break;
                default: CVT(((VARIANT & 4) ? PX : Pc)[1][2], e[12], e[13]); CVT(((VARIANT & 4) ? PX : Pc)[1][3], e[14], e[15]); break;
            }
        };
        const bool mf = !(VARIANT & 16);
        if (mf) MFMA_ACC(O0, vf[0], Pp[0]); chunk(0);
        if (mf) MFMA_ACC(O1, vf[1], Pp[0]); chunk(1);
        if (!(u & 1)) { ldsread(3, 0); ldsread(4, 0); }
        salu(5);
        if (VARIANT & 1) {
            if (mf) MFMA_ZERO(Sn, kf[0], qf[0]); chunk(2);
            if (mf) MFMA_ZERO(SX, kf[1], qf[1]); chunk(3);
            if (mf) MFMA_ZERO(SY, kf[2], qf[2]); chunk(4);
        } else {
            if (mf) MFMA_ZERO(Sn, kf[0], qf[0]); chunk(2);
            if (mf) MFMA_ACC(Sn, kf[1], qf[1]); chunk(3);
            if (mf) MFMA_ACC(Sn, kf[2], qf[2]); chunk(4);
        }
        if (!(u & 1)) { ldsread(0, 0); ldsread(1, 0); ldsread(2, 0); }
        salu(5);
        if (mf) MFMA_ACC(O0, vf[2], Pp[1]); chunk(5);
        if (mf) MFMA_ACC(O1, vf[3], Pp[1]); chunk(6);
        if (!(u & 1)) { ldsread(5, 0); ldsread(6, 0); }
        salu(5);
    };
    for (int it = 0; it < iters; ++it) {
        slot(SA, SB, PA, PB, 0);
        slot(SB, SA, PB, PA, 1);
        slot(SA, SB, PA, PB, 2);
        if (VARIANT & 64) __syncthreads();
        slot(SB, SA, PB, PA, 3);
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = m;
    for (int i = 0; i < 7; ++i) s += (float)(fr[i][0] + fr[i][3]);
    s += (float)sc0;
    for (int i = 0; i < 16; ++i) s += O0[i] + O1[i] + SA[i] + SB[i] + SX[i] + SY[i] + e[i] + priv[i];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 4; ++i) s += (float)(PA[j][i] + PB[j][i] + PX[j][i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* g_out;

template <int VARIANT>
void run(const char* name) {
    const int grid = 256, iters = 4000;
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 2; ++i) bench<VARIANT><<<grid, threads>>>(g_out, iters);
        (void)hipEventRecord(e0);
        bench<VARIANT><<<grid, threads>>>(g_out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s waves/SIMD %d: %7.1f ns per stage (4 slots) per wave\n", name, threads / 256, ms * 1e6 / iters / (threads / 256));
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    (void)hipMalloc(&g_out, 256 * 512 * sizeof(float));
    run<8>("MFMAs only (28 per stage)");
    run<16>("vector ops only");
    run<0>("slot as in the kernel");
    run<1>("  K.Q' chain on separate accumulators");
    run<2>("  vector ops on private registers");
    run<4>("  cvt into private registers");
    run<6>("  both (no MFMA<->VALU dependencies)");
    run<7>("  no dependencies at all");
    run<32>("slot + LDS fragment reads");
    run<64>("slot + barrier per stage");
    run<128>("slot + 60 scalar instructions per stage");
    run<32 + 64 + 128>("slot + all three");
    run<32 + 8>("LDS reads + MFMAs only");
    run<32 + 16>("LDS reads + vector ops only");
    return 0;
}
