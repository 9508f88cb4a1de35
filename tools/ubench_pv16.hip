// Would the P.V product of the flash kernel be cheaper as 16x16x32 MFMAs?  At head_dim 40 the product pads
// its 41 output rows (40 + the ones row) to 64 with 32x32x16 MFMAs (4 per 32-key x 32-query unit) and to 48
// with 16x16x32 MFMAs (6 per unit, half the work each: 25 % less matrix work), but the probabilities leave
// the logits MFMA in the 32x32 operand layout and need 4 v_permlane16_swap_b32 per unit to reach the
// 16x16x32 one.  This models one pipelined slot both ways with the kernel's dependency structure:
//   A: 2 P.V + 3 K.Q' + 2 P.V  (32x32x16)                 + 16 exp, 8 max3, 8 cvt_pk
//   B: 3 P.V + 3 K.Q' + 3 P.V  (P.V as 16x16x32)          + the same + 4 permlane16_swap
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_pv16.hip -o ab/ubench_pv16 && ab/ubench_pv16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define MFMA_ACC(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_ZERO(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
#define MFMA16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define EXP(d, s) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(s))
#define MAX3(x, y, z) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define SWAP(a, b) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b))

// Operand bits: near-constant values (little switching in the multipliers: the clock stays high and the figures
// compare CYCLES) or, with rnd, a pair of bf16 with random sign and mantissa in [0.5, 2) per lane and register (the
// power-limited regime the real kernel runs in: the figures then compare what the clock management leaves).
__device__ __forceinline__ unsigned operand_bits(int rnd, unsigned plain, unsigned salt) {
    if (!rnd) return plain;
    unsigned h = (threadIdx.x * 2654435761u) ^ (salt * 40503u + 0x9e3779b9u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    return (h & 0x80ff80ffu) | 0x3f003f00u;
}

// MODE 0: the kernel's slot (A).  MODE 1: P.V as six 16x16x32 MFMAs plus the swaps (B).  MODE 3: mixed (C).
// MODE 2: B without the swaps (the matrix side alone).  EXTRA (bits): 1 LDS reads, 2 barrier, 4 scalar work as in the kernel.
template <int MODE, int EXTRA>
__global__ void __launch_bounds__(512) bench(float* out, int iters, int rnd) {
    f32x16 O0, O1, SA, SB;
    f32x4 o16[6];
    for (int i = 0; i < 16; ++i) { O0[i] = O1[i] = SA[i] = SB[i] = 0.f; }
    for (int j = 0; j < 6; ++j) for (int i = 0; i < 4; ++i) o16[j][i] = 0.f;
    u32x4 kf[3], qf[3], vf[4], PA[2], PB[2];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 3; ++j) { kf[j][i] = operand_bits(rnd, 0x3c003c00u + threadIdx.x + j, 4 * j + i); qf[j][i] = operand_bits(rnd, 0x3c003c00u + i + j, 64 + 4 * j + i); }
        for (int j = 0; j < 4; ++j) vf[j][i] = operand_bits(rnd, 0x3c003c00u + i * j, 128 + 4 * j + i);
        for (int j = 0; j < 2; ++j) { PA[j][i] = PB[j][i] = 0x3c003c00u; }
    }
    float e[16], m = 0.f;
    for (int i = 0; i < 16; ++i) e[i] = 0.f;
    __shared__ u32x4 smem[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) smem[i] = u32x4{1u, 2u, 3u, 4u};
    __syncthreads();
    const unsigned laddr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + (threadIdx.x & 63) * 16;
    u32x4 fr[7];
    for (int i = 0; i < 7; ++i) fr[i] = u32x4{0u, 0u, 0u, 0u};
    unsigned sc0 = blockIdx.x, sc1 = 3;
    auto ldsread = [&](int i) { if (EXTRA & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(fr[i]) : "v"(laddr) : "memory"); };
    auto salu = [&](int n) { if (EXTRA & 4) for (int i = 0; i < n; ++i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sc0) : "s"(sc1) : "scc"); };

    auto slot = [&](f32x16& Sc, f32x16& Sn, u32x4 (&Pc)[2], u32x4 (&Pp)[2], int u) {
        if (EXTRA & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the 32 vector ops of a unit in the kernel's order, as a flat list of 32 steps
        auto vop = [&](int k) {
            const int g = k / 4, r = k % 4;          // group g of 8: exp, exp, max3, cvt(previous group)
            if (r == 0) EXP(e[2 * g], Sc[2 * g]);
            else if (r == 1) EXP(e[2 * g + 1], Sc[2 * g + 1]);
            else if (r == 2) MAX3(m, Sc[2 * g], Sc[2 * g + 1]);
            else { const int c = (g + 7) % 8; CVT(Pc[c / 4][c % 4], e[2 * c], e[2 * c + 1]); }
        };
        auto vops = [&](int a, int b) { for (int k = a; k < b; ++k) vop(k); };
        if (MODE == 0) {
            MFMA_ACC(O0, vf[0], Pp[0]); vops(0, 5);
            MFMA_ACC(O1, vf[1], Pp[0]); vops(5, 10);
            if (!(u & 1)) { ldsread(3); ldsread(4); }
            salu(5);
            MFMA_ZERO(Sn, kf[0], qf[0]); vops(10, 14);
            MFMA_ACC(Sn, kf[1], qf[1]); vops(14, 19);
            MFMA_ACC(Sn, kf[2], qf[2]); vops(19, 23);
            if (!(u & 1)) { ldsread(0); ldsread(1); ldsread(2); }
            salu(5);
            MFMA_ACC(O0, vf[2], Pp[1]); vops(23, 28);
            MFMA_ACC(O1, vf[3], Pp[1]); vops(28, 32);
            if (!(u & 1)) { ldsread(5); ldsread(6); }
            salu(5);
        } else if (MODE == 3) {
            // mixed: output rows 0..31 as two 32x32x16 MFMAs (one per k-step), rows 32..47 as two 16x16x32 MFMAs
            // (one per query half, on the swapped probabilities): 5 x 32 + 2 x 16 = 192 matrix-pipe cycles per
            // unit instead of 224, the same seven MFMA issues, four swaps more on the vector port
            MFMA_ACC(O0, vf[0], Pp[0]); vops(0, 5);
            MFMA_ACC(O0, vf[2], Pp[1]); vops(5, 10);
            SWAP(Pp[0][0], Pp[1][0]); SWAP(Pp[0][1], Pp[1][1]); SWAP(Pp[0][2], Pp[1][2]); SWAP(Pp[0][3], Pp[1][3]);
            if (!(u & 1)) { ldsread(3); ldsread(4); }
            salu(5);
            MFMA_ZERO(Sn, kf[0], qf[0]); vops(10, 14);
            MFMA_ACC(Sn, kf[1], qf[1]); vops(14, 19);
            MFMA_ACC(Sn, kf[2], qf[2]); vops(19, 23);
            if (!(u & 1)) { ldsread(0); ldsread(1); ldsread(2); }
            salu(5);
            MFMA16(o16[0], vf[1], Pp[0]); vops(23, 28);
            MFMA16(o16[1], vf[3], Pp[1]); vops(28, 32);
            if (!(u & 1)) { ldsread(5); ldsread(6); }
            salu(5);
        } else {
            // P.V of the previous unit: query half 0 reads Pp[0], half 1 reads Pp[1] (after the swaps)
            MFMA16(o16[0], vf[0], Pp[0]); vops(0, 4);
            MFMA16(o16[1], vf[1], Pp[0]); vops(4, 8);
            MFMA16(o16[2], vf[2], Pp[0]); vops(8, 11);
            if (!(u & 1)) { ldsread(3); ldsread(4); }
            salu(5);
            MFMA_ZERO(Sn, kf[0], qf[0]); vops(11, 15);
            MFMA_ACC(Sn, kf[1], qf[1]); vops(15, 19);
            MFMA_ACC(Sn, kf[2], qf[2]); vops(19, 23);
            if (!(u & 1)) { ldsread(0); ldsread(1); ldsread(2); }
            salu(5);
            MFMA16(o16[3], vf[0], Pp[1]); vops(23, 26);
            MFMA16(o16[4], vf[1], Pp[1]); vops(26, 29);
            MFMA16(o16[5], vf[2], Pp[1]); vops(29, 32);
            if (MODE == 1) { SWAP(Pc[0][0], Pc[1][0]); SWAP(Pc[0][1], Pc[1][1]); SWAP(Pc[0][2], Pc[1][2]); SWAP(Pc[0][3], Pc[1][3]); }
            if (!(u & 1)) { ldsread(5); }
            salu(5);
        }
    };
    for (int it = 0; it < iters; ++it) {
        slot(SA, SB, PA, PB, 0);
        slot(SB, SA, PB, PA, 1);
        slot(SA, SB, PA, PB, 2);
        if (EXTRA & 2) __syncthreads();
        slot(SB, SA, PB, PA, 3);
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = m + (float)sc0;
    for (int i = 0; i < 7; ++i) s += (float)(fr[i][0] + fr[i][3]);
    for (int i = 0; i < 16; ++i) s += O0[i] + O1[i] + SA[i] + SB[i] + e[i];
    for (int j = 0; j < 6; ++j) for (int i = 0; i < 4; ++i) s += o16[j][i];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 4; ++i) s += (float)(PA[j][i] + PB[j][i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// D: EVERY product as 16x16x32 MFMAs.  K.Q' of a 32-key x 32-query unit is four 16x16 tiles of two k-steps each
// (head_dim 40 pads to 64 here, against 48 with the 32x32x16 form), P.V is six (41 rows pad to 48): 8 + 6 = 14 MFMAs
// of 16 matrix-pipe cycles = the same 224 cycles per unit as the kernel's seven of 32.  With the V image's keys
// permuted to the order the logits accumulators hold them (key 4g + i of tile j -> k index 8g + 4j + i), the
// accumulators of the two key tiles of a query half, converted, ARE that half's P.V operand: no swaps.  Same vector
// work (16 exp, 8 max3, 8 cvt_pk per lane and unit), same fragment reads (4 K + 3 V per two units).
#define MFMA16_ZERO(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b))
template <int EXTRA>
__global__ void __launch_bounds__(512) bench16(float* out, int iters, int rnd) {
    f32x4 SA[4], SB[4], o16[6];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) SA[j][i] = SB[j][i] = 0.f;
    for (int j = 0; j < 6; ++j) for (int i = 0; i < 4; ++i) o16[j][i] = 0.f;
    u32x4 kf[4], qf[4], vf[3], PA[2], PB[2];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) { kf[j][i] = operand_bits(rnd, 0x3c003c00u + threadIdx.x + j, 4 * j + i); qf[j][i] = operand_bits(rnd, 0x3c003c00u + i + j, 64 + 4 * j + i); }
        for (int j = 0; j < 3; ++j) vf[j][i] = operand_bits(rnd, 0x3c003c00u + i * j, 128 + 4 * j + i);
        for (int j = 0; j < 2; ++j) { PA[j][i] = PB[j][i] = 0x3c003c00u; }
    }
    float e[16], m = 0.f;
    for (int i = 0; i < 16; ++i) e[i] = 0.f;
    __shared__ u32x4 smem[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) smem[i] = u32x4{1u, 2u, 3u, 4u};
    __syncthreads();
    const unsigned laddr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + (threadIdx.x & 63) * 16;
    u32x4 fr[7];
    for (int i = 0; i < 7; ++i) fr[i] = u32x4{0u, 0u, 0u, 0u};
    unsigned sc0 = blockIdx.x, sc1 = 3;
    auto ldsread = [&](int i) { if (EXTRA & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(fr[i]) : "v"(laddr) : "memory"); };
    auto salu = [&](int n) { if (EXTRA & 4) for (int i = 0; i < n; ++i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sc0) : "s"(sc1) : "scc"); };
    auto slot = [&](f32x4 (&Sc)[4], f32x4 (&Sn)[4], u32x4 (&Pc)[2], u32x4 (&Pp)[2], int u) {
        if (EXTRA & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // logit i of the unit sits in tile i / 4 (tiles 0, 1: query half 0; tiles 2, 3: query half 1), register i % 4
        auto vop = [&](int k) {
            const int g = k / 4, r = k % 4;
            const int a = 2 * g, b = 2 * g + 1;
            if (r == 0) EXP(e[a], Sc[a / 4][a % 4]);
            else if (r == 1) EXP(e[b], Sc[b / 4][b % 4]);
            else if (r == 2) MAX3(m, Sc[a / 4][a % 4], Sc[b / 4][b % 4]);
            else { const int c = (g + 7) % 8; CVT(Pc[c / 4][c % 4], e[2 * c], e[2 * c + 1]); }
        };
        auto vops = [&](int a, int b) { for (int k = a; k < b; ++k) vop(k); };
        MFMA16(o16[0], vf[0], Pp[0]); vops(0, 2);
        MFMA16(o16[1], vf[1], Pp[0]); vops(2, 5);
        MFMA16(o16[2], vf[2], Pp[0]); vops(5, 7);
        if (!(u & 1)) { ldsread(4); ldsread(5); }
        salu(5);
        MFMA16_ZERO(Sn[0], kf[0], qf[0]); vops(7, 9);
        MFMA16(Sn[0], kf[1], qf[1]); vops(9, 12);
        MFMA16_ZERO(Sn[1], kf[2], qf[0]); vops(12, 14);
        MFMA16(Sn[1], kf[3], qf[1]); vops(14, 16);
        MFMA16_ZERO(Sn[2], kf[0], qf[2]); vops(16, 18);
        MFMA16(Sn[2], kf[1], qf[3]); vops(18, 21);
        MFMA16_ZERO(Sn[3], kf[2], qf[2]); vops(21, 23);
        MFMA16(Sn[3], kf[3], qf[3]); vops(23, 25);
        if (!(u & 1)) { ldsread(0); ldsread(1); ldsread(2); ldsread(3); }
        salu(5);
        MFMA16(o16[3], vf[0], Pp[1]); vops(25, 27);
        MFMA16(o16[4], vf[1], Pp[1]); vops(27, 30);
        MFMA16(o16[5], vf[2], Pp[1]); vops(30, 32);
        if (!(u & 1)) { ldsread(6); }
        salu(5);
    };
    for (int it = 0; it < iters; ++it) {
        slot(SA, SB, PA, PB, 0);
        slot(SB, SA, PB, PA, 1);
        slot(SA, SB, PA, PB, 2);
        if (EXTRA & 2) __syncthreads();
        slot(SB, SA, PB, PA, 3);
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = m + (float)sc0;
    for (int i = 0; i < 7; ++i) s += (float)(fr[i][0] + fr[i][3]);
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) s += SA[j][i] + SB[j][i];
    for (int i = 0; i < 16; ++i) s += e[i];
    for (int j = 0; j < 6; ++j) for (int i = 0; i < 4; ++i) s += o16[j][i];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 4; ++i) s += (float)(PA[j][i] + PB[j][i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* g_out;
static int g_rnd = 0;

template <int EXTRA>
void run16(const char* name) {
    const int grid = 256, iters = 4000;
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 2; ++i) bench16<EXTRA><<<grid, threads>>>(g_out, iters, g_rnd);
        (void)hipEventRecord(e0);
        bench16<EXTRA><<<grid, threads>>>(g_out, iters, g_rnd);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s waves/SIMD %d: %7.1f ns per stage (4 slots) per wave\n", name, threads / 256, ms * 1e6 / iters / (threads / 256));
    }
}

template <int MODE, int EXTRA>
void run(const char* name) {
    const int grid = 256, iters = 4000;
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < 2; ++i) bench<MODE, EXTRA><<<grid, threads>>>(g_out, iters, g_rnd);
        (void)hipEventRecord(e0);
        bench<MODE, EXTRA><<<grid, threads>>>(g_out, iters, g_rnd);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s waves/SIMD %d: %7.1f ns per stage (4 slots) per wave\n", name, threads / 256, ms * 1e6 / iters / (threads / 256));
    }
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    (void)hipMalloc(&g_out, 256 * 512 * sizeof(float));
    if (argc > 1) {      // any argument: the kernel's slot against the all-16x16x32 one, near-constant and random operands
        for (int rep = 0; rep < 2; ++rep)
            for (int rnd = 0; rnd < 2; ++rnd) {
                g_rnd = rnd;
                printf("-- %s operands\n", rnd ? "random" : "near-constant");
                run<0, 0>("A  P.V as 4 x 32x32x16 (the kernel's slot)");
                run16<0>("D  all 16x16x32: 8 K.Q' + 6 P.V, V keys permuted, no swaps");
                run<0, 7>("A  + LDS reads, barrier, scalar work");
                run16<7>("D  + LDS reads, barrier, scalar work");
                run<1, 7>("B  P.V as 6 x 16x16x32 + 4 swaps, + LDS reads, barrier, scalar work");
                run<3, 7>("C  P.V rows 0..31 32x32x16, rows 32..47 16x16x32 + 4 swaps, + extras");
            }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("A  P.V as 4 x 32x32x16 (the kernel's slot)");
        run16<0>("D  all 16x16x32: 8 K.Q' + 6 P.V, V keys permuted, no swaps");
        run<0, 7>("A  + LDS reads, barrier, scalar work");
        run16<7>("D  + LDS reads, barrier, scalar work");
        run<1, 0>("B  P.V as 6 x 16x16x32 + 4 permlane16_swap");
        run<2, 0>("B  without the swaps");
        run<0, 7>("A  + LDS reads, barrier, scalar work");
        run<1, 7>("B  + LDS reads, barrier, scalar work");
        run<3, 0>("C  P.V rows 0..31 as 32x32x16, rows 32..47 as 16x16x32 + 4 swaps");
        run<3, 7>("C  + LDS reads, barrier, scalar work");
        run<0, 1>("A  + LDS reads only");
        run<0, 2>("A  + barrier only");
        run<0, 4>("A  + scalar work only");
        run<0, 3>("A  + LDS reads, barrier");
    }
    return 0;
}
