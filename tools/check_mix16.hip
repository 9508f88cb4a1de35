// Layout check for a mixed P.V product (one wave): the probabilities of a 32-key x 32-query unit as the flash kernel
// holds them (B operands of two 32x32x16 k-steps), turned into the B operands of two 16x16x32 MFMAs (one per query
// half) with v_permlane16_swap, against 16 rows of V^T read from the kernel's fragment image in the matching key order.
//   hipcc --offload-arch=gfx950 -O2 tools/check_mix16.hip -o tools/_bin/check_mix16 && tools/_bin/check_mix16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// actual key of MFMA k index (k-step s, lane half hh, element j) in the kernel's probability registers
__host__ __device__ inline int key_of(int s, int hh, int j) { return 16 * s + (j < 4 ? 4 * hh + j : 8 + 4 * hh + (j - 4)); }

__global__ void k(const float* P, const float* Vt, float* out) {       // P[32 keys][32 queries], Vt[16 rows][32 keys]
    const int l = threadIdx.x, q = l % 32, hh = l / 32;
    bf16x8 pk[2];
    for (int s = 0; s < 2; ++s)
        for (int j = 0; j < 8; ++j) pk[s][j] = (__bf16)P[key_of(s, hh, j) * 32 + q];
    // the kernel's V^T fragment units of dv block 1 (k-step s): lane l' holds row l' % 32, lane half l' / 32
    __shared__ u32x4 unit[2][64];
    for (int s = 0; s < 2; ++s) {
        bf16x8 v;
        const int row = l % 32;
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)(row < 16 ? Vt[row * 32 + key_of(s, hh, j)] : 0.f);
        unit[s][l] = __builtin_bit_cast(u32x4, v);
    }
    __syncthreads();
    const int rho = l / 16, c = l % 16;
    const bf16x8 a = __builtin_bit_cast(bf16x8, unit[rho % 2][32 * (rho / 2) + c]);
    u32x4 x = __builtin_bit_cast(u32x4, pk[0]), y = __builtin_bit_cast(u32x4, pk[1]);
    for (int j = 0; j < 4; ++j) {
        auto r = __builtin_amdgcn_permlane16_swap(x[j], y[j], false, false);
        x[j] = r[0]; y[j] = r[1];
    }
    f32x4 o0 = {0, 0, 0, 0}, o1 = {0, 0, 0, 0};
    o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, x), o0, 0, 0, 0);
    o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, y), o1, 0, 0, 0);
    for (int i = 0; i < 4; ++i) {       // lane: query 16 h + l % 16, rows 4 (l / 16) + i
        out[(4 * (l / 16) + i) * 32 + (l % 16)] = o0[i];
        out[(4 * (l / 16) + i) * 32 + 16 + (l % 16)] = o1[i];
    }
}

int main() {
    float hP[32 * 32], hV[16 * 32], ref[16 * 32], got[16 * 32];
    srand(1);
    for (auto& v : hP) v = (float)(rand() % 7 - 3);
    for (auto& v : hV) v = (float)(rand() % 5 - 2);
    for (int r = 0; r < 16; ++r)
        for (int q = 0; q < 32; ++q) {
            float s = 0;
            for (int kk = 0; kk < 32; ++kk) s += hV[r * 32 + kk] * hP[kk * 32 + q];
            ref[r * 32 + q] = s;
        }
    float *dP, *dV, *dO;
    (void)hipMalloc(&dP, sizeof hP); (void)hipMalloc(&dV, sizeof hV); (void)hipMalloc(&dO, sizeof got);
    (void)hipMemcpy(dP, hP, sizeof hP, hipMemcpyHostToDevice); (void)hipMemcpy(dV, hV, sizeof hV, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dP, dV, dO);
    (void)hipMemcpy(got, dO, sizeof got, hipMemcpyDeviceToHost);
    float err = 0;
    for (int i = 0; i < 16 * 32; ++i) err = fmaxf(err, fabsf(got[i] - ref[i]));
    printf("mixed P.V layout check: max |error| %g (%s)\n", err, err == 0.f ? "exact" : "MISMATCH");
    return err == 0.f ? 0 : 1;
}
