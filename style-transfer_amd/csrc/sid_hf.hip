// High-frequency radius filter: LDS-resident 2-D FFT -> zero the low-frequency disc -> inverse FFT,
// fused with the patched ResBlock sum  out = skip + h + HF(h_cnt).
//
// One 256-thread workgroup filters TWO feature maps at once as one complex map z = a + i*b: the
// keep-mask is real and even in frequency, so it filters the real and imaginary parts
// independently and two real transforms cost one complex transform.  The whole H x W complex map
// stays in LDS (128x128: 129 KB); HBM sees each input once and the output once.
//
// Square 16/32/64/128 maps (the decoder ResBlocks of the path) use the radix-16/8/4 kernel below
// (hf2_kernel): each thread does whole radix-R butterflies in registers, and the map makes eight
// round trips through LDS.  Other shapes use the general kernel:
//
// A length-N transform runs across the lanes of a wave (one element per lane, two for N = 128) as
// radix-2 butterflies exchanged with wavefront shuffles.  Forward passes are decimation in
// frequency (natural in, bit-reversed out), inverse passes decimation in time (bit-reversed in,
// natural out), so no reordering pass exists: the mask is simply evaluated at the bit-reversed
// bin each lane holds.  No fftshift is needed either: bin k carries the signed frequency
// f = k (k < N/2) else k - N, and the disc test is fy^2 + fx^2 < radius^2.
#include "sid_common.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

namespace {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {   // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 shfl_xor2(float2 v, int m) {
    return make_float2(__shfl_xor(v.x, m), __shfl_xor(v.y, m));
}

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n / 2); }

// Length-N transform over the lanes of a wave.  T[k] = exp(-2*pi*i*k/128), k < 64.
template <int N>
struct LaneFFT {
    static constexpr int E = N > 64 ? N / 64 : 1;   // elements per lane (positions pos + 64e)
    static constexpr int L = N > 64 ? 64 : N;       // lanes per transform

    static __device__ __forceinline__ int bin(int p) { return (int)(__brev((unsigned)p) >> (32 - ilog2(N))); }

    static __device__ __forceinline__ void fwd(float2 (&v)[E], int pos, const float2* T) {
        if (E == 2) {
            const float2 a = v[0], b = v[E - 1];
            v[0] = cadd(a, b);
            v[E - 1] = cmul(csub(a, b), T[pos]);
        }
#pragma unroll
        for (int h = L / 2; h >= 1; h >>= 1) {
            const bool up = (pos & h) != 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float2 o = shfl_xor2(v[e], h);
                float2 s = up ? csub(o, v[e]) : cadd(v[e], o);
                if (h > 1) {
                    const float2 tw = T[(pos & (h - 1)) * (64 / h)];
                    const float2 st = cmul(s, tw);
                    s = up ? st : s;
                }
                v[e] = s;
            }
        }
    }

    static __device__ __forceinline__ void inv(float2 (&v)[E], int pos, const float2* T) {
#pragma unroll
        for (int h = 1; h <= L / 2; h <<= 1) {
            const bool up = (pos & h) != 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float2 x = v[e];
                if (h > 1) {
                    const float2 xt = cmulc(x, T[(pos & (h - 1)) * (64 / h)]);
                    x = up ? xt : x;
                }
                const float2 o = shfl_xor2(x, h);
                v[e] = up ? csub(o, x) : cadd(x, o);
            }
        }
        if (E == 2) {
            const float2 a = v[0], b = cmulc(v[E - 1], T[pos]);
            v[0] = cadd(a, b);
            v[E - 1] = csub(a, b);
        }
    }
};

template <bool BF16>
__device__ __forceinline__ float ld(const void* p, size_t i) {
    if (BF16) return (float)((const __bf16*)p)[i];
    return ((const float*)p)[i];
}

template <int H, int W, bool BF16>
__global__ void __launch_bounds__(256) hf_kernel(const void* __restrict__ hcnt, const void* __restrict__ skip,
                                                 const void* __restrict__ hres, void* __restrict__ out, int C,
                                                 int pairs_per_b, int radius2) {
    extern __shared__ __attribute__((aligned(16))) float2 sm[];
    float2* T = sm;
    float2* buf = sm + 64;
    constexpr int LDW = W + 1;
    using RF = LaneFFT<W>;
    using CF = LaneFFT<H>;
    constexpr int RPW = 64 / RF::L;   // rows a wave transforms at once
    constexpr int CPW = 64 / CF::L;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) {
        float s, c;
        sincospif(-(float)tid / 64.0f, &s, &c);
        T[tid] = make_float2(c, s);
    }
    const int b = blockIdx.x / pairs_per_b, pc = blockIdx.x % pairs_per_b;
    const int c0 = 2 * pc;
    const bool two = c0 + 1 < C;
    const size_t base_a = ((size_t)b * C + c0) * H * W;
    const size_t base_b = base_a + (size_t)H * W;
    __syncthreads();

    // rows, forward
    for (int rg = wave; rg < H / RPW; rg += 4) {
        const int y = rg * RPW + lane / RF::L, pos = lane % RF::L;
        float2 v[RF::E];
#pragma unroll
        for (int e = 0; e < RF::E; ++e) {
            const size_t o = (size_t)y * W + pos + 64 * e;
            v[e] = make_float2(ld<BF16>(hcnt, base_a + o), two ? ld<BF16>(hcnt, base_b + o) : 0.f);
        }
        RF::fwd(v, pos, T);
#pragma unroll
        for (int e = 0; e < RF::E; ++e) buf[y * LDW + pos + 64 * e] = v[e];
    }
    __syncthreads();

    // columns: forward, mask (and 1/(H*W) of the two ortho norms), inverse
    constexpr float norm = 1.0f / (float)(H * W);
    for (int cg = wave; cg < W / CPW; cg += 4) {
        const int x = cg * CPW + lane / CF::L, pos = lane % CF::L;
        float2 v[CF::E];
#pragma unroll
        for (int e = 0; e < CF::E; ++e) v[e] = buf[(pos + 64 * e) * LDW + x];
        CF::fwd(v, pos, T);
        const int kx = RF::bin(x);
        const int fx = kx < W / 2 ? kx : kx - W;
#pragma unroll
        for (int e = 0; e < CF::E; ++e) {
            const int ky = CF::bin(pos + 64 * e);
            const int fy = ky < H / 2 ? ky : ky - H;
            const float g = (fy * fy + fx * fx >= radius2) ? norm : 0.f;
            v[e] = make_float2(v[e].x * g, v[e].y * g);
        }
        CF::inv(v, pos, T);
#pragma unroll
        for (int e = 0; e < CF::E; ++e) buf[(pos + 64 * e) * LDW + x] = v[e];
    }
    __syncthreads();

    // rows, inverse; real part -> map a, imaginary part -> map b; fused residual sum
    for (int rg = wave; rg < H / RPW; rg += 4) {
        const int y = rg * RPW + lane / RF::L, pos = lane % RF::L;
        float2 v[RF::E];
#pragma unroll
        for (int e = 0; e < RF::E; ++e) v[e] = buf[y * LDW + pos + 64 * e];
        RF::inv(v, pos, T);
#pragma unroll
        for (int e = 0; e < RF::E; ++e) {
            const size_t o = (size_t)y * W + pos + 64 * e;
            if (BF16) {
                __bf16* op = (__bf16*)out;
                __bf16 ra = (__bf16)v[e].x, rb = (__bf16)v[e].y;
                if (skip) {
                    // bf16 tensors: (skip + h) rounds to bf16, then + HF rounds again
                    const __bf16 sa = (__bf16)(ld<true>(skip, base_a + o) + ld<true>(hres, base_a + o));
                    ra = (__bf16)((float)sa + (float)ra);
                    if (two) {
                        const __bf16 sb = (__bf16)(ld<true>(skip, base_b + o) + ld<true>(hres, base_b + o));
                        rb = (__bf16)((float)sb + (float)rb);
                    }
                }
                op[base_a + o] = ra;
                if (two) op[base_b + o] = rb;
            } else {
                float* op = (float*)out;
                float ra = v[e].x, rb = v[e].y;
                if (skip) {
                    ra = __fadd_rn(__fadd_rn(ld<false>(skip, base_a + o), ld<false>(hres, base_a + o)), ra);
                    if (two) rb = __fadd_rn(__fadd_rn(ld<false>(skip, base_b + o), ld<false>(hres, base_b + o)), rb);
                }
                op[base_a + o] = ra;
                if (two) op[base_b + o] = rb;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Square maps, N = R1 * R2 in {4*4, 8*4, 8*8, 8*16}: two radix stages per line, butterflies in registers.
//
// A line x[n], n = R2*n1 + n2, goes to X[k], k = k1 + R1*k2:
//   stage 1  for each n2: DFT_R1 over n1 (elements R2 apart), times W_N^(n2*k1), stored at R2*k1 + n2
//   stage 2  for each k1: DFT_R2 over n2 (adjacent elements),                    stored at R2*k1 + k2
// so position p holds bin k = p / R2 + R1 * (p % R2).  The inverse runs the same two stages backwards
// with conjugate factors and consumes exactly that order: no reordering pass, and the middle of the
// column transform (stage 2, mask, inverse stage 2) stays in registers.
// ------------------------------------------------------------------------------------------
template <bool FWD>
__device__ __forceinline__ float2 rot90(float2 a) {           // a * (-i) forward, a * (+i) inverse
    return FWD ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

template <int R, bool FWD>
struct Dft;

template <bool FWD>
struct Dft<4, FWD> {
    static __device__ __forceinline__ void run(float2 (&v)[4]) {
        const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        const float2 t2 = cadd(v[1], v[3]), t3 = rot90<FWD>(csub(v[1], v[3]));
        v[0] = cadd(t0, t2); v[1] = cadd(t1, t3); v[2] = csub(t0, t2); v[3] = csub(t1, t3);
    }
};

template <bool FWD>
struct Dft<8, FWD> {
    static __device__ __forceinline__ void run(float2 (&v)[8]) {
        float2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
        Dft<4, FWD>::run(e);
        Dft<4, FWD>::run(o);
        constexpr float h = 0.70710678118654752440f;
        const float2 a = o[1], c = o[3];
        // W8 = (1 - i)/sqrt2 forward, (1 + i)/sqrt2 inverse; W8^3 = (-1 - i)/sqrt2, (-1 + i)/sqrt2
        o[1] = FWD ? make_float2((a.x + a.y) * h, (a.y - a.x) * h) : make_float2((a.x - a.y) * h, (a.x + a.y) * h);
        o[2] = rot90<FWD>(o[2]);
        o[3] = FWD ? make_float2((c.y - c.x) * h, (-c.x - c.y) * h) : make_float2((-c.x - c.y) * h, (c.x - c.y) * h);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = cadd(e[k], o[k]); v[4 + k] = csub(e[k], o[k]); }
    }
};

template <bool FWD>
struct Dft<16, FWD> {
    static __device__ __forceinline__ void run(float2 (&v)[16]) {
        float2 e[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { e[k] = v[2 * k]; o[k] = v[2 * k + 1]; }
        Dft<8, FWD>::run(e);
        Dft<8, FWD>::run(o);
        // W16^k = exp(-+ 2 pi i k / 16)
        constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
        const float2 w[8] = {make_float2(1.f, 0.f), make_float2(c1, -s1), make_float2(h, -h), make_float2(s1, -c1),
                             make_float2(0.f, -1.f), make_float2(-s1, -c1), make_float2(-h, -h), make_float2(-c1, -s1)};
#pragma unroll
        for (int k = 1; k < 8; ++k) o[k] = FWD ? cmul(o[k], w[k]) : cmulc(o[k], w[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = cadd(e[k], o[k]); v[8 + k] = csub(e[k], o[k]); }
    }
};

template <bool BF16>
__device__ __forceinline__ void ld4(const void* p, size_t i, float (&v)[4]) {
    if (BF16) {
        const uint2 raw = *(const uint2*)((const __bf16*)p + i);
        v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
        v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
    } else {
        const float4 f = *(const float4*)((const float*)p + i);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    }
}

template <int N, int THREADS, bool BF16>
__global__ void __launch_bounds__(THREADS) hf2_kernel(const void* __restrict__ hcnt, const void* __restrict__ skip,
                                                      const void* __restrict__ hres, void* __restrict__ out, int C,
                                                      int pairs_per_b, int radius2) {
    constexpr int R1 = N == 16 ? 4 : 8, R2 = N / R1;
    constexpr int LD = N + 1;
    constexpr int TS = N > 64 ? N : 64;
    extern __shared__ __attribute__((aligned(16))) float2 hf2_lds[];
    float2* T = hf2_lds;                     // exp(-2*pi*i*j/TS), j < TS
    float2* buf = hf2_lds + TS;              // N rows of LD
    const int tid = threadIdx.x;
    if (tid < TS) {
        float s, c;
        sincospif(-2.0f * (float)tid / (float)TS, &s, &c);
        T[tid] = make_float2(c, s);
    }
    const int b = blockIdx.x / pairs_per_b, pc = blockIdx.x % pairs_per_b;
    const int c0 = 2 * pc;
    const bool two = c0 + 1 < C;
    const size_t base_a = ((size_t)b * C + c0) * N * N;
    const size_t base_b = base_a + (size_t)N * N;

    // ---- load: z = a + i b, rows contiguous, 4 elements per thread and step
    constexpr int STEPS = N * N / 4 / THREADS;
    static_assert(STEPS * THREADS * 4 == N * N, "every thread takes the same number of 4-element steps");
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const int i = tid + st * THREADS;
        const int y = (4 * i) / N, x = (4 * i) % N;
        float a[4], bb[4] = {0.f, 0.f, 0.f, 0.f};
        ld4<BF16>(hcnt, base_a + 4 * (size_t)i, a);
        if (two) ld4<BF16>(hcnt, base_b + 4 * (size_t)i, bb);
#pragma unroll
        for (int j = 0; j < 4; ++j) buf[y * LD + x + j] = make_float2(a[j], bb[j]);
    }
    // the residual operands of the epilogue are requested now, so that their latency passes behind the
    // transform instead of after it
    float pre[STEPS][4][4];
    if (skip) {
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const size_t o = 4 * (size_t)(tid + st * THREADS);
            ld4<BF16>(skip, base_a + o, pre[st][0]); ld4<BF16>(hres, base_a + o, pre[st][1]);
            if (two) { ld4<BF16>(skip, base_b + o, pre[st][2]); ld4<BF16>(hres, base_b + o, pre[st][3]); }
        }
    }
    __syncthreads();

    // twiddle W_N^e (forward) from the table
    auto tw = [&](int e) { return T[(e % N) * (TS / N)]; };

    // One pass over all lines.  ROWS: line = row, positions along x; else line = column.
    // Work item = (line, j): line fastest, so that a wave's lanes touch different lines.
    auto addr = [&](bool rows, int line, int p) { return rows ? line * LD + p : p * LD + line; };

    auto stage1_fwd = [&](bool rows) {
        for (int w = tid; w < N * R2; w += THREADS) {
            const int line = w % N, n2 = w / N;
            float2 v[R1];
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) v[n1] = buf[addr(rows, line, R2 * n1 + n2)];
            Dft<R1, true>::run(v);
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) buf[addr(rows, line, R2 * k1 + n2)] = k1 ? cmul(v[k1], tw(n2 * k1)) : v[k1];
        }
        __syncthreads();
    };
    auto stage2_fwd = [&](bool rows) {
        for (int w = tid; w < N * R1; w += THREADS) {
            const int line = w % N, k1 = w / N;
            float2 v[R2];
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) v[n2] = buf[addr(rows, line, R2 * k1 + n2)];
            Dft<R2, true>::run(v);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) buf[addr(rows, line, R2 * k1 + k2)] = v[k2];
        }
        __syncthreads();
    };
    auto stage2_inv = [&](bool rows) {
        for (int w = tid; w < N * R1; w += THREADS) {
            const int line = w % N, k1 = w / N;
            float2 v[R2];
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) v[k2] = buf[addr(rows, line, R2 * k1 + k2)];
            Dft<R2, false>::run(v);
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) buf[addr(rows, line, R2 * k1 + n2)] = (k1 && n2) ? cmulc(v[n2], tw(n2 * k1)) : v[n2];
        }
        __syncthreads();
    };
    auto stage1_inv = [&](bool rows) {
        for (int w = tid; w < N * R2; w += THREADS) {
            const int line = w % N, n2 = w / N;
            float2 v[R1];
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) v[k1] = buf[addr(rows, line, R2 * k1 + n2)];
            Dft<R1, false>::run(v);
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) buf[addr(rows, line, R2 * n1 + n2)] = v[n1];
        }
        __syncthreads();
    };

    stage1_fwd(true);
    stage2_fwd(true);
    stage1_fwd(false);
    // columns: stage 2, mask (with the 1/(N*N) of the two ortho norms), inverse stage 2 -- in registers
    {
        constexpr float norm = 1.0f / (float)(N * N);
        for (int w = tid; w < N * R1; w += THREADS) {
            const int col = w % N, k1 = w / N;
            const int kx = col / R2 + R1 * (col % R2);
            const int fx = kx < N / 2 ? kx : kx - N;
            float2 v[R2];
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) v[n2] = buf[(R2 * k1 + n2) * LD + col];
            Dft<R2, true>::run(v);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
                const int ky = k1 + R1 * k2;
                const int fy = ky < N / 2 ? ky : ky - N;
                const float g = (fy * fy + fx * fx >= radius2) ? norm : 0.f;
                v[k2] = make_float2(v[k2].x * g, v[k2].y * g);
            }
            Dft<R2, false>::run(v);
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) buf[(R2 * k1 + n2) * LD + col] = (k1 && n2) ? cmulc(v[n2], tw(n2 * k1)) : v[n2];
        }
        __syncthreads();
    }
    stage1_inv(false);
    stage2_inv(true);
    stage1_inv(true);

    // ---- store: real part -> map a, imaginary part -> map b; fused residual sum
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const int i = tid + st * THREADS;
        const int y = (4 * i) / N, x = (4 * i) % N;
        const size_t o = 4 * (size_t)i;
        float ra[4], rb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float2 z = buf[y * LD + x + j]; ra[j] = z.x; rb[j] = z.y; }
        if (skip) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // bf16 tensors: (skip + h) rounds to bf16, then + HF rounds again
                if (BF16) ra[j] = (float)(__bf16)(pre[st][0][j] + pre[st][1][j]) + (float)(__bf16)ra[j];
                else ra[j] = __fadd_rn(__fadd_rn(pre[st][0][j], pre[st][1][j]), ra[j]);
            }
            if (two) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (BF16) rb[j] = (float)(__bf16)(pre[st][2][j] + pre[st][3][j]) + (float)(__bf16)rb[j];
                    else rb[j] = __fadd_rn(__fadd_rn(pre[st][2][j], pre[st][3][j]), rb[j]);
                }
            }
        }
        if (BF16) {
            __bf16* op = (__bf16*)out;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                op[base_a + o + j] = (__bf16)ra[j];
                if (two) op[base_b + o + j] = (__bf16)rb[j];
            }
        } else {
            float* op = (float*)out;
            *(float4*)(op + base_a + o) = make_float4(ra[0], ra[1], ra[2], ra[3]);
            if (two) *(float4*)(op + base_b + o) = make_float4(rb[0], rb[1], rb[2], rb[3]);
        }
    }
}

template <int N, bool BF16>
int launch_hf2(const void* hcnt, const void* skip, const void* hres, void* out, int B, int C, int radius,
               hipStream_t stream) {
    constexpr int THREADS = N == 128 ? 512 : (N == 64 ? 512 : (N == 32 ? 256 : 64));
    constexpr size_t lds = (size_t)((N > 64 ? N : 64) + N * (N + 1)) * sizeof(float2);
    constexpr auto kern = hf2_kernel<N, THREADS, BF16>;
    if (lds > 48 * 1024) {
        const int rc = sid::allow_dynamic_lds<kern>(lds);
        if (rc != SID_OK) return rc;
    }
    const int pairs = (C + 1) / 2;
    hipLaunchKernelGGL(kern, dim3(B * pairs), dim3(THREADS), lds, stream, hcnt, skip, hres, out,
                       C, pairs, radius * radius);
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}

template <int H, int W, bool BF16>
int launch_hf(const void* hcnt, const void* skip, const void* hres, void* out, int B, int C, int radius,
              hipStream_t stream) {
    const size_t lds = (size_t)(64 + H * (W + 1)) * sizeof(float2);
    constexpr auto kern = hf_kernel<H, W, BF16>;
    if (lds > 48 * 1024) {
        const int rc = sid::allow_dynamic_lds<kern>(lds);
        if (rc != SID_OK) return rc;
    }
    const int pairs = (C + 1) / 2;
    hipLaunchKernelGGL(kern, dim3(B * pairs), dim3(256), lds, stream, hcnt, skip, hres, out, C, pairs,
                       radius * radius);
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}

template <bool BF16>
int dispatch_hf(const void* hcnt, const void* skip, const void* hres, void* out, int B, int C, int H, int W,
                int radius, hipStream_t stream) {
    // 16-byte (f32) / 8-byte (bf16) vector access needs aligned maps; H*W is a multiple of 4 here
    const size_t al = BF16 ? 7 : 15;
    const bool aligned = (((uintptr_t)hcnt | (uintptr_t)out | (uintptr_t)skip | (uintptr_t)hres) & al) == 0;
    const char* e = getenv("SID_TUNE_HF_GENERAL");       // measurement and tests only
    if (H == W && aligned && !(e && e[0] == '1')) {
        if (H == 128) return launch_hf2<128, BF16>(hcnt, skip, hres, out, B, C, radius, stream);
        if (H == 64) return launch_hf2<64, BF16>(hcnt, skip, hres, out, B, C, radius, stream);
        if (H == 32) return launch_hf2<32, BF16>(hcnt, skip, hres, out, B, C, radius, stream);
        if (H == 16) return launch_hf2<16, BF16>(hcnt, skip, hres, out, B, C, radius, stream);
    }
#define SID_HF(h, w) if (H == h && W == w) return launch_hf<h, w, BF16>(hcnt, skip, hres, out, B, C, radius, stream);
#define SID_HF_ROW(h) SID_HF(h, 8) SID_HF(h, 16) SID_HF(h, 32) SID_HF(h, 64) SID_HF(h, 128)
    SID_HF_ROW(8) SID_HF_ROW(16) SID_HF_ROW(32) SID_HF_ROW(64) SID_HF_ROW(128)
#undef SID_HF_ROW
#undef SID_HF
    return sid::fail(SID_ENOSYS, "feature map %dx%d: H and W must be powers of two in [8,128]", H, W);
}

int hf_common(const void* h_cnt, const void* skip, const void* h, void* out, int B, int C, int H, int W, int radius,
              int dtype, void* stream) {
    SID_CHECK_ARG(h_cnt && out, "NULL tensor pointer");
    SID_CHECK_ARG((skip == nullptr) == (h == nullptr), "skip and h must both be given or both be NULL");
    SID_CHECK_ARG(B >= 1 && C >= 1, "B and C must be >= 1");
    SID_CHECK_ARG(radius >= 0, "radius must be >= 0");
    SID_CHECK_ARG(dtype == SID_DTYPE_F32 || dtype == SID_DTYPE_BF16, "unknown dtype %d", dtype);
    SID_CHECK_ARG((long long)B * ((C + 1) / 2) < (1LL << 31), "too many feature maps");
    if (dtype == SID_DTYPE_BF16) return dispatch_hf<true>(h_cnt, skip, h, out, B, C, H, W, radius, (hipStream_t)stream);
    return dispatch_hf<false>(h_cnt, skip, h, out, B, C, H, W, radius, (hipStream_t)stream);
}

}  // namespace

extern "C" int sid_hf_filter_radius(const void* h_cnt, const void* skip, const void* h, void* out, int B, int C,
                                    int H, int W, int radius, int dtype, void* stream) {
    return hf_common(h_cnt, skip, h, out, B, C, H, W, radius, dtype, stream);
}

extern "C" int sid_hf_filter(const void* h_cnt, const void* skip, const void* h, void* out, int B, int C, int H,
                             int W, float ratio, int dtype, void* stream) {
    SID_CHECK_ARG(std::isfinite(ratio) && ratio >= 0.f, "ratio must be finite and >= 0 (got %g)", (double)ratio);
    SID_CHECK_ARG(H >= 1 && W >= 1, "H and W must be >= 1");
    // int(min(H, W) * radius_ratio), truncation toward zero
    const int radius = (int)((double)std::min(H, W) * (double)ratio);
    return hf_common(h_cnt, skip, h, out, B, C, H, W, radius, dtype, stream);
}

// ------------------------------------------------------------------------------------------
// out = (skip + h) + hf, the patched ResBlock sum with HF(h_cnt) computed earlier: the same additions
// in the same order as the epilogue of the fused filter, so both routes give the same bits.
// ------------------------------------------------------------------------------------------
namespace {

template <bool BF16>
__global__ void __launch_bounds__(256) residual_sum_kernel(const void* __restrict__ skip, const void* __restrict__ hres,
                                                           const void* __restrict__ hf, void* __restrict__ out,
                                                           long long n4, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float a[4], b[4], c[4];
        ld4<BF16>(skip, 4 * (size_t)i, a); ld4<BF16>(hres, 4 * (size_t)i, b); ld4<BF16>(hf, 4 * (size_t)i, c);
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (BF16) r[j] = (float)(__bf16)(a[j] + b[j]) + c[j];       // (skip + h) rounds to bf16, then + HF rounds again
            else r[j] = __fadd_rn(__fadd_rn(a[j], b[j]), c[j]);
        }
        if (BF16) {
            __bf16* op = (__bf16*)out + 4 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) op[j] = (__bf16)r[j];
        } else {
            *(float4*)((float*)out + 4 * i) = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
    // the last n % 4 elements, one thread each
    const long long t = 4 * n4 + i;
    if (i < n - 4 * n4) {
        if (BF16) {
            const __bf16 *sa = (const __bf16*)skip, *sb = (const __bf16*)hres, *sc = (const __bf16*)hf;
            ((__bf16*)out)[t] = (__bf16)((float)(__bf16)((float)sa[t] + (float)sb[t]) + (float)sc[t]);
        } else {
            const float *sa = (const float*)skip, *sb = (const float*)hres, *sc = (const float*)hf;
            ((float*)out)[t] = __fadd_rn(__fadd_rn(sa[t], sb[t]), sc[t]);
        }
    }
}

}  // namespace

extern "C" int sid_residual_sum(const void* skip, const void* h, const void* hf, void* out, long long n, int dtype,
                                void* stream) {
    SID_CHECK_ARG(skip && h && hf && out, "NULL tensor pointer");
    SID_CHECK_ARG(n >= 1, "n must be >= 1 (got %lld)", n);
    SID_CHECK_ARG(dtype == SID_DTYPE_F32 || dtype == SID_DTYPE_BF16, "unknown dtype %d", dtype);
    const uintptr_t al = (uintptr_t)skip | (uintptr_t)h | (uintptr_t)hf | (uintptr_t)out;
    SID_CHECK_ARG((al & (dtype == SID_DTYPE_F32 ? 15 : 7)) == 0, "tensors must be aligned to four elements");
    const long long n4 = n / 4;
    const unsigned blocks = (unsigned)((std::max(n4, n - 4 * n4) + 255) / 256);
    if (dtype == SID_DTYPE_BF16)
        hipLaunchKernelGGL(residual_sum_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, skip, h, hf, out, n4, n);
    else
        hipLaunchKernelGGL(residual_sum_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, skip, h, hf, out, n4, n);
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}
