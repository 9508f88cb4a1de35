// Single-launch attention for short key sequences (Nk <= 256) on gfx950.
//
// The 16x16 and 8x8 self-attentions of the UNet and its 77-token cross-attentions are latency
// problems, not throughput problems: the three-kernel pipeline (pack, flash, merge) spends most of
// its time in launches and dependent memory round trips.  Here one workgroup owns one tile of 32
// queries of one head and does everything in one pass, reading the caller's tensors directly:
//
//   phase 1   wave w takes key tile w: K and Q rows -> bf16 MFMA fragments (converted in registers),
//             S^T = K Q^T, per-query max of its 32 keys -> LDS
//   barrier   every wave now knows the row maximum over all keys
//             p = exp2(x - max), row sums and P (bf16, already in B-operand order) -> LDS
//   barrier
//   phase 2   wave b takes 32 output columns: O^T[b] = sum over key tiles of V^T[b] P^T, with the
//             V fragments gathered from global memory (issued before phase 1, so the loads overlap
//             it); divides by the row sum and stores.
//
// No workspace, no padding traffic, no partial results: HBM traffic is the tensors themselves.
#include "sid_attn_kernel.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace sid_attn {

constexpr int kSmallMaxTiles = 8;   // key tiles (waves) per workgroup: Nk <= 256

struct SmallParams {
    const void* q; const void* k; const void* v; void* out;
    long long qsb, qsh, qsn, ksb, ksh, ksn;     // element strides of (batch, head, token); head_dim contiguous
    float c2;                                   // logit scale in the log2 domain
    int H, Nq, Nk, d, NQT, nkt, ndv;            // ndv = ceil(d / 32) output column blocks
    int out_bf16;
    int two_tiles;                              // host only: two query tiles per workgroup where they fit
};

namespace {

template <bool BF16IN>
__device__ __forceinline__ float ld(const void* p, long long i) {
    if (BF16IN) return (float)((const __bf16*)p)[i];
    return ((const float*)p)[i];
}

__device__ __forceinline__ u32x4 bits(bf16x8 v) { return __builtin_bit_cast(u32x4, v); }

// Row loads come in three flavours, chosen on the host so that the kernel has no divergent paths:
//   MODE 0  element loads (any head_dim, any alignment)
//   MODE 1  f32, head_dim % 4 == 0, 16-byte aligned rows: two 16-byte loads per fragment
//   MODE 2  bf16, head_dim % 8 == 0, 16-byte aligned rows: the fragment IS one 16-byte load
// Every load is unconditional (clamped address, value discarded afterwards) and all loads of a
// phase are issued before the first one is consumed: a predicated or consumed-at-once load would
// cost one memory round trip each.
template <int MODE>
struct RowFrag {                 // raw registers of one 8-element fragment
    static constexpr int NV = MODE == 0 ? 8 : (MODE == 1 ? 8 : 4);
    float raw[MODE == 2 ? 1 : 8];
    u32x4 packed;
    template <bool BF16IN>
    __device__ __forceinline__ void load(const void* base, long long off, int d0, int d) {
        if (MODE == 1) {
            const float* p = (const float*)base + off;
            const f32x4 a = *(const f32x4*)(p + min(d0, d - 4));
            const f32x4 b = *(const f32x4*)(p + min(d0 + 4, d - 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) { raw[j] = a[j]; raw[4 + j] = b[j]; }
        } else if (MODE == 2) {
            packed = *(const u32x4*)((const __bf16*)base + off + min(d0, d - 8));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = ld<BF16IN>(base, off + min(d0 + j, d - 1));
        }
    }
    // QK16 (f32 input only): the fragment holds IEEE halves (saturated), typed bf16x8 like every fragment
    template <bool X3, bool QK16 = false>
    __device__ __forceinline__ void finish(bool row_ok, int d0, int d, bf16x8& hi, bf16x8& lo) const {
        if (MODE == 2) {
            const bool ok = row_ok && d0 + 8 <= d;
            u32x4 z = {0u, 0u, 0u, 0u};
            hi = as_bf16x8(ok ? packed : z);
            if (X3) lo = as_bf16x8(z);
            return;
        }
        f16x8 hv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = row_ok && (MODE == 1 ? d0 + 4 * (j >> 2) + 4 <= d : d0 + j < d);
            const float x = ok ? raw[j] : 0.f;
            if (QK16) {
                hv[j] = (_Float16)fminf(fmaxf(x, -65504.f), 65504.f);
            } else {
                hi[j] = (__bf16)x;
                if (X3) lo[j] = (__bf16)(x - (float)hi[j]);
            }
        }
        if (QK16) hi = __builtin_bit_cast(bf16x8, hv);
    }
};

// NKT: compile-time bound on the key tiles (2, 4 or 8), so that the V gather is one straight run of
// loads; tiles in [nkt, NKT) read clamped addresses and are not used.
// QK16: half-precision operands for the logits product (f32 input, not with X3).
template <int DKS, int NKT, bool X3, bool BF16IN, int MODE, bool QK16 = false>
__global__ void __launch_bounds__(kSmallMaxTiles * 64, 1) small_attn_kernel(SmallParams P) {
    static_assert(!QK16 || (!X3 && !BF16IN), "half-precision logits operands: f32 input, single product");
    __shared__ float sm_m[kSmallMaxTiles][32];
    __shared__ float sm_l[kSmallMaxTiles][32];
    __shared__ __attribute__((aligned(16))) u32x4 sm_p[(X3 ? 2 : 1) * kSmallMaxTiles * 2 * 64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, r = lane & 31;
    const int qt = blockIdx.x % P.NQT;
    const int bh = blockIdx.x / P.NQT;
    const int b = bh / P.H, h = bh % P.H;
    const int d = P.d;
    const long long qbase = b * P.qsb + h * P.qsh;
    const long long kbase = b * P.ksb + h * P.ksh;
    const bool qk_wave = wave < P.nkt;      // phase 1: this wave owns key tile `wave`
    const bool pv_wave = wave < P.ndv;      // phase 2: this wave owns output columns [32 wave, 32 wave + 32)

    // ---- phase 1: logits of key tile `wave`.  All K and Q loads first, then conversions and MFMAs.
    float x[16];
    if (qk_wave) {
        const int qrow = qt * 32 + r, krow = wave * 32 + r;
        const long long koff = kbase + (long long)min(krow, P.Nk - 1) * P.ksn;
        const long long qoff = qbase + (long long)min(qrow, P.Nq - 1) * P.qsn;
        RowFrag<MODE> kr[DKS], qr[DKS];
#pragma unroll
        for (int s = 0; s < DKS; ++s) {
            kr[s].template load<BF16IN>(P.k, koff, 16 * s + 8 * hh, d);
            qr[s].template load<BF16IN>(P.q, qoff, 16 * s + 8 * hh, d);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 S;
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
        for (int s = 0; s < DKS; ++s) {
            bf16x8 kh, kl, qh, ql;
            kr[s].template finish<X3, QK16>(krow < P.Nk, 16 * s + 8 * hh, d, kh, kl);
            qr[s].template finish<X3, QK16>(qrow < P.Nq, 16 * s + 8 * hh, d, qh, ql);
            S = mfma_qk<QK16>(kh, qh, S);
            if (X3) {
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, S, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            x[i] = key < P.Nk ? S[i] * P.c2 : -SID_INF;
        }
    }

    // ---- V^T fragments of this wave's output column block, all key tiles: issue every load now, so
    // that they are in flight during the softmax exchange below.
    // A operand of O^T += V^T P^T: lane (r, hh) holds V[key = 32t + 16 s2 + 8 (j>>2) + 4 hh + (j&3)][32 wave + r]
    float vraw[NKT][2][8];
    const int dv = 32 * wave + r;
    if (pv_wave) {
        const long long voff = kbase + min(dv, d - 1);
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = 32 * t + 16 * s2 + 8 * (j >> 2) + 4 * hh + (j & 3);
                    vraw[t][s2][j] = ld<BF16IN>(P.v, voff + (long long)min(key, P.Nk - 1) * P.ksn);
                }
    }
    __builtin_amdgcn_sched_barrier(0);

    if (qk_wave) {
        float m = -SID_INF;
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmaxf(m, x[i]);
        m = fmaxf(m, __shfl_xor(m, 32));
        if (hh == 0) sm_m[wave][r] = m;
    }
    __syncthreads();
    if (qk_wave) {
        float M = -SID_INF;
        for (int t = 0; t < P.nkt; ++t) M = fmaxf(M, sm_m[t][r]);
        float pv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = __builtin_amdgcn_exp2f(x[i] - M);      // exp2(-inf) = 0 for masked keys
        bf16x8 p0, p1, p0l, p1l;
#pragma unroll
        for (int j = 0; j < 8; ++j) { p0[j] = (__bf16)pv[j]; p1[j] = (__bf16)pv[8 + j]; }
        float l = 0.f;
        if (X3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p0l[j] = (__bf16)(pv[j] - (float)p0[j]);
                p1l[j] = (__bf16)(pv[8 + j] - (float)p1[j]);
                l += ((float)p0[j] + (float)p0l[j]) + ((float)p1[j] + (float)p1l[j]);
            }
        } else {
            // the denominator sums the rounded numerators, so rounding cancels in the quotient
#pragma unroll
            for (int j = 0; j < 8; ++j) l += (float)p0[j] + (float)p1[j];
        }
        l += __shfl_xor(l, 32);
        if (hh == 0) sm_l[wave][r] = l;
        sm_p[(wave * 2 + 0) * 64 + lane] = bits(p0);
        sm_p[(wave * 2 + 1) * 64 + lane] = bits(p1);
        if (X3) {
            sm_p[(kSmallMaxTiles * 2 + wave * 2 + 0) * 64 + lane] = bits(p0l);
            sm_p[(kSmallMaxTiles * 2 + wave * 2 + 1) * 64 + lane] = bits(p1l);
        }
    }
    __syncthreads();

    // ---- phase 2: output columns [32 wave, 32 wave + 32)
    if (!pv_wave) return;
    f32x16 O;
#pragma unroll
    for (int i = 0; i < 16; ++i) O[i] = 0.f;
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        if (t < P.nkt) {
            l += sm_l[t][r];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 vh, vl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = 32 * t + 16 * s2 + 8 * (j >> 2) + 4 * hh + (j & 3);
                    const float val = (key < P.Nk && dv < d) ? vraw[t][s2][j] : 0.f;
                    vh[j] = (__bf16)val;
                    if (X3) vl[j] = (__bf16)(val - (float)vh[j]);
                }
                const bf16x8 p = as_bf16x8(sm_p[(t * 2 + s2) * 64 + lane]);
                O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, p, O, 0, 0, 0);
                if (X3) {
                    const bf16x8 pl = as_bf16x8(sm_p[(kSmallMaxTiles * 2 + t * 2 + s2) * 64 + lane]);
                    O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, O, 0, 0, 0);
                    O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, p, O, 0, 0, 0);
                }
            }
        }
    }
    const int q = qt * 32 + r;
    if (q >= P.Nq) return;
    const float inv = 1.0f / l;
    const size_t obase = ((size_t)b * P.Nq + q) * P.H * d + (size_t)h * d;
    const bool ovec = !P.out_bf16 && (d & 3) == 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int dv0 = 32 * wave + 8 * g + 4 * hh;
        if (ovec) {
            if (dv0 < d) {
                f32x4 o = {O[4 * g] * inv, O[4 * g + 1] * inv, O[4 * g + 2] * inv, O[4 * g + 3] * inv};
                *(f32x4*)((float*)P.out + obase + dv0) = o;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (dv0 + e < d) {
                    const float val = O[4 * g + e] * inv;
                    if (P.out_bf16) ((__bf16*)P.out)[obase + dv0 + e] = (__bf16)val;
                    else ((float*)P.out)[obase + dv0 + e] = val;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// LDS-staged variant (bf16 products, 16-byte aligned rows): every global load is a fully coalesced
// 16-byte-per-lane read of contiguous row pieces; the MFMA operand order is produced by where the
// converted pieces are WRITTEN in LDS.
//   K tile w / Q tile : fragment order, unit (k-step s, half hh) at (2s + hh) * kUnit + row * 16
//                       (kUnit = 528: the 16 spare bytes rotate the units over the banks)
//   V                 : bf16 row-major [key][vst], written after the logits are done into the space
//                       K and Q occupied; phase 2 gathers its A fragments with 2-byte reads
// MODE 1: f32 input (4 elements per piece), MODE 2: bf16 input (8 elements per piece).
// QT: query tiles per workgroup.  A workgroup reads its head's whole K and V whatever it does with them,
// and that read is what its time goes to (a CU pulls in about 11 bytes per cycle), so two query tiles per
// workgroup halve the work of a call: every K fragment read from LDS and every V fragment gathered in
// phase 2 feeds two MFMAs.
// ------------------------------------------------------------------------------------------
constexpr int kUnit = 528;

// QK16 (MODE 1 only): the K and Q images hold IEEE halves, the logits MFMA is the f16 form; V and P stay bf16.
template <int DKS, int MODE, int QT, bool QK16 = false>
__global__ void __launch_bounds__(kSmallMaxTiles * 64, 1) small_attn_lds_kernel(SmallParams P, int vst, int p_off, int st_off) {
    static_assert(!QK16 || MODE == 1, "half-precision logits operands need f32 input");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int EPP = MODE == 1 ? 4 : 8;           // elements per 16-byte piece
    constexpr int PRF = 16 * DKS / EPP;              // pieces per padded row
    constexpr int NPT = 32 * PRF;                    // pieces per 32-row tile
    constexpr int ITERS = NPT / 64;                  // per lane, when one wave copies one tile
    constexpr int TILE_BYTES = 2 * DKS * kUnit;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x;
    const int hh = lane >> 5, r = lane & 31;
    const int nqg = (P.NQT + QT - 1) / QT;           // query groups per head
    const int qt = (blockIdx.x % nqg) * QT;          // first query tile of this workgroup
    const int bh = blockIdx.x / nqg;
    const int b = bh / P.H, h = bh % P.H;
    const int d = P.d;
    const int pr = d / EPP;                          // valid pieces per row
    const long long qbase = b * P.qsb + h * P.qsh;
    const long long kbase = b * P.ksb + h * P.ksh;
    const bool qk_wave = wave < P.nkt, pv_wave = wave < P.ndv;
    unsigned char* kq = smem;                        // K tiles [nkt], then the QT query tiles
    unsigned char* qb = smem + P.nkt * TILE_BYTES;
    float* sm_m = (float*)(smem + st_off);           // [QT][key tile][32]
    float* sm_l = sm_m + QT * kSmallMaxTiles * 32;
    u32x4* sm_p = (u32x4*)(smem + p_off);            // [QT][key tile][2][64]

    // one 16-byte piece of a row -> registers (4 or 8 bf16), zero for padding columns
    auto load_piece = [&](const void* base, long long rowoff, int col, u32x4& raw) {
        const int c = min(col, pr - 1);
        if (MODE == 1) raw = __builtin_bit_cast(u32x4, *(const f32x4*)((const float*)base + rowoff + 4 * c));
        else raw = *(const u32x4*)((const __bf16*)base + rowoff + 8 * c);
    };
    // `half`: a row of the logits product in QK16 mode (K, Q): IEEE halves, saturated; V is always bf16
    auto to_bf16 = [&](const u32x4& raw, bool ok, unsigned (&o)[4], bool half = false) {       // MODE 1 fills o[0..1], MODE 2 o[0..3]
        if (MODE == 1) {
            const f32x4 f = __builtin_bit_cast(f32x4, raw);
            if (QK16 && half) {
                typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
                auto sat = [](float x) { return (_Float16)fminf(fmaxf(x, -65504.f), 65504.f); };
                f16x2 lo2 = {sat(f[0]), sat(f[1])}, hi2 = {sat(f[2]), sat(f[3])};
                o[0] = ok ? __builtin_bit_cast(unsigned, lo2) : 0u;
                o[1] = ok ? __builtin_bit_cast(unsigned, hi2) : 0u;
                return;
            }
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            bf16x2 lo2 = {(__bf16)f[0], (__bf16)f[1]}, hi2 = {(__bf16)f[2], (__bf16)f[3]};
            o[0] = ok ? __builtin_bit_cast(unsigned, lo2) : 0u;
            o[1] = ok ? __builtin_bit_cast(unsigned, hi2) : 0u;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = ok ? raw[i] : 0u;
        }
    };
    // fragment-order address of piece (row, col) inside a tile image
    auto frag_addr = [&](int row, int col) {
        const int dd = col * EPP;
        return ((dd >> 4) * 2 + ((dd >> 3) & 1)) * kUnit + row * 16 + (dd & 7) * 2;
    };

    // ---- issue every global load: this wave's K tile and V tile, and a share of the Q tile
    u32x4 kraw[ITERS], vraw[ITERS];
    if (qk_wave) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int f = i * 64 + lane, row = f / PRF, col = f % PRF;
            const long long off = kbase + (long long)min(wave * 32 + row, P.Nk - 1) * P.ksn;
            load_piece(P.k, off, col, kraw[i]);
            load_piece(P.v, off, col, vraw[i]);
        }
    }
    // the Q tile is shared: all threads copy it, four pieces per thread and round (one round for a
    // workgroup of 320+ threads at head_dim 160)
    auto store_piece = [&](unsigned char* dst, const unsigned (&o)[4]) {
        if (MODE == 1) { u32x2 w2 = {o[0], o[1]}; *(u32x2*)dst = w2; }
        else { u32x4 w4 = {o[0], o[1], o[2], o[3]}; *(u32x4*)dst = w4; }
    };
    for (int f0 = 0; f0 < QT * NPT; f0 += 4 * nthreads) {
        u32x4 qraw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = min(f0 + i * nthreads + tid, QT * NPT - 1), row = f / PRF, col = f % PRF;     // row of the QT * 32
            load_piece(P.q, qbase + (long long)min(qt * 32 + row, P.Nq - 1) * P.qsn, col, qraw[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = f0 + i * nthreads + tid;
            if (f < QT * NPT) {
                const int row = f / PRF, col = f % PRF;
                unsigned o[4];
                to_bf16(qraw[i], col < pr, o, true);
                store_piece(qb + (row >> 5) * TILE_BYTES + frag_addr(row & 31, col), o);
            }
        }
    }

    // ---- K into LDS in fragment order
    if (qk_wave) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int f = i * 64 + lane, row = f / PRF, col = f % PRF;
            unsigned o[4];
            to_bf16(kraw[i], col < pr, o, true);
            store_piece(kq + wave * TILE_BYTES + frag_addr(row, col), o);
        }
    }
    // V converted now (frees the raw registers), written after the logits
    unsigned vb16[ITERS][MODE == 1 ? 2 : 4];
    if (qk_wave) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            unsigned o[4];
            to_bf16(vraw[i], true, o);
#pragma unroll
            for (int e = 0; e < (MODE == 1 ? 2 : 4); ++e) vb16[i][e] = o[e];
        }
    }
    __syncthreads();

    // ---- phase 1: logits of key tile `wave`
    float x[QT][16];
    if (qk_wave) {
        f32x16 S[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
#pragma unroll
        for (int s = 0; s < DKS; ++s) {
            const bf16x8 kf = as_bf16x8(*(const u32x4*)(kq + wave * TILE_BYTES + (2 * s + hh) * kUnit + r * 16));
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const bf16x8 qf = as_bf16x8(*(const u32x4*)(qb + t * TILE_BYTES + (2 * s + hh) * kUnit + r * 16));
                S[t] = mfma_qk<QK16>(kf, qf, S[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float m = -SID_INF;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                x[t][i] = key < P.Nk ? S[t][i] * P.c2 : -SID_INF;
                m = fmaxf(m, x[t][i]);
            }
            m = fmaxf(m, __shfl_xor(m, 32));
            if (hh == 0) sm_m[(t * kSmallMaxTiles + wave) * 32 + r] = m;
        }
    }
    __syncthreads();          // maxima visible; nobody reads the K/Q images any more

    if (qk_wave) {
        // V tile -> LDS row-major bf16, over the K/Q images
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int f = i * 64 + lane, row = f / PRF, col = f % PRF;
            if (col < pr) {
                unsigned char* dst = smem + (wave * 32 + row) * vst + col * EPP * 2;
                if (MODE == 1) { u32x2 w2 = {vb16[i][0], vb16[i][1]}; *(u32x2*)dst = w2; }
                else { u32x4 w4 = {vb16[i][0], vb16[i][1], vb16[i][MODE == 1 ? 0 : 2], vb16[i][MODE == 1 ? 1 : 3]}; *(u32x4*)dst = w4; }
            }
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float M = -SID_INF;
            for (int kt = 0; kt < P.nkt; ++kt) M = fmaxf(M, sm_m[(t * kSmallMaxTiles + kt) * 32 + r]);
            float pv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) pv[i] = __builtin_amdgcn_exp2f(x[t][i] - M);      // exp2(-inf) = 0 for masked keys
            bf16x8 p0, p1;
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p0[j] = (__bf16)pv[j]; p1[j] = (__bf16)pv[8 + j];
                l += (float)p0[j] + (float)p1[j];      // the denominator sums the rounded numerators
            }
            l += __shfl_xor(l, 32);
            if (hh == 0) sm_l[(t * kSmallMaxTiles + wave) * 32 + r] = l;
            sm_p[((t * kSmallMaxTiles + wave) * 2 + 0) * 64 + lane] = bits(p0);
            sm_p[((t * kSmallMaxTiles + wave) * 2 + 1) * 64 + lane] = bits(p1);
        }
    }
    __syncthreads();

    // ---- phase 2: output columns [32 wave, 32 wave + 32): A fragments gathered from the V rows
    if (!pv_wave) return;
    const int dv = 32 * wave + r;
    const bool dv_ok = dv < d;
    const unsigned char* vcol = smem + min(dv, d - 1) * 2;
    f32x16 O[QT];
    float l[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        l[t] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) O[t][i] = 0.f;
    }
    for (int kt = 0; kt < P.nkt; ++kt) {
#pragma unroll
        for (int t = 0; t < QT; ++t) l[t] += sm_l[(t * kSmallMaxTiles + kt) * 32 + r];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            unsigned w[4];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                const int j = 2 * jp;
                const int key = 32 * kt + 16 * s2 + 8 * (j >> 2) + 4 * hh + (j & 3);     // j and j+1 are adjacent keys
                const unsigned lo = *(const unsigned short*)(vcol + key * vst);
                const unsigned hi = *(const unsigned short*)(vcol + (key + 1) * vst);
                w[jp] = dv_ok ? (lo | (hi << 16)) : 0u;
            }
            u32x4 vfrag = {w[0], w[1], w[2], w[3]};
#pragma unroll
            for (int t = 0; t < QT; ++t)
                O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vfrag),
                                                               as_bf16x8(sm_p[((t * kSmallMaxTiles + kt) * 2 + s2) * 64 + lane]), O[t], 0, 0, 0);
        }
    }
    const bool ovec = !P.out_bf16 && (d & 3) == 0;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = (qt + t) * 32 + r;
        if (q >= P.Nq) continue;
        const float inv = 1.0f / l[t];
        const size_t obase = ((size_t)b * P.Nq + q) * P.H * d + (size_t)h * d;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dv0 = 32 * wave + 8 * g + 4 * hh;
            if (ovec) {
                if (dv0 < d) {
                    f32x4 o = {O[t][4 * g] * inv, O[t][4 * g + 1] * inv, O[t][4 * g + 2] * inv, O[t][4 * g + 3] * inv};
                    *(f32x4*)((float*)P.out + obase + dv0) = o;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (dv0 + e < d) {
                        const float val = O[t][4 * g + e] * inv;
                        if (P.out_bf16) ((__bf16*)P.out)[obase + dv0 + e] = (__bf16)val;
                        else ((float*)P.out)[obase + dv0 + e] = val;
                    }
                }
            }
        }
    }
}

// LDS of the staged kernel with QT query tiles per workgroup: [K tiles + QT query tiles | V rows] (shared),
// the probabilities and the row statistics of every query tile
struct SmallLds { int vst; size_t p_off, st_off, bytes; };
template <int DKS>
SmallLds small_lds_layout(const SmallParams& p, int QT) {
    SmallLds L;
    // V row stride in bytes: 16-byte multiple whose bank offset between the two lane halves
    // (rows 4 apart) clears the 16 banks one half covers
    L.vst = (2 * p.d + 15) / 16 * 16;
    while (L.vst % 64 < 16 || L.vst % 64 > 48) L.vst += 16;
    const size_t images = std::max((size_t)(p.nkt + QT) * 2 * DKS * kUnit, (size_t)p.nkt * 32 * L.vst);
    L.p_off = (images + 255) / 256 * 256;
    L.st_off = L.p_off + (size_t)QT * kSmallMaxTiles * 2 * 1024;
    L.bytes = L.st_off + (size_t)QT * 2 * kSmallMaxTiles * 32 * sizeof(float);
    return L;
}

template <int DKS, int MODE, int QT, bool QK16>
int launch_small_lds_qt(const SmallParams& p, int BH, int waves, const SmallLds& L, hipStream_t stream) {
    constexpr auto kern = small_attn_lds_kernel<DKS, MODE, QT, QK16>;
    const int rc = sid::allow_dynamic_lds<kern>(160 * 1024);
    if (rc != SID_OK) return rc;
    const int nqg = (p.NQT + QT - 1) / QT;
    hipLaunchKernelGGL(kern, dim3((unsigned)(BH * nqg)), dim3(waves * 64), L.bytes, stream, p, L.vst, (int)L.p_off, (int)L.st_off);
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}

template <int DKS, int MODE, bool QK16 = false>
int launch_small_lds(const SmallParams& p, int BH, int waves, hipStream_t stream) {
    // Two query tiles per workgroup halve the work of a call and add a quarter to the time of one workgroup
    // (measured, one -> two tiles: (256,256,160) 11.1 -> 13.9 us at one image, 21.4 -> 18.6 us at four;
    // (4096,77,40) 9.3 -> 8.7 us at one image, 27.7 -> 23.4 us at four; (1024,77,80) 6.2 -> 7.5 us): taken
    // when the GPU is full anyway, i.e. the call shares it with other streams, launches two workgroups per
    // CU by itself, or one per CU with a K/V long enough to be the workgroup's main cost.
    // SID_TUNE_SMALL_QT = 1 / 2 forces the choice (measurement and tests only).
    const char* e = getenv("SID_TUNE_SMALL_QT");
    const SmallLds two = small_lds_layout<DKS>(p, 2);
    const long long wgs = (long long)BH * p.NQT;
    const bool full = p.two_tiles || wgs >= 512 || (wgs >= 256 && p.Nk >= 128);
    const bool want = (e && e[0] == '2') || (!(e && e[0] == '1') && full);
    if (want && p.NQT >= 2 && two.bytes <= 160 * 1024)
        return launch_small_lds_qt<DKS, MODE, 2, QK16>(p, BH, waves, two, stream);
    return launch_small_lds_qt<DKS, MODE, 1, QK16>(p, BH, waves, small_lds_layout<DKS>(p, 1), stream);
}

// qk16 with bf16 input runs the bf16 kernels: a bf16 value is a half value wherever half's range holds it,
// so the logits product gains nothing from the conversion.
template <int DKS, int NKT, bool X3>
int launch_small_x(const SmallParams& p, int BH, int waves, bool bf16in, bool qk16, hipStream_t stream) {
    const dim3 grid((unsigned)(BH * p.NQT)), block(waves * 64);
    auto al = [](const void* ptr, long long a, long long b, long long c, int m) {
        return ((uintptr_t)ptr & 15) == 0 && a % m == 0 && b % m == 0 && c % m == 0;
    };
    const int m = bf16in ? 8 : 4;
    const bool vec = p.d % m == 0 && al(p.q, p.qsb, p.qsh, p.qsn, m) && al(p.k, p.ksb, p.ksh, p.ksn, m) &&
                     ((uintptr_t)p.v & 15) == 0;
    const char* e = getenv("SID_TUNE_SMALL_DIRECT");       // measurement and tests only
    if (!X3 && vec && !(e && e[0] == '1')) {
        if (bf16in) return launch_small_lds<DKS, 2>(p, BH, waves, stream);
        return qk16 ? launch_small_lds<DKS, 1, true>(p, BH, waves, stream) : launch_small_lds<DKS, 1>(p, BH, waves, stream);
    }
    if constexpr (!X3) {
        if (qk16 && !bf16in) {
            if (vec) hipLaunchKernelGGL((small_attn_kernel<DKS, NKT, false, false, 1, true>), grid, block, 0, stream, p);
            else hipLaunchKernelGGL((small_attn_kernel<DKS, NKT, false, false, 0, true>), grid, block, 0, stream, p);
            SID_CHECK_HIP(hipGetLastError());
            return SID_OK;
        }
    }
    if (bf16in) {
        if (vec) hipLaunchKernelGGL((small_attn_kernel<DKS, NKT, X3, true, 2>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((small_attn_kernel<DKS, NKT, X3, true, 0>), grid, block, 0, stream, p);
    } else {
        if (vec) hipLaunchKernelGGL((small_attn_kernel<DKS, NKT, X3, false, 1>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((small_attn_kernel<DKS, NKT, X3, false, 0>), grid, block, 0, stream, p);
    }
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}

template <int DKS, int NKT>
int launch_small_inst(const SmallParams& p, int BH, int waves, bool x3, bool bf16in, bool qk16, hipStream_t stream) {
    return x3 ? launch_small_x<DKS, NKT, true>(p, BH, waves, bf16in, false, stream)
              : launch_small_x<DKS, NKT, false>(p, BH, waves, bf16in, qk16, stream);
}

template <int DKS>
int launch_small_dks(const SmallParams& p, int BH, int waves, bool x3, bool bf16in, bool qk16, hipStream_t stream) {
    if (p.nkt <= 2) return launch_small_inst<DKS, 2>(p, BH, waves, x3, bf16in, qk16, stream);
    if (p.nkt <= 4) return launch_small_inst<DKS, 4>(p, BH, waves, x3, bf16in, qk16, stream);
    return launch_small_inst<DKS, 8>(p, BH, waves, x3, bf16in, qk16, stream);
}

}  // namespace

bool small_attn_supported(int Nk, int d) { return Nk <= 32 * kSmallMaxTiles && (d + 31) / 32 <= kSmallMaxTiles && d <= 160; }

int launch_small_attn(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int d,
                      float c2, const long long qs[3], const long long ks[3], bool bf16io, bool x3, hipStream_t stream, bool shared_gpu,
                      bool qk16) {
    SmallParams p = {};
    p.two_tiles = shared_gpu;
    p.q = q; p.k = k; p.v = v; p.out = out;
    p.qsb = qs[0]; p.qsh = qs[1]; p.qsn = qs[2]; p.ksb = ks[0]; p.ksh = ks[1]; p.ksn = ks[2];
    p.c2 = c2; p.H = H; p.Nq = Nq; p.Nk = Nk; p.d = d;
    p.NQT = (Nq + 31) / 32; p.nkt = (Nk + 31) / 32; p.ndv = (d + 31) / 32;
    p.out_bf16 = bf16io;
    const int waves = p.nkt > p.ndv ? p.nkt : p.ndv;
    const int dks = (d + 15) / 16;
    switch (dks) {
        case 1: case 2: return launch_small_dks<2>(p, B * H, waves, x3, bf16io, qk16, stream);
        case 3: return launch_small_dks<3>(p, B * H, waves, x3, bf16io, qk16, stream);
        case 4: return launch_small_dks<4>(p, B * H, waves, x3, bf16io, qk16, stream);
        case 5: return launch_small_dks<5>(p, B * H, waves, x3, bf16io, qk16, stream);
        case 6: return launch_small_dks<6>(p, B * H, waves, x3, bf16io, qk16, stream);
        case 7: case 8: return launch_small_dks<8>(p, B * H, waves, x3, bf16io, qk16, stream);
        case 9: case 10: return launch_small_dks<10>(p, B * H, waves, x3, bf16io, qk16, stream);
    }
    return sid::fail(SID_ENOSYS, "head_dim %d has no short-sequence kernel", d);
}

}  // namespace sid_attn
