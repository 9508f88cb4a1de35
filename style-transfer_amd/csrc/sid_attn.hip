// Injected three-block attention and single-block attention for gfx950 (MI355X).
//
// Three kernels per call, all on the caller's stream:
//   1. pack   : f32/bf16 inputs -> bf16 MFMA operand fragments in the workspace.  The query blend
//               q = q_cnt*gamma + x_q*coef happens here; K is zero-padded to a multiple of 16 in
//               head_dim; V gets a column of ones (so P*V also yields the softmax row sums) and is
//               stored transposed in fragment order.  The fragment image of a 32-key tile is
//               exactly what a wave's lanes read, so the main kernel copies it to LDS verbatim
//               and reads it back with conflict-free lane-linear ds_read_b128.
//   2. flash  : per (key block, key split, head, 32-query tile) one wave runs an online-softmax
//               pass over its keys: S^T = K Q^T (keys on MFMA rows, queries on lanes), P = exp2,
//               O^T += V^T P^T.  The S^T accumulator is already the B operand of the second
//               product, so P never leaves registers.  Also tracks the raw row max/min of q.k,
//               which the mask bias needs.  Writes un-normalised partials (O, l, m) + stats.
//   3. merge  : per query row, turns the stats into the two bias terms
//               b = (min_j cc - max_j q.k_b) * (mask*ch_b), and combines the partials into the
//               softmax over the concatenated [char | back | content] keys.
//
// MFMA: v_mfma_f32_32x32x16_bf16.  Lane l = 32*hh + r holds A[row r][k = 8hh+j], B[k = 8hh+j][col r]
// (j = 0..7) and D[row = (i&3) + 8(i>>2) + 4hh][col r] in register i = 0..15.

#include "sid_common.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // 16-byte staging unit (native vector: stays in registers)

constexpr int kMaxJobs = 8;
constexpr int kBlocks = 3;          // key blocks of the injected attention: char, back, content
constexpr float kRescaleThr = 6.0f; // log2 units: P stays <= 2^6 between lazy rescales
#define SID_INF __builtin_huge_valf()

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ u32x4 as_u32x4(bf16x8 v) { return __builtin_bit_cast(u32x4, v); }

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// key index inside a 32-key tile that element j of lane-half hh feeds in k-step s2 of P*V
__host__ __device__ __forceinline__ int pv_key(int s2, int hh, int j) {
    return 16 * s2 + 8 * (j >> 2) + 4 * hh + (j & 3);
}

// ------------------------------------------------------------------------------------------
// pack
// ------------------------------------------------------------------------------------------
struct PackJob {
    const void* src;    // (b,h,n,dd) with strides sb,sh,sn
    const void* src2;   // optional second source for the blend, strides sb2,sh2,sn2
    u32x4* dst;         // hi (or only) part
    u32x4* dst_lo;      // lo part in split mode, else null
    long long sb, sh, sn, sb2, sh2, sn2;
    float c1, c2;       // value = src*c1 + src2*c2 when src2 != null, else src
    int kind;           // 0: row fragments (Q, K)   1: transposed V fragments with ones column
    int N;              // valid rows
    int NT;             // 32-row tiles in the image (>= ceil(N/32)); rows >= N are zero
    int first_block;    // first blockIdx.x of this job
    long long frags;    // 16-byte fragments to write
};

struct PackParams {
    PackJob job[kMaxJobs];
    int njobs;
    int H, d, DKS, DVB;
    int in_bf16;
};

template <bool BF16IN>
__device__ __forceinline__ float load_elem(const void* p, long long idx) {
    if (BF16IN) return (float)((const __bf16*)p)[idx];
    return ((const float*)p)[idx];
}

template <bool BF16IN>
__global__ void __launch_bounds__(256) pack_kernel(PackParams P) {
    int jb = 0;
#pragma unroll
    for (int i = 1; i < kMaxJobs; ++i)
        if (i < P.njobs && (int)blockIdx.x >= P.job[i].first_block) jb = i;
    const PackJob& J = P.job[jb];
    const long long gid = (long long)(blockIdx.x - J.first_block) * 256 + threadIdx.x;
    if (gid >= J.frags) return;
    const int lane = (int)(gid & 63);
    const int hh = lane >> 5, r = lane & 31;
    float val[8];
    if (J.kind == 0) {
        const long long t = gid >> 6;
        const int s = (int)(t % P.DKS);
        const long long t2 = t / P.DKS;
        const int tile = (int)(t2 % J.NT);
        const long long bh = t2 / J.NT;
        const int b = (int)(bh / P.H), h = (int)(bh % P.H);
        const int row = tile * 32 + r;
        const int d0 = 16 * s + 8 * hh;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int dd = d0 + j;
            float v = 0.f;
            if (row < J.N && dd < P.d) {
                v = load_elem<BF16IN>(J.src, b * J.sb + h * J.sh + (long long)row * J.sn + dd);
                if (J.src2) {
                    const float w = load_elem<BF16IN>(J.src2, b * J.sb2 + h * J.sh2 + (long long)row * J.sn2 + dd);
                    v = __fadd_rn(__fmul_rn(v, J.c1), __fmul_rn(w, J.c2));
                }
            }
            val[j] = v;
        }
    } else {
        const long long t = gid >> 6;
        const int dvb = (int)(t % P.DVB);
        const long long t1 = t / P.DVB;
        const int s2 = (int)(t1 & 1);
        const long long t2 = t1 >> 1;
        const int tile = (int)(t2 % J.NT);
        const long long bh = t2 / J.NT;
        const int b = (int)(bh / P.H), h = (int)(bh % P.H);
        const int dv = 32 * dvb + r;
        const int ones_col = 32 * P.DVB - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = tile * 32 + pv_key(s2, hh, j);
            float v = 0.f;
            if (row < J.N) {
                if (dv < P.d) v = load_elem<BF16IN>(J.src, b * J.sb + h * J.sh + (long long)row * J.sn + dv);
                else if (dv == ones_col) v = 1.f;
            }
            val[j] = v;
        }
    }
    bf16x8 hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) hi[j] = (__bf16)val[j];
    J.dst[gid] = as_u32x4(hi);
    if (J.dst_lo) {
        bf16x8 lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) lo[j] = (__bf16)(val[j] - (float)hi[j]);
        J.dst_lo[gid] = as_u32x4(lo);
    }
}

// ------------------------------------------------------------------------------------------
// flash
// ------------------------------------------------------------------------------------------
struct FlashParams {
    const u32x4* q[kBlocks];     // packed query used by each key block (hi part)
    const u32x4* q_lo[kBlocks];
    const u32x4* k[kBlocks];
    const u32x4* k_lo[kBlocks];
    const u32x4* v[kBlocks];
    const u32x4* v_lo[kBlocks];
    float c2[kBlocks];           // logit scale of each block, log2 domain: x = (q.k) * c2
    float* opart;                // [part][BH][Nq][d]
    float* stats;                // [part][BH][Nq][4] = (m2, l, smax, smin)
    int BH, Nq, Nk, d;
    int NQT, NKT;                // 32-row tiles in the packed query / key images
    int nQG;                     // workgroups per (bh, part): ceil(NQT / waves)
    int KS;                      // key splits per block
    int SPS;                     // stages per split
    int nblk;
};

template <int DKS, int DVB, int NW, int KVS, bool X3>
__global__ void __launch_bounds__(NW * 64) flash_kernel(FlashParams P) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    constexpr int KU = KVS * DKS * 64;        // u32x4 per stage: K image
    constexpr int VU = KVS * 2 * DVB * 64;    //                  V image
    constexpr int PU = KU + VU;               // one precision part
    constexpr int SU = PU * (X3 ? 2 : 1);     // one stage
    constexpr int NT = NW * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5;

    // workgroup -> (query group, head, part).  Workgroups that stream the same K/V image get
    // ids that are equal mod 8, i.e. they share an XCD and its L2 (speed only).
    const int groups = P.BH * P.nblk * P.KS;
    int grp, qg;
    {
        const int id = blockIdx.x;
        if ((groups & 7) == 0) {
            const int slot = id >> 3;
            grp = (id & 7) + 8 * (slot / P.nQG);
            qg = slot % P.nQG;
        } else {
            grp = id / P.nQG;
            qg = id % P.nQG;
        }
    }
    const int part = grp % (P.nblk * P.KS);
    const int bh = grp / (P.nblk * P.KS);
    const int blk = part / P.KS, ks = part % P.KS;

    const int qt_raw = qg * NW + wave;
    const bool wave_active = qt_raw < P.NQT;
    const int qt = wave_active ? qt_raw : P.NQT - 1;

    bf16x8 qf[DKS];
    bf16x8 ql[X3 ? DKS : 1];
    {
        const u32x4* qs = P.q[blk] + ((size_t)(bh * P.NQT + qt) * DKS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < DKS; ++s) qf[s] = as_bf16x8(qs[s * 64]);
        if (X3) {
            const u32x4* ql_ = P.q_lo[blk] + ((size_t)(bh * P.NQT + qt) * DKS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < DKS; ++s) ql[s] = as_bf16x8(ql_[s * 64]);
        }
    }

    f32x16 O[DVB];
#pragma unroll
    for (int b = 0; b < DVB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[b][i] = 0.f;
    float m2 = -SID_INF, smax = -SID_INF, smin = SID_INF;
    const float c2 = P.c2[blk];
    const bool need_min = (blk == P.nblk - 1 && P.nblk == kBlocks) || c2 < 0.f;

    const int nStages = P.NKT / KVS;
    const int st0 = ks * P.SPS;
    const int st1 = min(st0 + P.SPS, nStages);

    const u32x4* gk = P.k[blk] + (size_t)bh * P.NKT * DKS * 64;
    const u32x4* gv = P.v[blk] + (size_t)bh * P.NKT * 2 * DVB * 64;
    const u32x4* gkl = X3 ? P.k_lo[blk] + (size_t)bh * P.NKT * DKS * 64 : nullptr;
    const u32x4* gvl = X3 ? P.v_lo[blk] + (size_t)bh * P.NKT * 2 * DVB * 64 : nullptr;

    // Staging: the stage image in LDS is [K hi | V hi | K lo | V lo]; every part is a verbatim copy
    // of a contiguous run of the packed global image.
    constexpr int NPK = (KU + NT - 1) / NT, NPV = (VU + NT - 1) / NT;
    u32x4 preK[NPK], preV[NPV], preKl[X3 ? NPK : 1], preVl[X3 ? NPV : 1];
    auto stage_load = [&](int st) {
        static_for<0, NPK>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = min(tid + i * NT, KU - 1);   // surplus lanes reload the last unit
            preK[i] = gk[(size_t)st * KU + u];
            if (X3) preKl[X3 ? i : 0] = gkl[(size_t)st * KU + u];
        });
        static_for<0, NPV>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = min(tid + i * NT, VU - 1);
            preV[i] = gv[(size_t)st * VU + u];
            if (X3) preVl[X3 ? i : 0] = gvl[(size_t)st * VU + u];
        });
    };
    auto stage_store = [&](u32x4* Ld) {
        static_for<0, NPK>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + i * NT;
            if (KU % NT == 0 || u < KU) {
                Ld[u] = preK[i];
                if (X3) Ld[PU + u] = preKl[X3 ? i : 0];
            }
        });
        static_for<0, NPV>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + i * NT;
            if (VU % NT == 0 || u < VU) {
                Ld[KU + u] = preV[i];
                if (X3) Ld[PU + KU + u] = preVl[X3 ? i : 0];
            }
        });
    };

    if (st0 < st1) {
        stage_load(st0);
        stage_store(lds);
    }
    __syncthreads();

    for (int st = st0; st < st1; ++st) {
        const int cur = (st - st0) & 1;
        const u32x4* L = lds + cur * SU;
        const bool more = st + 1 < st1;
        if (more) stage_load(st + 1);

#pragma unroll
        for (int j = 0; j < KVS; ++j) {
            const int tile = st * KVS + j;
            const int key0 = tile * 32;
            if (key0 >= P.Nk) break;
            const u32x4* kl = L + (j * DKS) * 64 + lane;
            const u32x4* vl = L + KU + (j * 2 * DVB) * 64 + lane;

            f32x16 S;
#pragma unroll
            for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
            for (int s = 0; s < DKS; ++s) {
                const bf16x8 kf = as_bf16x8(kl[s * 64]);
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], S, 0, 0, 0);
                if (X3) {
                    S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, ql[s], S, 0, 0, 0);
                    const bf16x8 kfl = as_bf16x8(kl[PU + s * 64]);
                    S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qf[s], S, 0, 0, 0);
                }
            }

            const bool tail = key0 + 32 > P.Nk;
            float tmax, tmin = SID_INF;
            if (!tail) {
                tmax = fmaxf(fmaxf(S[0], S[1]), S[2]);
#pragma unroll
                for (int i = 3; i < 15; i += 2) tmax = fmaxf(fmaxf(tmax, S[i]), S[i + 1]);
                tmax = fmaxf(tmax, S[15]);
                if (need_min) {
                    tmin = fminf(fminf(S[0], S[1]), S[2]);
#pragma unroll
                    for (int i = 3; i < 15; i += 2) tmin = fminf(fminf(tmin, S[i]), S[i + 1]);
                    tmin = fminf(tmin, S[15]);
                }
            } else {
                tmax = -SID_INF;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool ok = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh < P.Nk;
                    tmax = ok ? fmaxf(tmax, S[i]) : tmax;
                    tmin = ok ? fminf(tmin, S[i]) : tmin;
                }
            }
            smax = fmaxf(smax, tmax);
            smin = fminf(smin, tmin);

            const float tref = (c2 >= 0.f) ? tmax * c2 : tmin * c2;   // -inf if this half has no valid key
            if (__any(tref > m2 + kRescaleThr)) {
                const float full = fmaxf(tref, __shfl_xor(tref, 32));
                const float mnew = fmaxf(m2, full);
                const float alpha = __builtin_amdgcn_exp2f(m2 - mnew);
#pragma unroll
                for (int b = 0; b < DVB; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) O[b][i] *= alpha;
                m2 = mnew;
            }

            float pv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) pv[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[i], c2, -m2));
            if (tail) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (!(key0 + (i & 3) + 8 * (i >> 2) + 4 * hh < P.Nk)) pv[i] = 0.f;
            }
            bf16x8 p0, p1;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) { p0[jj] = (__bf16)pv[jj]; p1[jj] = (__bf16)pv[8 + jj]; }
            bf16x8 p0l, p1l;
            if (X3) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    p0l[jj] = (__bf16)(pv[jj] - (float)p0[jj]);
                    p1l[jj] = (__bf16)(pv[8 + jj] - (float)p1[jj]);
                }
            }
#pragma unroll
            for (int b = 0; b < DVB; ++b) {
                const bf16x8 v0 = as_bf16x8(vl[(0 * DVB + b) * 64]);
                const bf16x8 v1 = as_bf16x8(vl[(1 * DVB + b) * 64]);
                O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p0, O[b], 0, 0, 0);
                O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p1, O[b], 0, 0, 0);
                if (X3) {
                    O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, p0l, O[b], 0, 0, 0);
                    O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p1l, O[b], 0, 0, 0);
                    const bf16x8 v0l = as_bf16x8(vl[PU + (0 * DVB + b) * 64]);
                    const bf16x8 v1l = as_bf16x8(vl[PU + (1 * DVB + b) * 64]);
                    O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0l, p0, O[b], 0, 0, 0);
                    O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1l, p1, O[b], 0, 0, 0);
                }
            }
        }

        if (more) stage_store(lds + (cur ^ 1) * SU);
        __syncthreads();
    }

    if (!wave_active) return;

    // ---- epilogue: combine the two lane halves of every query column and store the partial
    smax = fmaxf(smax, __shfl_xor(smax, 32));
    smin = fminf(smin, __shfl_xor(smin, 32));
    // softmax denominator = the ones column = row 31 of the last dv block = register 15, half 1
    float lsum = hh ? O[DVB - 1][15] : 0.f;
    lsum += __shfl_xor(lsum, 32);

    const int q = qt * 32 + (lane & 31);
    if (q < P.Nq) {
        const size_t row = ((size_t)part * P.BH + bh) * P.Nq + q;
        float* op = P.opart + row * P.d;
        const bool vec = (P.d & 3) == 0;
#pragma unroll
        for (int b = 0; b < DVB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dv0 = 32 * b + 8 * g + 4 * hh;
                if (vec) {
                    if (dv0 < P.d) {
                        f32x4 o = {O[b][4 * g], O[b][4 * g + 1], O[b][4 * g + 2], O[b][4 * g + 3]};
                        *(f32x4*)(op + dv0) = o;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (dv0 + e < P.d) op[dv0 + e] = O[b][4 * g + e];
                }
            }
        if (hh == 0) {
            f32x4 st = {m2, lsum, smax, smin};
            *(f32x4*)(P.stats + row * 4) = st;
        }
    }
}

// ------------------------------------------------------------------------------------------
// merge
// ------------------------------------------------------------------------------------------
struct MergeParams {
    const float* opart;
    const float* stats;
    const float* mask;     // (B, Nq) signed mask, or null when nblk == 1
    void* out;             // (B, Nq, H*d)
    float scale;           // content logits: cc = (q_cnt.k_cnt) * scale
    int B, H, Nq, d, KS, nblk;
    int out_bf16;
};

template <int VEC>
__global__ void __launch_bounds__(256) merge_kernel(MergeParams P) {
    const int dpv = P.d / VEC;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)P.B * P.H * P.Nq * dpv;
    if (gid >= total) return;
    const int c = (int)(gid % dpv);
    const long long rowid = gid / dpv;          // (b*H + h)*Nq + q
    const int q = (int)(rowid % P.Nq);
    const int bh = (int)(rowid / P.Nq);
    const int b = bh / P.H, h = bh % P.H;
    const int BH = P.B * P.H;
    const int nparts = P.nblk * P.KS;
    constexpr float LOG2E = 1.4426950408889634f;

    // block statistics
    float bias2[kBlocks] = {0.f, 0.f, 0.f};
    if (P.nblk == kBlocks) {
        float bmax[kBlocks], cmin = SID_INF;
#pragma unroll
        for (int blk = 0; blk < kBlocks; ++blk) {
            float mx = -SID_INF;
            for (int ks = 0; ks < P.KS; ++ks) {
                const f32x4 st = *(const f32x4*)(P.stats + (((size_t)(blk * P.KS + ks) * BH + bh) * P.Nq + q) * 4);
                mx = fmaxf(mx, st[2]);
                if (blk == kBlocks - 1) cmin = fminf(cmin, st[3]);
            }
            bmax[blk] = mx;
        }
        // min_j (q_cnt.k_cnt * scale): scaling by a constant is monotone, so it commutes with min
        const float min_cc = (P.scale >= 0.f) ? __fmul_rn(cmin, P.scale) : __fmul_rn(bmax[2], P.scale);
        const float m = P.mask[(size_t)b * P.Nq + q];
        bias2[0] = __fmul_rn(__fsub_rn(min_cc, bmax[0]), __fmul_rn(m, -1.0f)) * LOG2E;
        bias2[1] = __fmul_rn(__fsub_rn(min_cc, bmax[1]), __fmul_rn(m, 1.0f)) * LOG2E;
    }
    float M2 = -SID_INF;
    for (int p = 0; p < nparts; ++p) {
        const f32x4 st = *(const f32x4*)(P.stats + (((size_t)p * BH + bh) * P.Nq + q) * 4);
        if (st[1] > 0.f) M2 = fmaxf(M2, st[0] + bias2[p / P.KS]);
    }
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    float den = 0.f;
    for (int p = 0; p < nparts; ++p) {
        const size_t row = ((size_t)p * BH + bh) * P.Nq + q;
        const f32x4 st = *(const f32x4*)(P.stats + row * 4);
        if (!(st[1] > 0.f)) continue;
        const float w = __builtin_amdgcn_exp2f(st[0] + bias2[p / P.KS] - M2);
        den += w * st[1];
        const float* op = P.opart + row * P.d + c * VEC;
        if (VEC == 4) {
            const f32x4 o = *(const f32x4*)op;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += w * o[e];
        } else {
            acc[0] += w * op[0];
        }
    }
    const float inv = 1.0f / den;
    const size_t oidx = ((size_t)b * P.Nq + q) * P.H * P.d + (size_t)h * P.d + c * VEC;
    if (P.out_bf16) {
        __bf16* o = (__bf16*)P.out + oidx;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = (__bf16)(acc[e] * inv);
    } else {
        float* o = (float*)P.out + oidx;
        if (VEC == 4) {
            f32x4 r = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
            *(f32x4*)o = r;
        } else {
            o[0] = acc[0] * inv;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Geometry {
    int DKS, DVB;      // head_dim / 16 (rounded up), (head_dim + 1) / 32 (rounded up)
    int NW, KVS;       // waves per workgroup, 32-key tiles per LDS stage
    int NQT, NKT;      // tiles in the packed images
    int KS, SPS;       // key splits, stages per split
    int nQG;
    size_t q_units, k_units, v_units;   // u32x4 per packed tensor (one precision part)
};

bool supported_dims(int d, int* DKS, int* DVB) {
    const int dks = (d + 15) / 16, dvb = (d + 1 + 31) / 32;
    // instantiated (DKS, DVB) pairs
    static const int table[][2] = {{2, 1}, {2, 2}, {3, 2}, {4, 2}, {4, 3}, {5, 3}, {6, 4}, {8, 5}, {10, 6}};
    for (auto& t : table)
        if (t[0] == dks && t[1] == dvb) { *DKS = dks; *DVB = dvb; return true; }
    return false;
}

// 32-key tiles per LDS stage: two, or one where two double-buffered stages exceed the 160 KB LDS
constexpr int stage_tiles(int DKS, int DVB, bool x3) {
    return (size_t)2 * 2 * (DKS + 2 * DVB) * 1024 * (x3 ? 2 : 1) > 160 * 1024 ? 1 : 2;
}

int make_geometry(int B, int H, int Nq, int Nk, int d, int nblk, bool x3, Geometry* g) {
    if (!supported_dims(d, &g->DKS, &g->DVB))
        return sid::fail(SID_ENOSYS, "head_dim %d has no kernel instance (supported: 17-48, 49-64, 65-80, 81-96, 113-128, 145-160)", d);
    g->NW = 4;
    g->KVS = stage_tiles(g->DKS, g->DVB, x3);
    g->NQT = (Nq + 31) / 32;
    const int nkt = (Nk + 31) / 32;
    g->NKT = (nkt + g->KVS - 1) / g->KVS * g->KVS;
    g->nQG = (g->NQT + g->NW - 1) / g->NW;
    // key splits: enough workgroups to give every CU a few waves per SIMD
    const int nStages = g->NKT / g->KVS;
    const long long base_wg = (long long)g->nQG * B * H * nblk;
    int ks = 1;
    while (base_wg * ks < 768 && ks * 2 <= nStages && ks < 8) ks *= 2;
    g->SPS = (nStages + ks - 1) / ks;
    g->KS = (nStages + g->SPS - 1) / g->SPS;
    g->q_units = (size_t)B * H * g->NQT * g->DKS * 64;
    g->k_units = (size_t)B * H * g->NKT * g->DKS * 64;
    g->v_units = (size_t)B * H * g->NKT * 2 * g->DVB * 64;
    return SID_OK;
}

struct Workspace {
    u32x4* q[2][2];              // [which: 0 blended, 1 content][part: hi, lo]
    u32x4* k[kBlocks][2];
    u32x4* v[kBlocks][2];
    float* opart;
    float* stats;
    size_t bytes;
};

void carve(const Geometry& g, int B, int H, int Nq, int d, int nblk, int nq, bool x3, void* base, Workspace* w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = base ? (char*)base + off : nullptr; off += sid::align_up(bytes, 256); return p; };
    const int parts = x3 ? 2 : 1;
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 2; ++p) w->q[i][p] = nullptr;
    for (int i = 0; i < kBlocks; ++i) for (int p = 0; p < 2; ++p) { w->k[i][p] = nullptr; w->v[i][p] = nullptr; }
    for (int i = 0; i < nq; ++i) for (int p = 0; p < parts; ++p) w->q[i][p] = (u32x4*)take(g.q_units * 16);
    for (int i = 0; i < nblk; ++i) for (int p = 0; p < parts; ++p) w->k[i][p] = (u32x4*)take(g.k_units * 16);
    for (int i = 0; i < nblk; ++i) for (int p = 0; p < parts; ++p) w->v[i][p] = (u32x4*)take(g.v_units * 16);
    w->opart = (float*)take((size_t)nblk * g.KS * B * H * Nq * d * 4);
    w->stats = (float*)take((size_t)nblk * g.KS * B * H * Nq * 4 * 4);
    w->bytes = off;
}

template <int DKS, int DVB, bool X3>
int launch_flash_inst(const FlashParams& fp, const Geometry& g, hipStream_t stream) {
    constexpr int NW = 4, KVS = stage_tiles(DKS, DVB, X3);
    const size_t lds = (size_t)2 * (KVS * DKS + KVS * 2 * DVB) * 64 * 16 * (X3 ? 2 : 1);
    auto kern = flash_kernel<DKS, DVB, NW, KVS, X3>;
    if (lds > 64 * 1024) {
        static bool done = false;   // attribute is per function; set once per process
        if (!done) {
            SID_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            done = true;
        }
    }
    if (lds > 160 * 1024) return sid::fail(SID_ENOSYS, "head_dim needs %zu bytes of LDS per workgroup", lds);
    const int grid = g.nQG * fp.BH * fp.nblk * fp.KS;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, fp);
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}

template <bool X3>
int launch_flash(const FlashParams& fp, const Geometry& g, hipStream_t stream) {
#define SID_CASE(a, b) if (g.DKS == a && g.DVB == b) return launch_flash_inst<a, b, X3>(fp, g, stream);
    SID_CASE(2, 1) SID_CASE(2, 2) SID_CASE(3, 2) SID_CASE(4, 2) SID_CASE(4, 3) SID_CASE(5, 3)
    SID_CASE(6, 4) SID_CASE(8, 5) SID_CASE(10, 6)
#undef SID_CASE
    return sid::fail(SID_ENOSYS, "no flash kernel for DKS=%d DVB=%d", g.DKS, g.DVB);
}

void set_strides(PackJob& j, int layout, int H, int N, int d, bool second) {
    long long sb, sh, sn;
    if (layout == SID_LAYOUT_BHND) { sb = (long long)H * N * d; sh = (long long)N * d; sn = d; }
    else { sb = (long long)N * H * d; sh = d; sn = (long long)H * d; }
    if (second) { j.sb2 = sb; j.sh2 = sh; j.sn2 = sn; } else { j.sb = sb; j.sh = sh; j.sn = sn; }
}

int launch_pack(PackParams& pp, const long long* frags, hipStream_t stream) {
    int blocks = 0;
    for (int i = 0; i < pp.njobs; ++i) {
        pp.job[i].first_block = blocks;
        pp.job[i].frags = frags[i];
        blocks += (int)((frags[i] + 255) / 256);
    }
    if (pp.in_bf16) hipLaunchKernelGGL(pack_kernel<true>, dim3(blocks), dim3(256), 0, stream, pp);
    else hipLaunchKernelGGL(pack_kernel<false>, dim3(blocks), dim3(256), 0, stream, pp);
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}

struct Stages {
    bool pack, flash, merge;
    explicit Stages(int flags) {
        const int sel = flags & (SID_FLAG_STAGE_PACK | SID_FLAG_STAGE_FLASH | SID_FLAG_STAGE_MERGE);
        pack = sel == 0 || (sel & SID_FLAG_STAGE_PACK);
        flash = sel == 0 || (sel & SID_FLAG_STAGE_FLASH);
        merge = sel == 0 || (sel & SID_FLAG_STAGE_MERGE);
    }
};

int check_common(int B, int H, int Nq, int Nk, int d, int dtype, int compute) {
    SID_CHECK_ARG(B >= 1 && H >= 1 && Nq >= 1 && Nk >= 1, "B,H,N must be >= 1 (got B=%d H=%d Nq=%d Nk=%d)", B, H, Nq, Nk);
    SID_CHECK_ARG(d >= 1 && d <= 160, "head_dim %d out of range [1,160]", d);
    SID_CHECK_ARG(dtype == SID_DTYPE_F32 || dtype == SID_DTYPE_BF16, "unknown dtype %d", dtype);
    SID_CHECK_ARG(compute == SID_COMPUTE_BF16 || compute == SID_COMPUTE_F32X3, "unknown compute mode %d", compute);
    SID_CHECK_ARG((long long)B * H * (long long)std::max(Nq, Nk) * d < (1LL << 31), "tensor too large for 32-bit tile indexing");
    return SID_OK;
}

}  // namespace

extern "C" size_t sid_injected_attn_workspace_bytes(int B, int H, int N, int d, int compute) {
    Geometry g;
    if (check_common(B, H, N, N, d, SID_DTYPE_F32, compute) != SID_OK) return 0;
    if (make_geometry(B, H, N, N, d, kBlocks, compute == SID_COMPUTE_F32X3, &g) != SID_OK) return 0;
    Workspace w;
    carve(g, B, H, N, d, kBlocks, 2, compute == SID_COMPUTE_F32X3, nullptr, &w);
    return w.bytes;
}

extern "C" int sid_injected_attn_fwd(const void* x_q, const void* q_cnt, const void* k1, const void* v1,
                                     const void* k2, const void* v2, const void* k_cnt, const void* v_cnt,
                                     const float* mask_signed, void* out, int B, int H, int N, int d,
                                     float gamma, float q_self_coef, float T, float scale, int dtype,
                                     int compute, int flags, void* workspace, size_t workspace_bytes,
                                     void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const Stages stg(flags);
    int rc = check_common(B, H, N, N, d, dtype, compute);
    if (rc != SID_OK) return rc;
    SID_CHECK_ARG(x_q && q_cnt && k1 && v1 && k2 && v2 && k_cnt && v_cnt && mask_signed && out, "NULL tensor pointer");
    SID_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace must be non-NULL and 256-byte aligned");
    SID_CHECK_ARG(((uintptr_t)out & 15) == 0, "out must be 16-byte aligned");
    SID_CHECK_ARG(std::isfinite(gamma) && std::isfinite(q_self_coef) && std::isfinite(T) && std::isfinite(scale),
                  "gamma, q_self_coef, T and scale must be finite");
    Geometry g;
    rc = make_geometry(B, H, N, N, d, kBlocks, compute == SID_COMPUTE_F32X3, &g);
    if (rc != SID_OK) return rc;
    const bool x3 = compute == SID_COMPUTE_F32X3;
    Workspace w;
    carve(g, B, H, N, d, kBlocks, 2, x3, workspace, &w);
    if (w.bytes > workspace_bytes)
        return sid::fail(SID_ENOMEM, "workspace of %zu bytes is smaller than the %zu required", workspace_bytes, w.bytes);

    // ---- pack
    PackParams pp = {};
    pp.njobs = 8; pp.H = H; pp.d = d; pp.DKS = g.DKS; pp.DVB = g.DVB; pp.in_bf16 = dtype == SID_DTYPE_BF16;
    long long frags[kMaxJobs];
    auto rowjob = [&](int i, const void* src, u32x4* dst, u32x4* dst_lo, int NT, size_t units) {
        PackJob& j = pp.job[i];
        j = PackJob{};
        j.src = src; j.dst = dst; j.dst_lo = dst_lo; j.kind = 0; j.N = N; j.NT = NT;
        set_strides(j, SID_LAYOUT_BHND, H, N, d, false);
        frags[i] = (long long)units;
    };
    rowjob(0, q_cnt, w.q[0][0], w.q[0][1], g.NQT, g.q_units);      // blended query
    pp.job[0].src2 = x_q; pp.job[0].c1 = gamma; pp.job[0].c2 = q_self_coef;
    set_strides(pp.job[0], SID_LAYOUT_BNHD, H, N, d, true);
    rowjob(1, q_cnt, w.q[1][0], w.q[1][1], g.NQT, g.q_units);      // content query
    const void* ks[kBlocks] = {k1, k2, k_cnt};
    const void* vs[kBlocks] = {v1, v2, v_cnt};
    for (int b = 0; b < kBlocks; ++b) {
        rowjob(2 + b, ks[b], w.k[b][0], w.k[b][1], g.NKT, g.k_units);
        rowjob(5 + b, vs[b], w.v[b][0], w.v[b][1], g.NKT, g.v_units);
        pp.job[5 + b].kind = 1;
    }
    if (stg.pack) {
        rc = launch_pack(pp, frags, stream);
        if (rc != SID_OK) return rc;
    }

    // ---- flash
    constexpr float LOG2E = 1.4426950408889634f;
    FlashParams fp = {};
    for (int b = 0; b < kBlocks; ++b) {
        const int qi = (b == kBlocks - 1) ? 1 : 0;
        fp.q[b] = w.q[qi][0]; fp.q_lo[b] = w.q[qi][1];
        fp.k[b] = w.k[b][0]; fp.k_lo[b] = w.k[b][1];
        fp.v[b] = w.v[b][0]; fp.v_lo[b] = w.v[b][1];
    }
    fp.c2[0] = fp.c2[1] = T * scale * LOG2E;
    fp.c2[2] = scale * LOG2E;
    fp.opart = w.opart; fp.stats = w.stats;
    fp.BH = B * H; fp.Nq = N; fp.Nk = N; fp.d = d; fp.NQT = g.NQT; fp.NKT = g.NKT; fp.nQG = g.nQG;
    fp.KS = g.KS; fp.SPS = g.SPS; fp.nblk = kBlocks;
    if (stg.flash) {
        rc = x3 ? launch_flash<true>(fp, g, stream) : launch_flash<false>(fp, g, stream);
        if (rc != SID_OK) return rc;
    }
    if (!stg.merge) return SID_OK;

    // ---- merge
    MergeParams mp = {};
    mp.opart = w.opart; mp.stats = w.stats; mp.mask = mask_signed; mp.out = out; mp.scale = scale;
    mp.B = B; mp.H = H; mp.Nq = N; mp.d = d; mp.KS = g.KS; mp.nblk = kBlocks; mp.out_bf16 = dtype == SID_DTYPE_BF16;
    if ((d & 3) == 0) {
        const long long total = (long long)B * H * N * (d / 4);
        hipLaunchKernelGGL(merge_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, mp);
    } else {
        const long long total = (long long)B * H * N * d;
        hipLaunchKernelGGL(merge_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, mp);
    }
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}

extern "C" size_t sid_plain_attn_workspace_bytes(int B, int H, int Nq, int Nk, int d, int compute) {
    Geometry g;
    if (check_common(B, H, Nq, Nk, d, SID_DTYPE_F32, compute) != SID_OK) return 0;
    if (make_geometry(B, H, Nq, Nk, d, 1, compute == SID_COMPUTE_F32X3, &g) != SID_OK) return 0;
    Workspace w;
    carve(g, B, H, Nq, d, 1, 1, compute == SID_COMPUTE_F32X3, nullptr, &w);
    return w.bytes;
}

extern "C" int sid_plain_attn_fwd(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq,
                                  int Nk, int d, float logit_scale, int q_layout, int kv_layout, int dtype,
                                  int compute, int flags, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const Stages stg(flags);
    int rc = check_common(B, H, Nq, Nk, d, dtype, compute);
    if (rc != SID_OK) return rc;
    SID_CHECK_ARG(q && k && v && out, "NULL tensor pointer");
    SID_CHECK_ARG(q_layout == SID_LAYOUT_BHND || q_layout == SID_LAYOUT_BNHD, "unknown q_layout %d", q_layout);
    SID_CHECK_ARG(kv_layout == SID_LAYOUT_BHND || kv_layout == SID_LAYOUT_BNHD, "unknown kv_layout %d", kv_layout);
    SID_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace must be non-NULL and 256-byte aligned");
    SID_CHECK_ARG(((uintptr_t)out & 15) == 0, "out must be 16-byte aligned");
    SID_CHECK_ARG(std::isfinite(logit_scale), "logit_scale must be finite");
    Geometry g;
    rc = make_geometry(B, H, Nq, Nk, d, 1, compute == SID_COMPUTE_F32X3, &g);
    if (rc != SID_OK) return rc;
    const bool x3 = compute == SID_COMPUTE_F32X3;
    Workspace w;
    carve(g, B, H, Nq, d, 1, 1, x3, workspace, &w);
    if (w.bytes > workspace_bytes)
        return sid::fail(SID_ENOMEM, "workspace of %zu bytes is smaller than the %zu required", workspace_bytes, w.bytes);

    PackParams pp = {};
    pp.njobs = 3; pp.H = H; pp.d = d; pp.DKS = g.DKS; pp.DVB = g.DVB; pp.in_bf16 = dtype == SID_DTYPE_BF16;
    long long frags[kMaxJobs];
    auto job = [&](int i, const void* src, u32x4* dst, u32x4* dst_lo, int kind, int N, int NT, size_t units, int layout) {
        PackJob& j = pp.job[i];
        j = PackJob{};
        j.src = src; j.dst = dst; j.dst_lo = dst_lo; j.kind = kind; j.N = N; j.NT = NT;
        set_strides(j, layout, H, N, d, false);
        frags[i] = (long long)units;
    };
    job(0, q, w.q[0][0], w.q[0][1], 0, Nq, g.NQT, g.q_units, q_layout);
    job(1, k, w.k[0][0], w.k[0][1], 0, Nk, g.NKT, g.k_units, kv_layout);
    job(2, v, w.v[0][0], w.v[0][1], 1, Nk, g.NKT, g.v_units, kv_layout);
    if (stg.pack) {
        rc = launch_pack(pp, frags, stream);
        if (rc != SID_OK) return rc;
    }

    constexpr float LOG2E = 1.4426950408889634f;
    FlashParams fp = {};
    fp.q[0] = w.q[0][0]; fp.q_lo[0] = w.q[0][1];
    fp.k[0] = w.k[0][0]; fp.k_lo[0] = w.k[0][1];
    fp.v[0] = w.v[0][0]; fp.v_lo[0] = w.v[0][1];
    fp.c2[0] = logit_scale * LOG2E;
    fp.opart = w.opart; fp.stats = w.stats;
    fp.BH = B * H; fp.Nq = Nq; fp.Nk = Nk; fp.d = d; fp.NQT = g.NQT; fp.NKT = g.NKT; fp.nQG = g.nQG;
    fp.KS = g.KS; fp.SPS = g.SPS; fp.nblk = 1;
    if (stg.flash) {
        rc = x3 ? launch_flash<true>(fp, g, stream) : launch_flash<false>(fp, g, stream);
        if (rc != SID_OK) return rc;
    }
    if (!stg.merge) return SID_OK;

    MergeParams mp = {};
    mp.opart = w.opart; mp.stats = w.stats; mp.mask = nullptr; mp.out = out; mp.scale = 1.f;
    mp.B = B; mp.H = H; mp.Nq = Nq; mp.d = d; mp.KS = g.KS; mp.nblk = 1; mp.out_bf16 = dtype == SID_DTYPE_BF16;
    if ((d & 3) == 0) {
        const long long total = (long long)B * H * Nq * (d / 4);
        hipLaunchKernelGGL(merge_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, mp);
    } else {
        const long long total = (long long)B * H * Nq * d;
        hipLaunchKernelGGL(merge_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, mp);
    }
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}
