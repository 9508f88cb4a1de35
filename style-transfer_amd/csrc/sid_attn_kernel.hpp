// Flash-style attention kernel of libsid.so (gfx950): device code, parameters and launch template.
//
// Included by the two instantiation units (bf16 products / split-bf16 f32-grade products) and by
// the host-side entry points in sid_attn.hip.
//
// Operands arrive pre-packed in MFMA fragment order (pack_kernel in sid_attn.hip), so a 32-key tile
// of K or V is a contiguous run of 1 KiB units that is copied to LDS verbatim with direct
// global->LDS loads and read back with lane-linear, conflict-free ds_read_b128.
//
// MFMA: v_mfma_f32_32x32x16_bf16.  Lane l = 32*hh + r holds A[row r][k = 8hh+j], B[k = 8hh+j][col r]
// (j = 0..7) and D[row = (i&3) + 8(i>>2) + 4hh][col r] in register i = 0..15.
//   S^T = K Q^T : keys on MFMA rows, queries on lanes.  The accumulator of S^T, rounded to bf16, is
//   already the B operand of  O^T += V^T P^T  (dv on rows, queries on lanes): P never leaves
//   registers, and every per-query quantity (running max, scale, sum) is a per-lane scalar.
#pragma once
#include "sid_common.h"

#include <cstdint>
#include <type_traits>

namespace sid_attn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // 16-byte unit (a native vector: stays in registers)

constexpr int kBlocks = 3;            // key blocks of the injected attention: char, back, content
constexpr float kOverflowThr = 96.f;  // log2 units a logit may exceed the fixed reference by
constexpr size_t kLdsBytes = 160 * 1024;
#define SID_INF __builtin_huge_valf()

struct FlashParams {
    const u32x4* q[kBlocks];     // packed query used by each key block (hi part)
    const u32x4* q_lo[kBlocks];
    const u32x4* k[kBlocks];
    const u32x4* k_lo[kBlocks];
    const u32x4* v[kBlocks];
    const u32x4* v_lo[kBlocks];
    float c2[kBlocks];           // logit scale of each block, log2 domain: x = (q.k) * c2
    const float* mask;           // (B, Nq) signed mask; null when nblk == 1
    void* out;                   // (B, Nq, H*d)
    float scale;                 // content logits are (q_cnt.k_cnt) * scale (for min_j cc)
    int B, H, Nq, Nk, d;
    int NQT, NKT;                // 32-row tiles in the packed query / key images
    int nQG;                     // workgroups per head: ceil(NQT / NQW)
    int KS;                      // key splits per block; parts = nblk * KS
    int SPS;                     // stages per split
    int nblk;                    // 3: injected [char | back | content], 1: single block
    int nbuf;                    // LDS stage buffers per part: 2, or 1 where two do not fit
    int out_bf16;
    int flag_off;                // byte offset in LDS of the workgroup's "redo" flag (after all buffers)
};

// Launch geometry chosen on the host (make_geometry in sid_attn.hip).
struct Geometry {
    int DKS, DVB;      // head_dim / 16 (rounded up), (head_dim + 1) / 32 (rounded up)
    int KVS;           // 32-key tiles per LDS stage
    int NBUF;          // stage buffers per part
    int NQW, KS;       // query tiles per part, key splits per block
    int SPS;           // stages per split
    int NQT, NKT;      // tiles in the packed images
    int nQG;
    int threads;
    size_t lds;
    size_t q_units, k_units, v_units;   // u32x4 per packed tensor (one precision part)
};

// Threads per workgroup the register budget allows: an estimate of the live registers of one wave
// (accumulators O and two logit tiles, query / key / value fragments, P, addressing) picks 3, 2 or
// 1 waves per SIMD.
constexpr int max_threads(int DKS, int DVB, bool x3) {
    const int x = x3 ? 2 : 1;
    const int est = 16 * DVB + 32 + 4 * DKS * x + 4 * DKS * x + 8 * DVB * x + 8 * x + 30;
    // never below 512 threads: a 512-register (1 wave per SIMD) build of this kernel crashes hipcc 7.2
    // in its AGPR rewrite pass; the widest head dims spill a few registers instead
    return est <= 154 ? 768 : 512;
}

int launch_flash_bf16(const FlashParams& fp, const Geometry& g, hipStream_t stream);
int launch_flash_x3(const FlashParams& fp, const Geometry& g, hipStream_t stream);

#if defined(__HIPCC__)

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct RowState {
    float m2;      // log2-domain reference the exponentials are taken against
    float smax;    // running max of the raw q.k (this lane half's keys)
    float smin;    // running min (tracked only where needed)
    float ovf;     // largest (logit - m2) seen while m2 was held fixed
};

__device__ __forceinline__ float max16(const f32x16& S) {
    float t = fmaxf(fmaxf(S[0], S[1]), S[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) t = fmaxf(fmaxf(t, S[i]), S[i + 1]);
    return fmaxf(t, S[15]);
}
__device__ __forceinline__ float min16(const f32x16& S) {
    float t = fminf(fminf(S[0], S[1]), S[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) t = fminf(fminf(t, S[i]), S[i + 1]);
    return fminf(t, S[15]);
}

// Fragments of one tile read from an LDS (or global) image: NF units of 16 bytes, 64 units apart.
template <int NF, bool X3>
struct Frags {
    bf16x8 hi[NF];
    bf16x8 lo[X3 ? NF : 1];
    __device__ __forceinline__ void load(const u32x4* p, int lo_off) {
#pragma unroll
        for (int i = 0; i < NF; ++i) hi[i] = as_bf16x8(p[i * 64]);
        if (X3) {
#pragma unroll
            for (int i = 0; i < NF; ++i) lo[X3 ? i : 0] = as_bf16x8(p[lo_off + i * 64]);
        }
    }
};

// S^T = K Q^T for one tile of 32 keys (K fragments of the DKS head_dim steps in kf).
template <int DKS, bool X3>
__device__ __forceinline__ f32x16 qk_tile(const Frags<DKS, X3>& kf, const bf16x8 (&qf)[DKS],
                                          const bf16x8 (&ql)[X3 ? DKS : 1]) {
    f32x16 S;
#pragma unroll
    for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
    for (int s = 0; s < DKS; ++s) {
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.hi[s], qf[s], S, 0, 0, 0);
        if (X3) {
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.hi[s], ql[X3 ? s : 0], S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.lo[X3 ? s : 0], qf[s], S, 0, 0, 0);
        }
    }
    return S;
}

// Softmax step for one tile whose logits S are ready, then O^T += V^T P^T.
//   RESCALE = true : exact online softmax, the reference m2 follows the running row max and O is
//                    rescaled every tile (used for the first stage, tails, and the fallback pass).
//   RESCALE = false: m2 stays fixed; the block only records how far a logit exceeded it.  The
//                    body is then straight-line code: max, fma, exp2, cvt, MFMA.
//   TAIL           : keys >= Nk are masked.
// vf holds the V^T fragments of the tile: index s2 * DVB + b (k-step s2, dv block b).
template <int DVB, bool X3, bool TAIL, bool TRACK_MIN, bool RESCALE>
__device__ __forceinline__ void softmax_pv_tile(const f32x16& S, const Frags<2 * DVB, X3>& vf, int hh, int key0,
                                                int Nk, float c2, f32x16 (&O)[DVB], RowState& rs) {
    float tmax, tmin = SID_INF;
    if (!TAIL) {
        tmax = max16(S);
        if (TRACK_MIN) tmin = min16(S);
    } else {
        tmax = -SID_INF;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh < Nk;
            tmax = ok ? fmaxf(tmax, S[i]) : tmax;
            tmin = ok ? fminf(tmin, S[i]) : tmin;
        }
    }
    rs.smax = fmaxf(rs.smax, tmax);
    if (TRACK_MIN || TAIL) rs.smin = fminf(rs.smin, tmin);

    const float tref = (c2 >= 0.f) ? tmax * c2 : tmin * c2;     // -inf where this half has no valid key
    if (RESCALE) {
        const float full = fmaxf(tref, __shfl_xor(tref, 32));
        const float mnew = fmaxf(rs.m2, full);
        const float alpha = __builtin_amdgcn_exp2f(rs.m2 - mnew);
#pragma unroll
        for (int b = 0; b < DVB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[b][i] *= alpha;
        rs.m2 = mnew;
    } else {
        rs.ovf = fmaxf(rs.ovf, tref - rs.m2);
    }
    const float m2 = rs.m2;

    float pv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pv[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[i], c2, -m2));
    if (TAIL) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (!(key0 + (i & 3) + 8 * (i >> 2) + 4 * hh < Nk)) pv[i] = 0.f;
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
        O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.hi[b], p0, O[b], 0, 0, 0);
        O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.hi[DVB + b], p1, O[b], 0, 0, 0);
        if (X3) {
            O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.hi[b], p0l, O[b], 0, 0, 0);
            O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.hi[DVB + b], p1l, O[b], 0, 0, 0);
            O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.lo[X3 ? b : 0], p0, O[b], 0, 0, 0);
            O[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.lo[X3 ? DVB + b : 0], p1, O[b], 0, 0, 0);
        }
    }
}

// One tile phase of the software pipeline: the QK^T MFMAs of tile t+1 (K slot j) are issued first,
// then the VALU softmax of tile t runs while the matrix pipe works, then its PV MFMAs (V slot j).
template <int DKS, int DVB, int KVS, bool X3, bool TAIL, bool TRACK_MIN, bool RESCALE>
__device__ __forceinline__ void tile_phase(const u32x4* L, int j, int lane, int hh, int key0, int Nk, float c2,
                                           const bf16x8 (&qf)[DKS], const bf16x8 (&ql)[X3 ? DKS : 1],
                                           const f32x16& Scur, f32x16& Snext, f32x16 (&O)[DVB], RowState& rs) {
    constexpr int KU = KVS * DKS * 64;
    constexpr int PU = KU + KVS * 2 * DVB * 64;
    Frags<DKS, X3> kf;
    kf.load(L + (j * DKS) * 64 + lane, PU);
    Snext = qk_tile<DKS, X3>(kf, qf, ql);
    if (!TAIL || key0 < Nk) {
        Frags<2 * DVB, X3> vf;
        vf.load(L + KU + (j * 2 * DVB) * 64 + lane, PU);
        softmax_pv_tile<DVB, X3, TAIL, TRACK_MIN, RESCALE>(Scur, vf, hh, key0, Nk, c2, O, rs);
    }
}

// One LDS stage of KVS tiles.  On entry Sa holds the logits of the stage's first tile, on exit
// those of the next stage's first tile (the K image of a stage is shifted by one tile for this).
template <int DKS, int DVB, int KVS, bool X3, bool TAIL, bool TRACK_MIN, bool RESCALE>
__device__ __forceinline__ void process_stage(const u32x4* L, int lane, int hh, int key0, int Nk, float c2,
                                              const bf16x8 (&qf)[DKS], const bf16x8 (&ql)[X3 ? DKS : 1],
                                              f32x16& Sa, f32x16 (&O)[DVB], RowState& rs) {
    f32x16 Sb;
    if (KVS == 2) {
        tile_phase<DKS, DVB, KVS, X3, TAIL, TRACK_MIN, RESCALE>(L, 0, lane, hh, key0, Nk, c2, qf, ql, Sa, Sb, O, rs);
        tile_phase<DKS, DVB, KVS, X3, TAIL, TRACK_MIN, RESCALE>(L, 1, lane, hh, key0 + 32, Nk, c2, qf, ql, Sb, Sa, O, rs);
    } else {
        tile_phase<DKS, DVB, KVS, X3, TAIL, TRACK_MIN, RESCALE>(L, 0, lane, hh, key0, Nk, c2, qf, ql, Sa, Sb, O, rs);
        Sa = Sb;
    }
}

// Direct global->LDS load of one 1 KiB unit: lane l's 16 bytes at src + l land at dst + l.
// Written as asm on purpose: the compiler then neither counts the load nor orders LDS reads behind
// it (with the builtin it waits vmcnt(0) before the next ds_read, which serialises the prefetch
// with the compute it is meant to overlap).  The kernel waits for these loads itself
// (wait_direct_loads) before the barrier that publishes a stage.
__device__ __forceinline__ void glds16(const u32x4* src_uniform, unsigned lane_bytes, u32x4* dst) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)dst));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_bytes), "s"(src_uniform), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void wait_direct_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Workgroup = NQW query tiles x NP parts, one wave each.  A part is one key block (or one key range
// of it); the NQW waves of a part share its K/V stages in LDS.  After the key loop the parts of a
// query tile are combined through LDS and the workgroup writes the final rows.
template <int DKS, int DVB, int KVS, bool X3, int NQW, int MAXT>
__global__ void __launch_bounds__(MAXT) flash_kernel(FlashParams P) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    constexpr int KU = KVS * DKS * 64;        // u32x4 per stage: K image
    constexpr int VU = KVS * 2 * DVB * 64;    //                  V image
    constexpr int PU = KU + VU;               // one precision part
    constexpr int SU = PU * (X3 ? 2 : 1);     // one stage: [K hi | V hi | K lo | V lo]
    constexpr int KUn = KU / 64, VUn = VU / 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;
    const int NP = P.nblk * P.KS;
    const int part = wave / NQW, qi = wave % NQW;
    const int blk = part / P.KS, ks = part - blk * P.KS;

    // workgroup -> (head, query group); workgroups of one head get ids equal mod 8 when the head
    // count allows, i.e. they share an XCD and find the head's K/V in its L2 (speed only)
    const int BH = P.B * P.H;
    int bh, qg;
    {
        const int id = blockIdx.x;
        if ((BH & 7) == 0) {
            const int slot = id >> 3;
            bh = (id & 7) + 8 * (slot / P.nQG);
            qg = slot % P.nQG;
        } else {
            bh = id / P.nQG;
            qg = id % P.nQG;
        }
    }
    const int qt_raw = qg * NQW + qi;
    const int qt = min(qt_raw, P.NQT - 1);    // surplus waves redo the last tile and store nothing

    bf16x8 qf[DKS];
    bf16x8 ql[X3 ? DKS : 1];
    {
        const u32x4* qs = P.q[blk] + ((size_t)(bh * P.NQT + qt) * DKS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < DKS; ++s) qf[s] = as_bf16x8(qs[s * 64]);
        if (X3) {
            const u32x4* ql_ = P.q_lo[blk] + ((size_t)(bh * P.NQT + qt) * DKS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < DKS; ++s) ql[X3 ? s : 0] = as_bf16x8(ql_[s * 64]);
        }
    }

    const float c2 = P.c2[blk];
    const bool need_min = (P.nblk == kBlocks && blk == kBlocks - 1) || c2 < 0.f;
    const int nStages = P.NKT / KVS;
    const int st0 = ks * P.SPS;
    const int st1 = min(st0 + P.SPS, nStages);

    // The K image holds NKT+1 tiles (the last one zero): stage s of K is tiles s*KVS+1 .. s*KVS+KVS,
    // one tile ahead of the V stage, because QK^T runs one tile ahead of softmax and PV.
    // (wave-uniform pointers: the lane offset is added at each use so the addressing stays scalar)
    const u32x4* gk0 = P.k[blk] + (size_t)bh * (P.NKT + 1) * DKS * 64;
    const u32x4* gv = P.v[blk] + (size_t)bh * P.NKT * 2 * DVB * 64;
    const u32x4* gkl0 = X3 ? P.k_lo[blk] + (size_t)bh * (P.NKT + 1) * DKS * 64 : nullptr;
    const u32x4* gvl = X3 ? P.v_lo[blk] + (size_t)bh * P.NKT * 2 * DVB * 64 : nullptr;
    u32x4* Lp = lds + (size_t)part * P.nbuf * SU;   // this part's stage buffers

    // Stage copy, global -> LDS without registers: every 1 KiB unit is one wave-wide direct load.
    // The part's NQW waves take the units of each image round-robin.
    auto issue_stage = [&](int st, u32x4* dst) {
        const u32x4* sk = gk0 + DKS * 64 + (size_t)st * KU;
        const u32x4* sv = gv + (size_t)st * VU;
        static_for<0, (KUn + NQW - 1) / NQW>([&](auto ic) {
            const int u = qi + decltype(ic)::value * NQW;
            if (KUn % NQW == 0 || u < KUn) {
                glds16(sk + u * 64, lane * 16, dst + u * 64);
                if (X3) glds16(gkl0 + DKS * 64 + (size_t)st * KU + u * 64, lane * 16, dst + PU + u * 64);
            }
        });
        static_for<0, (VUn + NQW - 1) / NQW>([&](auto ic) {
            const int u = qi + decltype(ic)::value * NQW;
            if (VUn % NQW == 0 || u < VUn) {
                glds16(sv + u * 64, lane * 16, dst + KU + u * 64);
                if (X3) glds16(gvl + (size_t)st * VU + u * 64, lane * 16, dst + PU + KU + u * 64);
            }
        });
    };

    f32x16 O[DVB];
    RowState rs;
    f32x16 Scur;
    const bool dbuf = P.nbuf == 2;
    volatile int* redo = (volatile int*)((char*)lds + P.flag_off);
    if (tid == 0) *redo = 0;

    // Stages [it0, it1) of this part with ONE variant of the stage body per loop, so that every loop
    // is straight-line code.  Every wave of the workgroup runs SPS stages in total (parts whose key
    // range ends early run empty tail stages), which keeps the barrier count uniform.
    auto run = [&](auto tail_c, auto min_c, auto resc_c, int it0, int it1) {
        constexpr bool TAIL = decltype(tail_c)::value, TMIN = decltype(min_c)::value, RESC = decltype(resc_c)::value;
        for (int it = it0; it < it1; ++it) {
            // Nothing the compiler counts may be outstanding once the direct loads are in flight: its
            // counted vmcnt waits do not know about them and would wait for the oldest one in every
            // iteration.  At this point nothing is in flight (the previous iteration drained it), so
            // this wait is free; it only resets the compiler's bookkeeping.
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            const int st = st0 + it;
#ifndef SID_EXP_NOSTAGE
            if (dbuf && it + 1 < P.SPS) issue_stage(st + 1, Lp + ((it + 1) & 1) * SU);
#endif
            process_stage<DKS, DVB, KVS, X3, TAIL, TMIN, RESC>(Lp + (dbuf ? (it & 1) : 0) * SU, lane, hh,
                                                               st * KVS * 32, P.Nk, c2, qf, ql, Scur, O, rs);
#ifndef SID_EXP_NOSYNC
            wait_direct_loads();
            __syncthreads();          // stage it+1 has landed in every wave's view; stage it is free
#endif
            if (!dbuf) {              // single buffer: refill only now, and wait for it
                if (it + 1 < P.SPS) issue_stage(st + 1, Lp);
                wait_direct_loads();
                __syncthreads();
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    // stages of this part that hold only valid keys
    const int nfull = max(0, min(P.SPS, P.Nk / (KVS * 32) - st0));

    // Pass 0 holds the softmax reference fixed after the first stage (no rescaling in the loop) and
    // records the largest excess over it.  If some row of the workgroup exceeded the safe range
    // (data whose logits grow by more than 2^96 along the keys), pass 1 redoes the workgroup with
    // the exact online rescale in every tile.
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int b = 0; b < DVB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[b][i] = 0.f;
        rs.m2 = -SID_INF; rs.smax = -SID_INF; rs.smin = SID_INF; rs.ovf = -SID_INF;

        issue_stage(st0, Lp);
        {   // logits of the part's first tile, K fragments straight from the packed image
            Frags<DKS, X3> kf;
            kf.load(gk0 + (size_t)st0 * KU + lane, X3 ? (int)(gkl0 - gk0) : 0);
            Scur = qk_tile<DKS, X3>(kf, qf, ql);
        }
        wait_direct_loads();
        __syncthreads();

        if (pass == 0) {
            run(F_{}, T_{}, T_{}, 0, min(1, nfull));                  // first stage: exact running max
            if (need_min) run(F_{}, T_{}, F_{}, 1, nfull);            // steady state, fixed reference
            else run(F_{}, F_{}, F_{}, 1, nfull);
        } else {
            run(F_{}, T_{}, T_{}, 0, nfull);
        }
        run(T_{}, T_{}, T_{}, nfull, P.SPS);                          // stages with masked keys

        if (pass == 0 && __any(rs.ovf > kOverflowThr) && lane == 0) *redo = 1;
        __syncthreads();
#ifdef SID_EXP_NOREDO
        break;
#endif
        if (pass == 1 || *redo == 0) break;
    }
    __syncthreads();

    // ---- combine the parts of each query tile through LDS (the stage buffers are free now)
    const int d = P.d;
    const int RS = ((d + 3) & ~3) + 4;         // floats per row: O[0..d) then (m2, l, smax, smin)
    float* M = (float*)lds;
    {
        float smax = fmaxf(rs.smax, __shfl_xor(rs.smax, 32));
        float smin = fminf(rs.smin, __shfl_xor(rs.smin, 32));
        // softmax denominator = the ones column = row 31 of the last dv block = register 15, half 1
        float lsum = hh ? O[DVB - 1][15] : 0.f;
        lsum += __shfl_xor(lsum, 32);
        float* row = M + ((size_t)wave * 32 + (lane & 31)) * RS;
        const bool vec = (d & 3) == 0;
#pragma unroll
        for (int b = 0; b < DVB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dv0 = 32 * b + 8 * g + 4 * hh;
                if (vec) {
                    if (dv0 < d) {
                        f32x4 o = {O[b][4 * g], O[b][4 * g + 1], O[b][4 * g + 2], O[b][4 * g + 3]};
                        *(f32x4*)(row + dv0) = o;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (dv0 + e < d) row[dv0 + e] = O[b][4 * g + e];
                }
            }
        if (hh == 0) {
            f32x4 st = {rs.m2, lsum, smax, smin};
            *(f32x4*)(row + RS - 4) = st;
        }
    }
    __syncthreads();

    constexpr float LOG2E = 1.4426950408889634f;
    const bool vec = (d & 3) == 0;
    const int CH = vec ? d / 4 : d;
    const int items = NQW * 32 * CH;
    const int b_ = bh / P.H, h_ = bh % P.H;
    for (int item = tid; item < items; item += blockDim.x) {
        const int c = item % CH;
        const int r = (item / CH) & 31;
        const int qi2 = item / (CH * 32);
        const int qt2 = qg * NQW + qi2;
        const int q = qt2 * 32 + r;
        if (qt2 >= P.NQT || q >= P.Nq) continue;
        auto stat = [&](int p) { return *(const f32x4*)(M + ((size_t)(p * NQW + qi2) * 32 + r) * RS + RS - 4); };
        float bias2[kBlocks] = {0.f, 0.f, 0.f};
        if (P.nblk == kBlocks) {
            float bmax[kBlocks], cmin = SID_INF;
#pragma unroll
            for (int bk = 0; bk < kBlocks; ++bk) {
                float mx = -SID_INF;
                for (int s = 0; s < P.KS; ++s) {
                    const f32x4 st = stat(bk * P.KS + s);
                    mx = fmaxf(mx, st[2]);
                    if (bk == kBlocks - 1) cmin = fminf(cmin, st[3]);
                }
                bmax[bk] = mx;
            }
            // min_j (q_cnt.k_cnt * scale): scaling by a constant is monotone, so it commutes with min
            const float min_cc = (P.scale >= 0.f) ? __fmul_rn(cmin, P.scale) : __fmul_rn(bmax[2], P.scale);
            const float m = P.mask[(size_t)b_ * P.Nq + q];
            bias2[0] = __fmul_rn(__fsub_rn(min_cc, bmax[0]), __fmul_rn(m, -1.0f)) * LOG2E;
            bias2[1] = __fmul_rn(__fsub_rn(min_cc, bmax[1]), __fmul_rn(m, 1.0f)) * LOG2E;
        }
        // reference of the combined softmax: the largest biased logit of the row (true maxima, not
        // the parts' fixed references, which a logit may exceed by up to 2^kOverflowThr)
        float M2 = -SID_INF;
        for (int p = 0; p < NP; ++p) {
            const f32x4 st = stat(p);
            const float cp = P.c2[p / P.KS];
            if (st[1] > 0.f) M2 = fmaxf(M2, (cp >= 0.f ? st[2] : st[3]) * cp + bias2[p / P.KS]);
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float den = 0.f;
        for (int p = 0; p < NP; ++p) {
            const f32x4 st = stat(p);
            if (!(st[1] > 0.f)) continue;
            const float w = __builtin_amdgcn_exp2f(st[0] + bias2[p / P.KS] - M2);
            den += w * st[1];
            const float* op = M + ((size_t)(p * NQW + qi2) * 32 + r) * RS;
            if (vec) {
                const f32x4 o = *(const f32x4*)(op + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += w * o[e];
            } else {
                acc[0] += w * op[c];
            }
        }
        const float inv = 1.0f / den;
        const size_t oidx = ((size_t)b_ * P.Nq + q) * P.H * d + (size_t)h_ * d + (vec ? 4 * c : c);
        if (P.out_bf16) {
            __bf16* o = (__bf16*)P.out + oidx;
            if (vec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)(acc[e] * inv);
            } else {
                o[0] = (__bf16)(acc[0] * inv);
            }
        } else {
            float* o = (float*)P.out + oidx;
            if (vec) {
                f32x4 rr = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
                *(f32x4*)o = rr;
            } else {
                o[0] = acc[0] * inv;
            }
        }
    }
}

template <int DKS, int DVB, int KVS, bool X3, int NQW>
int launch_flash_inst(const FlashParams& fp, const Geometry& g, hipStream_t stream) {
    constexpr int MAXT = max_threads(DKS, DVB, X3);
    auto kern = flash_kernel<DKS, DVB, KVS, X3, NQW, MAXT>;
    static bool done = false;   // one device per process: the attribute is set once per instance
    if (!done) {
        SID_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        done = true;
    }
    const long long grid = (long long)g.nQG * fp.B * fp.H;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(g.threads), g.lds + 16, stream, fp);
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}

template <bool X3>
int launch_flash(const FlashParams& fp, const Geometry& g, hipStream_t stream) {
#define SID_NQW(a, b, k)                                                                     \
    switch (g.NQW) {                                                                         \
        case 4: return launch_flash_inst<a, b, k, X3, 4>(fp, g, stream);                     \
        case 2: return launch_flash_inst<a, b, k, X3, 2>(fp, g, stream);                     \
        default: return launch_flash_inst<a, b, k, X3, 1>(fp, g, stream);                    \
    }
#define SID_CASE(a, b)                                                                       \
    if (g.DKS == a && g.DVB == b) {                                                          \
        if (g.KVS == 2) { SID_NQW(a, b, 2) } else { SID_NQW(a, b, 1) }                        \
    }
    SID_CASE(2, 1) SID_CASE(2, 2) SID_CASE(3, 2) SID_CASE(4, 2) SID_CASE(4, 3) SID_CASE(5, 3)
    SID_CASE(6, 4) SID_CASE(8, 5) SID_CASE(10, 6)
#undef SID_CASE
#undef SID_NQW
    return sid::fail(SID_ENOSYS, "no flash kernel for DKS=%d DVB=%d", g.DKS, g.DVB);
}

#endif  // __HIPCC__

}  // namespace sid_attn
