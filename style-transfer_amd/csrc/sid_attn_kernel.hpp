// Flash-style attention kernels of libsid.so (gfx950): device code, parameters and launch templates.
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
//
// Decomposition: a workgroup is NW waves that share K/V stages of one head; a wave owns QT tiles of 32
// queries, so each K/V fragment read from LDS feeds QT MFMAs.  The key stages of all blocks of a head
// form one sequence that is cut into W equal runs, one workgroup per (run, query group): a run that
// crosses a block boundary is processed as consecutive segments.  Every (segment, head, query) writes
// an un-normalised partial (O, l, m) plus the raw row max/min of q.k; merge_kernel turns the statistics
// into the mask bias and combines the partials.  A call with a single segment writes the normalised
// rows directly.
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
constexpr int kMaxRuns = 8 * kBlocks; // runs a head's key stages are cut into, at most
constexpr float kOverflowThr = 96.f;  // log2 units a logit may exceed the fixed reference by
constexpr float kOverflowSum = 1.2676506e30f;   // 2^100: a row sum beyond it (or not a number) means the same
#define SID_INF __builtin_huge_valf()

struct FlashParams {
    const u32x4* q[kBlocks];     // packed query used by each key block (hi part)
    const u32x4* q_lo[kBlocks];
    const u32x4* k[kBlocks];
    const u32x4* k_lo[kBlocks];
    const u32x4* v[kBlocks];
    const u32x4* v_lo[kBlocks];
    float c2[kBlocks];           // logit scale of each block, log2 domain: x = (q.k) * c2; already applied to
                                 // the packed query images, never zero (the host clamps |c2| >= 2^-40)
    int shift_cols;              // head_dim % 16 == 8: key images carry ones in columns d, d+1
    float* opart;                // [slot][BH][Nq][d] partial outputs, slot = blk * KS + segment: un-normalised f32, or
                                 // (part_f16) normalised by the row sum and stored as half -- a convex combination of
                                 // values, so half's range holds it and its 11 bits sit below the bf16 operand rounding
    int part_f16;
    float* stats;                // [slot][BH][Nq][4] = (m2, l, smax, smin); l == 0 marks an unused slot
    void* out;                   // (B, Nq, H*d), written here only when direct != 0
    int B, H, Nq, Nk, d;
    int NQT, NKT;                // 32-row tiles in the packed query / key images
    int nQG;                     // query groups (workgroups per (head, run))
    int KS;                      // partial slots per block (the most segments any block is cut into)
    int SPB;                     // stages per block
    int W;                       // runs per (head, query group) over the nblk * SPB key stages
    unsigned short bound[kMaxRuns + 1];   // run w covers stages [bound[w], bound[w + 1]) of that sequence
    unsigned char first_run[kBlocks];     // the run that holds a block's first stage
    unsigned char nseg[kBlocks];          // runs that overlap a block = partial slots it uses (<= KS)
    // SID_FLAG_BLOCK_SKIP (rowmap null: off; served by the SKIP kernel instances).  The queries of an image
    // are regrouped by mask side: position i of image b holds token rowmap[b * NP + i] (-1: padding; sides
    // start on workgroup boundaries, so no workgroup straddles two of them), rowmap_groups[b] query groups
    // are in use (nQG counts the launch's groups: up to two more than the tokens need).  A wave whose rows
    // are all padding takes part in the stage copies and barriers of its workgroup and computes nothing.  Every
    // workgroup bounds the logits of its rows from nsamp sampled key tiles per block (tiles j * samp_stride)
    // and drops what the bounds prove dead (margin dead_gap, see kDeadGapBase; ts = T * scale).  A group
    // with a dead style block runs the two remaining blocks as ONE sequence of 2 * SPB stages cut by
    // bound2 (first_run2 / nseg2 indexed by position in that sequence).
    // skip_count[0..3] = query groups with a dead style block / content segments reduced to their logits /
    // such segments redone in full / query groups
    const int* rowmap;
    const int* rowmap_groups;
    const float* mask;           // (B, Nq) signed mask, by token
    unsigned* skip_count;
    int NP;
    int nsamp, samp_stride;
    float ts, dead_gap;
    unsigned short bound2[kMaxRuns + 1];
    unsigned char first_run2[2], nseg2[2];
    // block skipping with the content block cached (nblk == 2): the statistics of the cached partial,
    // [BH][Nq][4]; entry 3 of a row is its smallest raw content logit, which the bounds then take exactly
    // instead of from samples.  A group with a dead style block runs the other one alone: bound2 cuts ONE block.
    const float* cstats;
    int nblk;                    // 3: injected [char | back | content], 2: the style blocks alone (the content
                                 // block comes from a cached partial), 1: single block
    int content_only;            // nblk == 1 and the block is a content block whose partial is kept: its smallest
                                 // raw q.k is the extreme the mask bias needs, as in the third block of an injected call
    int direct;                  // single part: normalise and write `out` here, no merge kernel
    int out_bf16;
#ifdef SID_DIAG
    unsigned long long* diag;    // diagnostic build: clock stamps of workgroup (blockIdx.x), 8 per workgroup
#endif
};

struct MergeParams {
    const float* opart;
    int part_f16;          // partial rows are normalised halves (see FlashParams)
    const float* stats;
    const float* mask;     // (B, Nq) signed mask, or null when nblk == 1
    void* out;             // (B, Nq, H*d)
    float scale;           // content logits: cc = (q_cnt.k_cnt) * scale
    float c2[kBlocks];
    int B, H, Nq, d, KS, nblk;
    int out_bf16;
    // cached content block (nblk == 2): one more part, [BH][Nq][d] rows and [BH][Nq][4] statistics in the
    // layout of one partial slot, kept outside the workspace; null otherwise
    const void* cpart;
    const float* cstats;
};

// Launch geometry chosen on the host (make_geometry in sid_attn.hip).
struct Geometry {
    int DKS, DVB;      // head_dim / 16 (rounded up), (head_dim + 1) / 32 (rounded up)
    int KVS;           // 32-key tiles per LDS stage
    int QT;            // 32-query tiles per wave
    int NW;            // waves per workgroup
    int KS;            // partial slots per block
    int SPB;           // stages per block
    int W, G;          // runs per (head, query group) over the nblk * SPB stages; stages of the longest run
    unsigned short bound[kMaxRuns + 1];
    unsigned char first_run[kBlocks], nseg[kBlocks];
    unsigned short bound2[kMaxRuns + 1];          // block skipping: the same W runs over two blocks
    unsigned char first_run2[2], nseg2[2];
    int NQT, NKT;      // tiles in the packed images
    int nQG;
    size_t lds;
    size_t q_units, k_units, v_units;   // u32x4 per packed tensor (one precision part)
};

// Shape policy shared by host and kernels.
constexpr size_t tile_bytes(int DKS, int DVB, bool x3) { return (size_t)(DKS + 2 * DVB) * 1024 * (x3 ? 2 : 1); }
// tiles per LDS stage: two, unless two double-buffered stages exceed what lets two workgroups share a CU
constexpr int stage_tiles(int DKS, int DVB, bool x3) { return 2 * 2 * tile_bytes(DKS, DVB, x3) <= 80 * 1024 ? 2 : 1; }
// query tiles per wave: two where the accumulators of both fit the 256-register budget
// (SID_WIDE_QT: measurement builds only -- 4 with SID_TUNE_NW=4 runs one wave per SIMD over four query
// tiles, 512 registers per lane; measured equal to the shipped two waves x two tiles at head_dim 40)
#ifndef SID_WIDE_QT
#define SID_WIDE_QT 2
#endif
constexpr int query_tiles(int DKS, int DVB, bool x3) { return (!x3 && DKS <= 3 && DVB <= 2) ? SID_WIDE_QT : 1; }

// Registers the software-pipelined loop keeps live (accumulators, two sets of logits and packed
// probabilities, Q, K and V fragments); the kernel uses it where they leave room in the 256-register
// budget of two waves per SIMD, and the plain stage loop elsewhere.
constexpr int pipeline_registers(int DKS, int DVB, int QT) {
    return 16 * DVB * QT + 32 + 16 + 4 * DKS * QT + 4 * DKS + (QT == 1 ? 2 : 1) * 8 * DVB;
}

// the software-pipelined loop is used by this instance
constexpr bool pipelined(int DKS, int DVB, bool x3) {
    return !x3 && stage_tiles(DKS, DVB, x3) % 2 == 0 && pipeline_registers(DKS, DVB, query_tiles(DKS, DVB, x3)) <= (SID_WIDE_QT > 2 ? 440 : 190);
}
// LDS stage buffers (a ring: stage s lives in buffer s % ring_stages, and a stage is requested
// ring_stages - 1 stages of compute before it is used).  Two: deeper rings were measured (three and
// four buffers at head_dim 40) and gain nothing -- the loop waits on LDS fragment reads and on issue
// slots, not on the copy -- while every extra buffer takes LDS from whatever shares the CU.
#ifndef SID_RING
#define SID_RING 2
#endif
constexpr int ring_stages(int DKS, int DVB, bool x3) { return SID_RING; }
// Block-skipping instances: the logits-only pass over a content block stages key tiles alone, through a
// deeper ring (it has little arithmetic per stage to hide the copy behind): up to 8 stages in up to 64 KiB.
constexpr int skip_ring(int DKS, int DVB, bool x3) {
    const int per_stage = stage_tiles(DKS, DVB, x3) * DKS * 1024;
    return 64 * 1024 / per_stage < 8 ? 64 * 1024 / per_stage : 8;
}
constexpr size_t skip_lds_bytes(int DKS, int DVB, bool x3) {
    const size_t normal = (size_t)ring_stages(DKS, DVB, x3) * stage_tiles(DKS, DVB, x3) * tile_bytes(DKS, DVB, x3);
    const size_t ring = (size_t)skip_ring(DKS, DVB, x3) * stage_tiles(DKS, DVB, x3) * DKS * 1024;
    return ring > normal ? ring : normal;
}

int launch_flash_bf16(const FlashParams& fp, const Geometry& g, hipStream_t stream);
int launch_flash_x3(const FlashParams& fp, const Geometry& g, hipStream_t stream);
int launch_flash_f16qk(const FlashParams& fp, const Geometry& g, hipStream_t stream);
int launch_merge(const MergeParams& mp, hipStream_t stream);

// Single-launch path for short key sequences (sid_attn_small.hip): reads the caller's tensors
// directly (element strides {batch, head, token}), needs no workspace.
bool small_attn_supported(int Nk, int d);
int launch_small_attn(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int d,
                      float c2, const long long qs[3], const long long ks[3], bool bf16io, bool x3, hipStream_t stream, bool shared_gpu = false,
                      bool qk16 = false);

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

// Logits reach the kernel already in the log2 domain: the pack kernel multiplies the query image by
// c2 = logit_scale * log2(e), so x = K.Q' needs no multiply.
struct RowState {
    float m2;      // reference the exponentials are taken against
    float smax;    // running max of x (this lane half's keys)
    float smin;    // running min (tracked only where needed)
};

// How a tile's exponentials are referenced:
//   kOnline : exact online softmax; m2 follows the running row max and O is rescaled every tile
//             (tail stages and the fallback pass)
//   kFixed  : m2 frozen, p = exp2(x - m2)
//   kShifted: m2 frozen AND already subtracted by the MFMA itself: two spare (zero-padding) columns of
//             the query fragments hold -m2 as a bf16 pair and the key image holds ones there, so the
//             accumulator is x - m2 and p = exp2(S) with no VALU arithmetic in front of the exponential
enum { kOnline = 0, kFixed = 1, kShifted = 2 };

// Block skipping (SID_FLAG_BLOCK_SKIP): a key block may be left out for a query row when its whole
// softmax mass is below half an ulp of the f32 row sum, i.e. when the largest biased logit it can hold
// lies more than kDeadGapBase + log2(keys of the block) log2 units under a logit the row is known to hold.
constexpr float kDeadGapBase = 26.f;
constexpr int kProbeTiles = 4;        // key tiles (of 32 keys) sampled per block for the bounds
constexpr int kSkipPadGroups = 2;     // query groups a launch adds for the padding between the mask sides

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

// Fragments of one tile read from an LDS image: NF units of 16 bytes, 64 units apart.
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

// One MFMA of the logits product.  QK16: the query and key images hold IEEE half values (11 significant
// bits against the 8 of bf16; same MFMA rate); the registers are typed bf16x8 either way, only the bits
// differ.  The P.V product always takes bf16 operands: the probabilities need bf16's exponent range.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool QK16>
__device__ __forceinline__ f32x16 mfma_qk(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (QK16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// S^T = K Q^T for one tile of 32 keys x 32 queries.
template <int DKS, bool X3, bool QK16>
__device__ __forceinline__ f32x16 qk_tile(const Frags<DKS, X3>& kf, const Frags<DKS, X3>& qf) {
    static_assert(!(X3 && QK16), "the split-operand mode keeps bf16 parts");
    f32x16 S;
#pragma unroll
    for (int i = 0; i < 16; ++i) S[i] = 0.f;
#pragma unroll
    for (int s = 0; s < DKS; ++s) {
        S = mfma_qk<QK16>(kf.hi[s], qf.hi[s], S);
        if (X3) {
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.hi[s], qf.lo[X3 ? s : 0], S, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.lo[X3 ? s : 0], qf.hi[s], S, 0, 0, 0);
        }
    }
    return S;
}

// Softmax step for one 32x32 tile whose logits S are ready, then O^T += V^T P^T.
//   TAIL: keys >= Nk are masked (only with kOnline).
// vf holds the V^T fragments of the tile: index s2 * DVB + b (k-step s2, dv block b).
template <int DVB, bool X3, bool TAIL, bool TRACK_MIN, int MODE>
__device__ __forceinline__ void softmax_pv_tile(const f32x16& S, const Frags<2 * DVB, X3>& vf, int hh, int key0,
                                                int Nk, f32x16 (&O)[DVB], RowState& rs) {
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
    // the statistics stay in unshifted coordinates
    rs.smax = fmaxf(rs.smax, MODE == kShifted ? tmax + rs.m2 : tmax);
    if (TRACK_MIN || TAIL) rs.smin = fminf(rs.smin, MODE == kShifted ? tmin + rs.m2 : tmin);

    if (MODE == kOnline) {
        const float full = fmaxf(tmax, __shfl_xor(tmax, 32));       // -inf where neither half has a valid key
        const float mnew = fmaxf(rs.m2, full);
        const float alpha = (mnew == -SID_INF) ? 1.f : __builtin_amdgcn_exp2f(rs.m2 - mnew);
#pragma unroll
        for (int b = 0; b < DVB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[b][i] *= alpha;
        rs.m2 = mnew;
    }
    const float m2 = rs.m2;

    float pv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pv[i] = __builtin_amdgcn_exp2f(MODE == kShifted ? S[i] : S[i] - m2);
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

// One LDS stage of KVS key tiles for the wave's QT query tiles.  Per key tile the K and V fragments
// are read from LDS once and feed the MFMAs of all QT query tiles; the QT independent softmax
// chains give the scheduler VALU work to run beside the matrix pipe.
template <int DKS, int DVB, int KVS, bool X3, int QT, bool TAIL, bool TRACK_MIN, int MODE, bool QK16>
__device__ __forceinline__ void process_stage(const u32x4* L, int lane, int hh, int key0, int Nk,
                                              const Frags<DKS, X3> (&qf)[QT], f32x16 (&O)[QT][DVB],
                                              RowState (&rs)[QT]) {
    constexpr int KU = KVS * DKS * 64;
    constexpr int PU = KU + KVS * 2 * DVB * 64;
#pragma unroll
    for (int j = 0; j < KVS; ++j) {
        if (TAIL && key0 + 32 * j >= Nk) break;
        Frags<DKS, X3> kf;
        Frags<2 * DVB, X3> vf;
        kf.load(L + (j * DKS) * 64 + lane, PU);
        vf.load(L + KU + (j * 2 * DVB) * 64 + lane, PU);
        f32x16 S[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) S[t] = qk_tile<DKS, X3, QK16>(kf, qf[t]);
#pragma unroll
        for (int t = 0; t < QT; ++t)
            softmax_pv_tile<DVB, X3, TAIL, TRACK_MIN, MODE>(S[t], vf, hh, key0 + 32 * j, Nk, O[t], rs[t]);
        // keep the tiles of a stage apart in the schedule: the QT query tiles already give the matrix
        // and vector pipes independent work, and overlapping two key tiles as well costs registers
        if (QT > 1 && j + 1 < KVS) __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- Software-pipelined steady state ------------------------------------------------------------
// On gfx950 a 32x32x16 MFMA occupies the matrix pipe for 32 cycles but the wave's issue port for only
// about 8; v_exp_f32 takes about 7 issue cycles, the other vector ops about 4.  A wave issues in order,
// so matrix and vector work only overlap when they alternate in the instruction stream.  The pipelined
// stage therefore runs three units (a unit = one query tile x one key tile) at once:
//
//      slot u:   vector ops of softmax(u)   interleaved with   MFMAs of  P.V(u-1)  and  K.Q'(u+1)
//
// one MFMA followed by an equal share of the unit's vector ops, with a scheduling fence after each
// group so the compiler keeps the interleave.  The pipeline runs across stage boundaries: the logits
// of the next stage's first unit are computed in the last slot of a stage (after the barrier that
// publishes that stage), and the P.V of a stage's last unit runs in the first slot of the next.
enum { kOpExp, kOpCvt, kOpMax, kOpMin };
struct SoftmaxOps { int n; int kind[64]; int arg[64]; };
// One instruction each, whatever the compiler knows about the operands: written with fmaxf / fminf the
// tree on MFMA results gets canonicalising v_max x, x in front (the results might be signalling NaNs for all
// the compiler knows), and every extra vector instruction of the steady loop costs the issue port four cycles.
__device__ __forceinline__ float max3_f32(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float min3_f32(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// The vector ops of one unit in issue order: per pair of logits two exponentials, a step of the running max
// OR min (one three-operand instruction that folds the pair into the segment's running extreme, which
// lives in the accumulator's own coordinates until the loop ends: 32 vector instructions per unit, none
// besides the exponentials, the extreme and the packing), and the bf16 packing of the previous pair.
// The pipelined loop tracks exactly one
// extreme of the logits per segment, the one the mask bias needs from it (the largest raw q.k of a
// style block, the smallest content logit): overflow is caught from the row sums instead of from the
// row maximum, and the merge takes its reference from the sums as well.
constexpr SoftmaxOps softmax_ops(bool track_min) {
    SoftmaxOps t{};
    int n = 0;
    for (int g = 0; g < 8; ++g) {
        t.kind[n] = kOpExp; t.arg[n++] = 2 * g;
        t.kind[n] = kOpExp; t.arg[n++] = 2 * g + 1;
        t.kind[n] = track_min ? kOpMin : kOpMax; t.arg[n++] = g;
        if (g > 0) { t.kind[n] = kOpCvt; t.arg[n++] = g - 1; }
    }
    t.kind[n] = kOpCvt; t.arg[n++] = 7;
    t.n = n;
    return t;
}

template <int DKS, int DVB, int KVS, int QT, bool TRACK_MIN, int MODE, bool QK16>
struct StagePipe {
    static_assert(KVS % 2 == 0, "fragment registers alternate by key tile parity");
    static_assert(MODE == kFixed || MODE == kShifted, "the reference is frozen in the pipelined loop");
    static constexpr int U = KVS * QT;            // units per stage
    static constexpr int NM = DKS + 2 * DVB;      // MFMAs per slot
    static constexpr int KU = KVS * DKS * 64;     // u32x4 offset of the V image in a stage
    static constexpr SoftmaxOps T = softmax_ops(TRACK_MIN);
    using QF = Frags<DKS, false>;

    f32x16 S[2];                       // logits: unit parity
    bf16x8 Pk[2][2];                   // packed probabilities: unit parity, k-step
    Frags<DKS, false> kf;              // K fragments of the tile the next logits are taken against
    static constexpr int NVS = QT == 1 ? 2 : 1;
    Frags<2 * DVB, false> vf[NVS];     // V^T fragments of the tile(s) of the running P.V
    float e[16];
    float ext[QT];                     // running extreme of every query tile's logits, as the MFMA leaves them
                                       // (with the reference already subtracted in the shifted mode)

    template <int I>
    __device__ __forceinline__ void op(const f32x16& Sx, bf16x8 (&Px)[2], RowState& r, float& x) {
        constexpr int kind = T.kind[I], a = T.arg[I];
        if constexpr (kind == kOpExp) {
            e[a] = __builtin_amdgcn_exp2f(MODE == kShifted ? Sx[a] : Sx[a] - r.m2);
        } else if constexpr (kind == kOpCvt) {
            Px[a / 4][2 * (a % 4)] = (__bf16)e[2 * a];
            Px[a / 4][2 * (a % 4) + 1] = (__bf16)e[2 * a + 1];
        } else if constexpr (kind == kOpMax) {
            x = max3_f32(x, Sx[2 * a], Sx[2 * a + 1]);
        } else {
            x = min3_f32(x, Sx[2 * a], Sx[2 * a + 1]);
        }
    }

    // After the loop: the running extremes go back to the row state, in unshifted coordinates.
    __device__ __forceinline__ void fold(RowState (&rs)[QT]) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const float x = MODE == kShifted ? ext[t] + rs[t].m2 : ext[t];
            if (TRACK_MIN) rs[t].smin = fminf(rs[t].smin, x);
            else rs[t].smax = fmaxf(rs[t].smax, x);
        }
    }

    __device__ __forceinline__ void load_k(const u32x4* L, int tile, int lane) { kf.load(L + (tile * DKS) * 64 + lane, 0); }
    // one k-step half (DVB fragments) of a V tile
    template <int SET>
    __device__ __forceinline__ void load_v(const u32x4* L, int tile, int half, int lane) {
#pragma unroll
        for (int b = 0; b < DVB; ++b)
            vf[SET].hi[half * DVB + b] = as_bf16x8(L[KU + (tile * 2 * DVB + half * DVB + b) * 64 + lane]);
    }

    // Before the first slot: the logits of the stage's first unit; a zero P.V stands in for unit -1.
    __device__ __forceinline__ void begin(const u32x4* L, int lane, const QF (&qf)[QT]) {
#pragma unroll
        for (int t = 0; t < QT; ++t) ext[t] = TRACK_MIN ? SID_INF : -SID_INF;
        load_k(L, 0, lane);
#pragma unroll
        for (int i = 0; i < 2 * DVB; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) vf[NVS - 1].hi[i][c] = (__bf16)0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) { Pk[1][0][c] = (__bf16)0.f; Pk[1][1][c] = (__bf16)0.f; }
        S[0] = qk_tile<DKS, false, QK16>(kf, qf[0]);
        if constexpr (QT == 1) {
            load_k(L, 1, lane);
            load_v<0>(L, 0, 0, lane);
            load_v<0>(L, 0, 1, lane);
        }
    }

    // Slot u of the stage at L; Lnext is the following stage (read only in the last slot).
    //
    // Fragment registers are reloaded as soon as the last MFMA that reads them has been issued.  The
    // MFMAs of a slot run  P.V k-step 0 | K.Q' chain | P.V k-step 1, so
    //   * the k-step-0 half of a V set is free after the first DVB MFMAs and the k-step-1 half after
    //     the last, and the halves of a later tile are read then.  With QT > 1 there is one set and
    //     the reload happens in the slot of a tile's FIRST unit; with QT == 1 every slot ends a tile, so
    //     there are two sets (tile parity) and slot u fetches tile u+1 into the set tile u-1 leaves.
    //     Either way all V reads of a stage precede the barrier, after which its buffer is overwritten
    //     (with QT == 1 the last slot reads the first V tile of the NEXT stage);
    //   * K is free after the chain; the next tile's K is read then, one slot ahead of its use, except
    //     the next stage's first tile, which is only visible in the last slot and is read at its start
    //     (the chain comes DVB MFMAs later).
    template <int u>
    __device__ __forceinline__ void slot(const u32x4* L, const u32x4* Lnext, int lane, const QF (&qf)[QT],
                                         f32x16 (&O)[QT][DVB], RowState (&rs)[QT]) {
        constexpr int t = u % QT;
        constexpr int tn = (u + 1) % QT;                       // query tile of the unit whose logits are produced
        constexpr int tp = (u + U - 1) % QT;                   // ... of the unit whose P.V is produced
        constexpr int sc = u & 1, sn = (u + 1) & 1;
        constexpr int vs = QT == 1 ? (u + 1) & 1 : 0;          // V set read by this slot's P.V
        auto reload_v = [&](int half) {
            if constexpr (QT > 1) {
                if constexpr (u % QT == 0) load_v<0>(L, u / QT, half, lane);
            } else if constexpr (u + 1 < U) {
                load_v<vs>(L, u + 1, half, lane);
            } else {
                load_v<vs>(Lnext, 0, half, lane);
            }
        };
        if constexpr (u == U - 1) load_k(Lnext, 0, lane);
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            if constexpr (m < DVB) {
                O[tp][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[vs].hi[m], Pk[sn][0], O[tp][m], 0, 0, 0);
                if constexpr (m == DVB - 1) reload_v(0);
            } else if constexpr (m < DVB + DKS) {
                constexpr int s = m - DVB;
                if constexpr (s == 0) {
                    f32x16 z;
#pragma unroll
                    for (int i = 0; i < 16; ++i) z[i] = 0.f;
                    S[sn] = mfma_qk<QK16>(kf.hi[0], qf[tn].hi[0], z);
                } else {
                    S[sn] = mfma_qk<QK16>(kf.hi[s], qf[tn].hi[s], S[sn]);
                }
                if constexpr (s == DKS - 1 && (u + 2) % QT == 0) {
                    if constexpr (u + 2 < U) load_k(L, (u + 2) / QT, lane);
                    else if constexpr (u == U - 1) load_k(Lnext, (u + 2 - U) / QT, lane);   // QT == 1 only
                }
            } else {
                constexpr int b = m - DVB - DKS;
                O[tp][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[vs].hi[DVB + b], Pk[sn][1], O[tp][b], 0, 0, 0);
                if constexpr (b == DVB - 1) reload_v(1);
            }
            static_for<m * T.n / NM, (m + 1) * T.n / NM>([&](auto ic) { op<decltype(ic)::value>(S[sc], Pk[sc], rs[t], ext[t]); });
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // After the last slot: the P.V of the last unit.
    __device__ __forceinline__ void finish(f32x16 (&O)[QT][DVB]) {
#pragma unroll
        for (int b = 0; b < DVB; ++b) {
            O[QT - 1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[NVS - 1].hi[b], Pk[1][0], O[QT - 1][b], 0, 0, 0);
            O[QT - 1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[NVS - 1].hi[DVB + b], Pk[1][1], O[QT - 1][b], 0, 0, 0);
        }
    }
};

// Direct global->LDS load of one 1 KiB unit: lane l's 16 bytes at src + l land at dst + l.
// Written as asm on purpose: the compiler then neither counts the load nor orders LDS reads behind
// it (with the builtin it waits vmcnt(0) before the next ds_read, which serialises the prefetch
// with the compute it is meant to overlap).  The kernel waits for these loads itself
// (wait_direct_loads) before the barrier that publishes a stage.
__device__ __forceinline__ void glds16(const u32x4* src_uniform, unsigned lane_bytes, unsigned lds_addr) {
    // m0 carries the LDS address of the load; it is declared clobbered (nothing else in these kernels
    // uses it), which saves a save/restore pair of scalar instructions per load
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_bytes), "s"(src_uniform), "s"(lds_addr)
                 : "memory", "m0");
}
__device__ __forceinline__ void wait_direct_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// SHIFT: the kShifted steady state (head_dim % 16 == 8); a separate instance rather than a run-time
// branch, because a kernel that carries both steady-state loops no longer fits its registers.
// QK16: half-precision (f16) query and key images (SID_COMPUTE_F16QK); see mfma_qk.
// SKIP: block skipping (SID_FLAG_BLOCK_SKIP); instances of their own, so that calls without the flag run
// exactly the code they ran before it existed.
template <int DKS, int DVB, int KVS, bool X3, int QT, int NW, bool SHIFT, bool QK16, bool SKIP = false>
__global__ void __launch_bounds__(NW * 64, (NW == 8 || QT > 2) ? 1 : 2) flash_kernel(FlashParams P) {
    static_assert(!(SKIP && X3), "block skipping reads the f32 statistics of the bf16 / f16 kernels");
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    constexpr int KU = KVS * DKS * 64;        // u32x4 per stage: K image
    constexpr int VU = KVS * 2 * DVB * 64;    //                  V image
    constexpr int PU = KU + VU;               // one precision part
    constexpr int SU = PU * (X3 ? 2 : 1);     // one stage: [K hi | V hi | K lo | V lo]
    constexpr int KUn = KU / 64, VUn = VU / 64;
    constexpr int NB = ring_stages(DKS, DVB, X3);                                    // stage buffers; stage s lives in s % NB
    constexpr int NL = ((KUn + NW - 1) / NW + (VUn + NW - 1) / NW) * (X3 ? 2 : 1);  // direct loads per wave and stage

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5;
#ifdef SID_DIAG
    // 8 stamps per segment (up to 3 segments per workgroup): 0..5 from the constant 100 MHz counter,
    // 6, 7 = the shader cycle counter at stamps 2 and 3 (the steady loop)
    int diag_seg = 0;
    auto stamp = [&](int i) {
        if (tid == 0 && P.diag && diag_seg < 3) {
            unsigned long long* dg = P.diag + ((size_t)blockIdx.x * 3 + diag_seg) * 8;
            dg[i] = __builtin_amdgcn_s_memrealtime();
            if (i == 2 || i == 3) dg[4 + i] = __builtin_readcyclecounter();
        }
    };
    stamp(0);
#else
    auto stamp = [&](int) {};
#endif
    const int BH = P.B * P.H;

    // workgroup -> ((head, run), query group); workgroups that stream the same K/V images get ids
    // equal mod 8 when the group count allows, i.e. they share an XCD and its L2 (speed only)
    const int groups = BH * P.W;
    int grp, qg;
    if constexpr (SKIP) {
        // run-major: the runs over the style blocks, which are never shortened, start before the runs over
        // the content block; heads alternate fastest (the workgroups of a head share an XCD when BH % 8 == 0)
        const int id = blockIdx.x, per_run = BH * P.nQG;
        const int rest = id % per_run;
        grp = (rest % BH) * P.W + id / per_run;
        qg = rest / BH;
    } else {
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
    // The key stages of the nblk blocks form one sequence of nblk * SPB stages; run w of W covers
    // stages [bound[w], bound[w + 1]) of it (the host balances the runs, counting what a change of
    // block costs).  Where a run crosses a block boundary the workgroup finishes one segment (its own
    // partial output and statistics) and starts the next.
    const int run_w = grp % P.W, bh = grp / P.W;

    // this wave's query tiles; surplus tiles redo the last one and store nothing
    const int qt0 = (qg * NW + wave) * QT;
    // SKIP: the token behind each of this lane's query rows (-1: padding)
    int tok[SKIP ? QT : 1];
    if constexpr (SKIP) {
        const int b_ = bh / P.H;
        if (qg >= P.rowmap_groups[b_]) return;            // the whole workgroup: a padding group of the launch
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            // (anything that is not a token of the image counts as padding: a map that a measurement call with
            // a stage flag finds overwritten by another call must not turn into addresses)
            const int tk = P.rowmap[(size_t)b_ * P.NP + (qt0 + t) * 32 + (lane & 31)];
            tok[t] = (unsigned)tk < (unsigned)P.Nq ? tk : -1;
        }
    }
    // a wave of padding alone (the tail of a mask side): no arithmetic, only its share of copies and barriers
    bool wave_live = true;
    if constexpr (SKIP) {
        bool any_row = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) any_row |= tok[t] >= 0;
        wave_live = __builtin_amdgcn_readfirstlane((int)(__ballot(any_row) != 0ull)) != 0;
    }
    // token of this lane's row of query tile t, -1 where there is nothing to store
    auto row_token = [&](int t) {
        if constexpr (SKIP) {
            return tok[t];
        } else {
            const int qt = qt0 + t;
            const int q = qt * 32 + (lane & 31);
            return (qt >= P.NQT || q >= P.Nq) ? -1 : q;
        }
    };
    // this lane's part of the fragments of query tile t in a packed query image
    auto q_frag_ptr = [&](const u32x4* image, int t) {
        if constexpr (SKIP) {
            const int tc = max(tok[t], 0);                // padding rows compute on token 0 and store nothing
            return image + ((size_t)(bh * P.NQT + (tc >> 5)) * DKS) * 64 + hh * 32 + (tc & 31);
        } else {
            return image + ((size_t)(bh * P.NQT + min(qt0 + t, P.NQT - 1)) * DKS) * 64 + lane;
        }
    };
    Frags<DKS, X3> qf[QT];

    // ---- block skipping: bounds from sampled keys, one decision for the whole workgroup.  In log2 units,
    // with x the packed-image logits (style: T*s*log2e * q.k, content: s*log2e * q_cnt.k_cnt), ts = T*s:
    //     style block b, key j :  x_bj + (min_j xc - max_j x_b / ts) * (m * ch_b),   ch_1 = -1, ch_2 = +1
    //     content, key j       :  xc_j
    // For a row with m > ts > 0 (mirrored for m < -ts), from ANY sampled keys:
    //     the row holds a logit >= LB  = x1 + (x1 / ts - xc) * m         (x1: a block-1 sample, xc: a content sample)
    //     every logit of block 2 <= UB2 = x2 * (1 - m / ts) + m * xc      (x2: a block-2 sample)
    // because the true max_j x_b is at least any sample and the true min_j xc at most any sample, and both
    // bounds are monotone in those.  Block 2 is dead for the row when LB - UB2 >= dead_gap.  The content
    // block's maximum is not bounded by samples: it is only PREDICTED dead here and verified below with
    // the exact maximum of its segment.  The workgroups of one query group read the same samples with the
    // same instructions, so they agree.
    int dead = -1;
    unsigned gflags = 0;
    float row_lb[SKIP ? QT : 1];
    if constexpr (SKIP) {
        // (content block cached: no content images in the call; min_j xc comes from the partial's statistics
        // and the content block is never a candidate.  Two instantiations of one probe behind a uniform branch,
        // so that each is the straight-line code it would be alone)
        const bool with_content = P.nblk == kBlocks;
        float xs[kBlocks][QT], cmn[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) cmn[t] = SID_INF;
#pragma unroll
        for (int b = 0; b < kBlocks; ++b)
#pragma unroll
            for (int t = 0; t < QT; ++t) xs[b][t] = -SID_INF;
        auto probe = [&](auto wc) {
            constexpr bool WC = decltype(wc)::value;
            Frags<DKS, false> qb[QT], qc[WC ? QT : 1];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                qb[t].load(q_frag_ptr(P.q[0], t), 0);
                if constexpr (WC) qc[t].load(q_frag_ptr(P.q[kBlocks - 1], t), 0);
            }
            if constexpr (!WC) {
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    if (tok[t] >= 0) cmn[t] = P.cstats[((size_t)bh * P.Nq + tok[t]) * 4 + 3] * P.c2[kBlocks - 1];
            }
#pragma unroll
            for (int b = 0; b < (WC ? kBlocks : 2); ++b) {
                // a fixed trip count with a clamped index (a repeated tile changes no extreme): the loads do not
                // depend on each other, so as many are in flight as registers allow
#pragma unroll
                for (int j = 0; j < kProbeTiles; ++j) {
                    Frags<DKS, false> kf;
                    kf.load(P.k[b] + ((size_t)(bh * P.NKT + min(j, P.nsamp - 1) * P.samp_stride) * DKS) * 64 + lane, 0);
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        const f32x16 S = qk_tile<DKS, false, QK16>(kf, (WC && b == kBlocks - 1) ? qc[WC ? t : 0] : qb[t]);
                        xs[b][t] = fmaxf(xs[b][t], max16(S));
                        if (WC && b == kBlocks - 1) cmn[t] = fminf(cmn[t], min16(S));
                    }
                }
            }
        };
        if (with_content) probe(std::true_type{}); else probe(std::false_type{});
        unsigned bad = 0;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float x[kBlocks];
#pragma unroll
            for (int b = 0; b < kBlocks; ++b) x[b] = fmaxf(xs[b][t], __shfl_xor(xs[b][t], 32));
            const float cmin = fminf(cmn[t], __shfl_xor(cmn[t], 32)), cmax = x[kBlocks - 1];
            row_lb[t] = -SID_INF;
            if (tok[t] >= 0) {                            // padding rows agree with everything
                const float m = P.mask[(size_t)(bh / P.H) * P.Nq + tok[t]];
                const float am = fabsf(m);
                bool d1 = false, d2 = false, dc = false;
                if (am > P.ts) {
                    const float xl = m > 0.f ? x[0] : x[1], xd = m > 0.f ? x[1] : x[0];
                    const float lbv = xl + (xl / P.ts - cmin) * am;
                    const float ub = xd * (1.f - am / P.ts) + am * cmin;
                    const bool far_dead = lbv - ub >= P.dead_gap;
                    d1 = far_dead && m < 0.f;
                    d2 = far_dead && m > 0.f;
                    // a guess with half the sampled range of headroom
                    dc = with_content && lbv - (cmax + 0.5f * (cmax - cmin)) >= P.dead_gap;
                    row_lb[t] = lbv;
                }
                bad |= (d1 ? 0u : 1u) | (d2 ? 0u : 2u) | (dc ? 0u : 4u);
            }
        }
        gflags = 0;                                       // (the workgroup votes are boolean: one per bit)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            gflags |= __builtin_amdgcn_readfirstlane(__syncthreads_and((int)((bad >> k) & 1u) == 0)) ? 1u << k : 0u;
        if ((gflags & 3u) == 3u) gflags &= ~3u;           // (only a group of padding alone could say so)
        dead = (gflags & 3u) == 1u ? 0 : ((gflags & 3u) == 2u ? 1 : -1);
    }
    const int g0 = dead >= 0 ? P.bound2[run_w] : P.bound[run_w];
    const int g1 = dead >= 0 ? P.bound2[run_w + 1] : P.bound[run_w + 1];
    auto real_block = [&](int tb) { return dead < 0 ? tb : (tb == 0 ? 1 - dead : kBlocks - 1); };

    // per segment: block, first stage and stage count within the block, partial slot, scale, images
    int blk = 0, st0 = 0, slen = 0, part = 0;
    float c2 = 1.f;
    bool need_min = false;
    const u32x4 *gk = nullptr, *gv = nullptr, *gkl = nullptr, *gvl = nullptr;   // wave-uniform; lane offset added by the load
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)lds));
    const unsigned lane_bytes = lane * 16;

    // Stage copy: every 1 KiB unit is one wave-wide direct load; the NW waves take the units of each
    // image round-robin.  Branch-free (the pipelined loop must stay one basic block): where the unit
    // count is not a multiple of NW, the surplus waves copy the image's last unit once more.
    auto issue_stage = [&](int st, int buf) {
        const unsigned dk = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * SU) * 16);   // uniform by construction
        static_for<0, (KUn + NW - 1) / NW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int un = (KUn % NW == 0 || (i + 1) * NW <= KUn) ? wave + i * NW : min(wave + i * NW, KUn - 1);
            glds16(gk + (size_t)st * KU + un * 64, lane_bytes, dk + un * 1024);
            if (X3) glds16(gkl + (size_t)st * KU + un * 64, lane_bytes, dk + (PU + un * 64) * 16);
        });
        static_for<0, (VUn + NW - 1) / NW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int un = (VUn % NW == 0 || (i + 1) * NW <= VUn) ? wave + i * NW : min(wave + i * NW, VUn - 1);
            glds16(gv + (size_t)st * VU + un * 64, lane_bytes, dk + (KU + un * 64) * 16);
            if (X3) glds16(gvl + (size_t)st * VU + un * 64, lane_bytes, dk + (PU + KU + un * 64) * 16);
        });
    };

    f32x16 O[QT][DVB];
    RowState rs[QT];
    __shared__ int redo;

    // Stages [it0, it1) of this part with ONE variant of the stage body per loop, so that every loop
    // is straight-line code.
    // Stages of the segment whose copy has been issued so far (the prologue starts NB at once and the
    // pipelined loop runs NB - 1 stages in front, so the plain loop below must not issue those again).
    int nissued = 0;
    auto run = [&](auto tail_c, auto min_c, auto mode_c, int it0, int it1) {
        constexpr bool TAIL = decltype(tail_c)::value, TMIN = decltype(min_c)::value;
        constexpr int MODE = decltype(mode_c)::value;
        for (int it = it0; it < it1; ++it) {
            // Nothing the compiler counts may be outstanding once the direct loads are in flight: its
            // counted vmcnt waits do not know about them and would wait for the oldest one in every
            // iteration.  At this point nothing is in flight as a rule (the previous iteration
            // drained it), so this wait is free; it only resets the compiler's bookkeeping.
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            if (it + 1 < slen && nissued <= it + 1) {
                issue_stage(st0 + it + 1, (it + 1) % NB);
                nissued = it + 2;
            }
            if (wave_live)
                process_stage<DKS, DVB, KVS, X3, QT, TAIL, TMIN, MODE, QK16>(lds + (it % NB) * SU, lane, hh,
                                                                       (st0 + it) * KVS * 32, P.Nk, qf, O, rs);
            wait_direct_loads();
            __syncthreads();          // stage it+1 has landed in every wave's view; stage it is free
        }
    };
    // The same stages through the software pipeline (StagePipe), with the NB stage buffers as a ring.
    // On entry stage it0 is visible; stages up to it0 + NB - 1 are issued here unless they already are.
    // In iteration `it` the barrier sits before the last slot: by then every wave has read all fragments
    // of stage `it` into registers, so its buffer can take stage it + NB, and stage it + 1 must be
    // visible because the last slot starts on it.  Loads complete in order and every iteration issues
    // the same number, so "all but the youngest NB - 2 stages have landed" is a constant vmcnt.
    constexpr bool kPipelined = pipelined(DKS, DVB, X3);
    auto run_pipe = [&](auto min_c, auto mode_c, int it0, int it1) {
        if constexpr (kPipelined) {
            constexpr int U = KVS * QT;
            StagePipe<DKS, DVB, KVS, QT, decltype(min_c)::value, decltype(mode_c)::value, QK16> pipe;
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see `run`
            for (int k = 1; k < NB; ++k)
                if (nissued <= it0 + k && it0 + k < slen) { issue_stage(st0 + it0 + k, (it0 + k) % NB); nissued = it0 + k + 1; }
            wait_direct_loads();                  // (nothing to wait for after the prologue)
            __syncthreads();
            if (wave_live) pipe.begin(lds + (it0 % NB) * SU, lane, qf);
            // The copies issued inside the loop go through running pointers (one per unit this wave
            // copies): a handful of scalar adds per stage instead of the full address arithmetic of
            // issue_stage -- scalar instructions cost issue slots the loop does not have to spare.
            constexpr int NKL = (KUn + NW - 1) / NW, NVL = (VUn + NW - 1) / NW;
            const u32x4* rp[NKL + NVL];
            unsigned ro[NKL + NVL];           // byte offset of the unit within a stage buffer
            {
                const int nx = min(it0 + NB, slen - 1);
#pragma unroll
                for (int i = 0; i < NKL; ++i) {
                    const int un = min(wave + i * NW, KUn - 1);
                    rp[i] = gk + (size_t)(st0 + nx) * KU + un * 64;
                    ro[i] = un * 1024;
                }
#pragma unroll
                for (int i = 0; i < NVL; ++i) {
                    const int un = min(wave + i * NW, VUn - 1);
                    rp[NKL + i] = gv + (size_t)(st0 + nx) * VU + un * 64;
                    ro[NKL + i] = (KU + un * 64) * 16;
                }
            }
            // One stage; `buf` is the ring buffer it lives in (it % NB).
            auto stage = [&](int it, int buf) {
                const u32x4* L = lds + buf * SU;
                const u32x4* Ln = lds + ((buf + 1) % NB) * SU;
                if (wave_live)
                    static_for<0, U - 1>([&](auto uc) { pipe.template slot<decltype(uc)::value>(L, Ln, lane, qf, O, rs); });
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * NL) : "memory");
                __syncthreads();
                // unconditional (no branch in the loop body): past the end the last stage is copied again
                // into the buffer that has just become free, which nothing reads
                {
                    const unsigned bb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * SU) * 16);
                    const bool more = it + 1 + NB < slen;
                    const int stepk = more ? KU : 0, stepv = more ? VU : 0;
#pragma unroll
                    for (int i = 0; i < NKL + NVL; ++i) {
                        glds16(rp[i], lane_bytes, bb + ro[i]);
                        rp[i] += i < NKL ? stepk : stepv;
                    }
                }
                if (wave_live) pipe.template slot<U - 1>(L, Ln, lane, qf, O, rs);
            };
#ifndef SID_NO_UNROLL2
            if constexpr (NB == 2) {
                // two stages per trip: the ring buffer of every fragment read and stage copy is then a constant
                // (an immediate offset) instead of scalar arithmetic on the stage's parity, which costs issue
                // slots of the in-order stream in every stage
                int it = it0;
                if (it < it1 && (it & 1)) { stage(it, 1); ++it; }
                for (; it + 1 < it1; it += 2) {
                    stage(it, 0);
                    stage(it + 1, 1);
                }
                if (it < it1) stage(it, 0);
            } else
#endif
            {
                for (int it = it0; it < it1; ++it) stage(it, it % NB);
            }
            if (wave_live) { pipe.finish(O); pipe.fold(rs); }
            nissued = max(nissued, min(slen, it1 + NB));
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using Online = std::integral_constant<int, kOnline>;
    using Fixed = std::integral_constant<int, kFixed>;
    using Shifted = std::integral_constant<int, kShifted>;

    // The spare columns d, d+1 of the query fragments (zero padding when head_dim % 16 == 8; the key
    // image holds ones there): dword 0 of the last k-step in the upper lane half.
    auto set_shift = [&](int t, unsigned pair) {
        u32x4 w = __builtin_bit_cast(u32x4, qf[t].hi[DKS - 1]);
        w[0] = hh ? pair : w[0];
        qf[t].hi[DKS - 1] = as_bf16x8(w);
    };

    // images and query fragments of a block
    auto enter_block = [&](int b_) {
        gk = P.k[b_] + (size_t)bh * P.NKT * DKS * 64;
        gv = P.v[b_] + (size_t)bh * P.NKT * 2 * DVB * 64;
        gkl = X3 ? P.k_lo[b_] + (size_t)bh * P.NKT * DKS * 64 : nullptr;
        gvl = X3 ? P.v_lo[b_] + (size_t)bh * P.NKT * 2 * DVB * 64 : nullptr;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            qf[t].load(q_frag_ptr(P.q[b_], t), X3 ? (int)(P.q_lo[b_] - P.q[b_]) : 0);
        }
    };
    // A segment that is followed by another one requests that segment's first stages and query
    // fragments before it stores its own results, so their latency passes behind the store.
    int prefetched = 0;               // stages of the segment about to start that are already issued

    unsigned skipc = 0;               // block-skipping counters of this workgroup, one per byte (<= 3 segments)
    if constexpr (SKIP) {
        if (run_w == 0) skipc = (1u << 24) | (dead >= 0 ? 1u : 0u);
        if (dead >= 0 && run_w == 0) {
            // nobody computes the dead block: its partial slots are marked empty here
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int q = row_token(t);
                if (q < 0 || hh != 0) continue;
                for (int pp = 0; pp < P.KS; ++pp) {
                    f32x4 none = {-SID_INF, 0.f, -SID_INF, SID_INF};
                    *(f32x4*)(P.stats + ((((size_t)dead * P.KS + pp) * BH + bh) * P.Nq + q) * 4) = none;
                }
            }
        }
    }

    for (int tb = g0 / P.SPB; tb * P.SPB < g1; ++tb) {
    blk = real_block(tb);
    st0 = max(g0 - tb * P.SPB, 0);
    slen = min(g1 - tb * P.SPB, P.SPB) - st0;
    const int first_run = dead >= 0 ? P.first_run2[tb] : P.first_run[tb];   // the run that holds the block's first stage
    const int nseg_blk = dead >= 0 ? P.nseg2[tb] : P.nseg[tb];
    part = blk * P.KS + (run_w - first_run);
    if (!prefetched) enter_block(blk);
    c2 = P.c2[blk];
    need_min = (P.nblk == kBlocks && blk == kBlocks - 1) || P.content_only || c2 < 0.f;

    // ---- block skipping, content block predicted dead for the whole query group: only the extremes of
    // its logits are needed (the mask bias takes min cc exactly), then the exact segment maximum is checked
    // against every row's bound.  Only the key images are staged, through a ring of as many key-only
    // buffers as the LDS of the normal stages holds, one barrier per stage.
    if constexpr (SKIP) if ((gflags & 4u) && blk == kBlocks - 1) {
        // The copy runs in chunks of CH stages (one barrier per chunk) through a ring of RC chunk buffers,
        // RC - 1 chunks ahead.
        constexpr int CH = skip_ring(DKS, DVB, X3) >= 6 ? 2 : 1, RC = skip_ring(DKS, DVB, X3) / CH;
        constexpr int CU = CH * KUn;                  // 1 KiB units per chunk
        constexpr int NLC = (CU + NW - 1) / NW;       // direct loads per wave and chunk
        static_assert(RC >= 3 && (size_t)RC * CH * KU * 16 <= skip_lds_bytes(DKS, DVB, X3), "logits-only ring");
        wait_direct_loads();          // stages the previous segment requested ahead: let them land, unused
        prefetched = 0;
        __syncthreads();
        const int nchunk = (slen + CH - 1) / CH;
        const int last_unit = slen * KUn - 1;         // (units past the segment repeat its last one, unread)
        auto issue_chunk = [&](int c, int buf) {
            const unsigned dk = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * CH * KU) * 16);
            static_for<0, NLC>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int u = (CU % NW == 0 || (i + 1) * NW <= CU) ? wave + i * NW : min(wave + i * NW, CU - 1);
                glds16(gk + (size_t)st0 * KU + (size_t)min(c * CU + u, last_unit) * 64, lane_bytes, dk + u * 1024);
            });
        };
        float lmax[QT], lmin[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) { lmax[t] = -SID_INF; lmin[t] = SID_INF; }
        const int nfull_ = max(0, min(slen, P.Nk / (KVS * 32) - st0));
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): nothing the compiler counts is outstanding
        // chunks past the end repeat the last one (into a free buffer, unread): every iteration issues the
        // same number of loads, so "chunk c has landed" is a constant count of younger loads
        for (int k = 0; k < RC - 1; ++k) issue_chunk(min(k, nchunk - 1), k);
        for (int c = 0; c < nchunk; ++c) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RC - 2) * NLC) : "memory");
            __syncthreads();          // chunk c is visible; every wave is done with chunk c - 1
            issue_chunk(min(c + RC - 1, nchunk - 1), (c + RC - 1) % RC);
            const u32x4* L = lds + (c % RC) * CH * KU;
            if (!wave_live) continue;
            if (c * CH + CH <= nfull_) {
#pragma unroll
                for (int j = 0; j < CH * KVS; ++j) {
                    Frags<DKS, X3> kf;
                    kf.load(L + (j * DKS) * 64 + lane, 0);
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        const f32x16 S = qk_tile<DKS, X3, QK16>(kf, qf[t]);
                        lmax[t] = fmaxf(lmax[t], max16(S));
                        lmin[t] = fminf(lmin[t], min16(S));
                    }
                }
            } else {                              // the chunk holds padding keys or ends the segment early
                for (int j = 0; j < CH * KVS; ++j) {
                    if (c * CH + j / KVS >= slen) break;
                    Frags<DKS, X3> kf;
                    kf.load(L + (j * DKS) * 64 + lane, 0);
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        const f32x16 S = qk_tile<DKS, X3, QK16>(kf, qf[t]);
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const bool valid = ((st0 + c * CH) * KVS + j) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh < P.Nk;
                            lmax[t] = valid ? fmaxf(lmax[t], S[i]) : lmax[t];
                            lmin[t] = valid ? fminf(lmin[t], S[i]) : lmin[t];
                        }
                    }
                }
            }
        }
        wait_direct_loads();
        bool ok = true;
        float xmax[QT], xmin[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            xmax[t] = fmaxf(lmax[t], __shfl_xor(lmax[t], 32));
            xmin[t] = fminf(lmin[t], __shfl_xor(lmin[t], 32));
            if (row_token(t) >= 0) ok &= xmax[t] <= row_lb[t] - P.dead_gap;
        }
        skipc += 1u << 8;
        // (the barrier inside also keeps a redo from refilling LDS under a wave that still reads it)
        if (__builtin_amdgcn_readfirstlane(__syncthreads_and((int)ok))) {
            // statistics of a slot that weighs nothing but still reports its logit extremes
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                const int q = row_token(t);
                if (q < 0 || hh != 0) continue;
                const float inv_c2 = 1.0f / c2;           // c2 > 0 whenever skipping is on
                f32x4 st = {-SID_INF, 0.f, xmax[t] * inv_c2, xmin[t] * inv_c2};
                *(f32x4*)(P.stats + (((size_t)part * BH + bh) * P.Nq + q) * 4) = st;
                if (run_w == first_run) {
                    for (int pp = nseg_blk; pp < P.KS; ++pp) {
                        f32x4 none = {-SID_INF, 0.f, -SID_INF, SID_INF};
                        *(f32x4*)(P.stats + ((((size_t)blk * P.KS + pp) * BH + bh) * P.Nq + q) * 4) = none;
                    }
                }
            }
            continue;
        }
        skipc += 1u << 16;            // the prediction failed: the segment runs in full
    }
    if (tid == 0) redo = 0;           // published by the barrier of the prologue below

    // stages of this segment that hold only valid keys (the rest are tails)
    const int nfull = max(0, min(slen, P.Nk / (KVS * 32) - st0));

    // Pass 0 takes every row's largest logit over the FIRST stage as the softmax reference and holds it
    // fixed (no rescaling in the loop).  If afterwards some row of the workgroup turns out to have
    // exceeded its reference by more than 2^kOverflowThr (logits that grow that much along the keys),
    // pass 1 redoes the workgroup with the exact online rescale in every tile.
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
#pragma unroll
            for (int b = 0; b < DVB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[t][b][i] = 0.f;
            rs[t].m2 = -SID_INF; rs[t].smax = -SID_INF; rs[t].smin = SID_INF;
        }
        // the first NB stages start together: their latencies (and that of the query fragments above)
        // overlap
        // (redundant stage copies the pipelined loop left in flight target the same buffers from the
        // same waves: direct loads of a wave complete in order, so the copies issued here win)
        if (prefetched) {
            nissued = prefetched;
            prefetched = 0;
        } else {
            nissued = min(NB, slen);
            for (int k = 0; k < nissued; ++k) issue_stage(st0 + k, k);
        }
        wait_direct_loads();
        __syncthreads();

        stamp(1);
        if (pass == 0) {
            if (nfull > 0) {
                // the reference: every row's largest logit over the first stage (logits only, no P.V)
                if (wave_live) {
                    float m[QT];
#pragma unroll
                    for (int t = 0; t < QT; ++t) m[t] = -SID_INF;
#pragma unroll
                    for (int j = 0; j < KVS; ++j) {
                        Frags<DKS, X3> kf;
                        kf.load(lds + (j * DKS) * 64 + lane, PU);
#pragma unroll
                        for (int t = 0; t < QT; ++t) m[t] = fmaxf(m[t], max16(qk_tile<DKS, X3, QK16>(kf, qf[t])));
                    }
#pragma unroll
                    for (int t = 0; t < QT; ++t) rs[t].m2 = fmaxf(m[t], __shfl_xor(m[t], 32));
                }
                stamp(2);
                if constexpr (SHIFT) {
                    // freeze the reference at a value two bf16 represent exactly and hand it to the MFMA
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        if constexpr (QK16) {
                            // a half pair; beyond half's range the reference stays at the largest pair (any
                            // reference is valid, and a logit that outgrows it is caught below like any other)
                            const float neg = fminf(fmaxf(-rs[t].m2, -60000.f), 60000.f);
                            const _Float16 h1 = (_Float16)neg;
                            const _Float16 h2 = (_Float16)(neg - (float)h1);
                            rs[t].m2 = -((float)h1 + (float)h2);
                            typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
                            f16x2 pr = {h1, h2};
                            set_shift(t, __builtin_bit_cast(unsigned, pr));
                        } else {
                            const __bf16 h1 = (__bf16)(-rs[t].m2);
                            const __bf16 h2 = (__bf16)(-rs[t].m2 - (float)h1);
                            rs[t].m2 = -((float)h1 + (float)h2);
                            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                            bf16x2 pr = {h1, h2};
                            set_shift(t, __builtin_bit_cast(unsigned, pr));
                        }
                    }
                    if constexpr (kPipelined) {
                        if (need_min) run_pipe(T_{}, Shifted{}, 0, nfull);
                        else run_pipe(F_{}, Shifted{}, 0, nfull);
                    } else {
                        if (need_min) run(F_{}, T_{}, Shifted{}, 0, nfull);
                        else run(F_{}, F_{}, Shifted{}, 0, nfull);
                    }
#pragma unroll
                    for (int t = 0; t < QT; ++t) set_shift(t, 0u);
                } else if constexpr (kPipelined) {
                    if (need_min) run_pipe(T_{}, Fixed{}, 0, nfull);
                    else run_pipe(F_{}, Fixed{}, 0, nfull);
                } else {
                    if (need_min) run(F_{}, T_{}, Fixed{}, 0, nfull);
                    else run(F_{}, F_{}, Fixed{}, 0, nfull);
                }
            }
        } else {
            run(F_{}, T_{}, Online{}, 0, nfull);
        }
        stamp(3);
        run(T_{}, T_{}, Online{}, nfull, slen);                      // stages with masked keys
        stamp(4);

        // a logit that outgrew the fixed reference: seen in the tracked maximum, or (where only the
        // minimum is tracked) in a row sum -- the ones column of V -- that left the safe range
        bool over = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const float lrow = hh ? O[t][DVB - 1][15] : 0.f;
            over |= rs[t].smax - rs[t].m2 > kOverflowThr || !(lrow < kOverflowSum);
        }
        if (pass == 0 && __any(over) && lane == 0) redo = 1;
        __syncthreads();
        if (pass == 1 || redo == 0) break;
    }
    // every wave is past its last LDS read of this segment: start the next segment (it begins at
    // stage 0 of the next block); nothing below touches LDS, the images or the query fragments
    if ((tb + 1) * P.SPB < g1) {
        enter_block(real_block(tb + 1));
        prefetched = min(NB, min(g1 - (tb + 1) * P.SPB, P.SPB));
        for (int k = 0; k < prefetched; ++k) issue_stage(k, k);
    }

    // ---- store: combine the two lane halves of every query column
    const int d = P.d;
    const bool vec = (d & 3) == 0;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        // raw q.k extremes from the log2-domain ones: x = c2 * (q.k)
        const float xmax = fmaxf(rs[t].smax, __shfl_xor(rs[t].smax, 32));
        const float xmin = fminf(rs[t].smin, __shfl_xor(rs[t].smin, 32));
        const float inv_c2 = 1.0f / c2;
        const float smax = (c2 > 0.f ? xmax : xmin) * inv_c2;
        const float smin = (c2 > 0.f ? xmin : xmax) * inv_c2;
        // softmax denominator = the ones column = row 31 of the last dv block = register 15, half 1
        float lsum = hh ? O[t][DVB - 1][15] : 0.f;
        lsum += __shfl_xor(lsum, 32);
        const int q = row_token(t);
        if (q < 0) continue;
        if (P.direct) {
            const float inv = 1.0f / lsum;
            const int b_ = bh / P.H, h_ = bh % P.H;
            const size_t obase = ((size_t)b_ * P.Nq + q) * P.H * d + (size_t)h_ * d;
#pragma unroll
            for (int b = 0; b < DVB; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dv0 = 32 * b + 8 * g + 4 * hh;
                    if (vec && !P.out_bf16) {
                        if (dv0 < d) {
                            f32x4 o = {O[t][b][4 * g] * inv, O[t][b][4 * g + 1] * inv, O[t][b][4 * g + 2] * inv,
                                       O[t][b][4 * g + 3] * inv};
                            *(f32x4*)((float*)P.out + obase + dv0) = o;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (dv0 + e < d) {
                                const float val = O[t][b][4 * g + e] * inv;
                                if (P.out_bf16) ((__bf16*)P.out)[obase + dv0 + e] = (__bf16)val;
                                else ((float*)P.out)[obase + dv0 + e] = val;
                            }
                        }
                    }
                }
        } else {
            const size_t row = ((size_t)part * BH + bh) * P.Nq + q;
            float* op = P.opart + row * d;
            if (P.part_f16) {
                typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                _Float16* oh = (_Float16*)P.opart + row * d;
                const float inv = 1.0f / lsum;
                auto sat = [](float x) { return (_Float16)fminf(fmaxf(x, -65504.f), 65504.f); };
#pragma unroll
                for (int b = 0; b < DVB; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int dv0 = 32 * b + 8 * g + 4 * hh;
                        if (vec) {
                            if (dv0 < d) {
                                f16x4 o = {sat(O[t][b][4 * g] * inv), sat(O[t][b][4 * g + 1] * inv), sat(O[t][b][4 * g + 2] * inv),
                                           sat(O[t][b][4 * g + 3] * inv)};
                                *(f16x4*)(oh + dv0) = o;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (dv0 + e < d) oh[dv0 + e] = sat(O[t][b][4 * g + e] * inv);
                        }
                    }
            } else
#pragma unroll
            for (int b = 0; b < DVB; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dv0 = 32 * b + 8 * g + 4 * hh;
                    if (vec) {
                        if (dv0 < d) {
                            f32x4 o = {O[t][b][4 * g], O[t][b][4 * g + 1], O[t][b][4 * g + 2], O[t][b][4 * g + 3]};
                            *(f32x4*)(op + dv0) = o;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (dv0 + e < d) op[dv0 + e] = O[t][b][4 * g + e];
                    }
                }
            if (hh == 0) {
                f32x4 st = {rs[t].m2, lsum, smax, smin};
                *(f32x4*)(P.stats + row * 4) = st;
                // a block cut into fewer segments than there are partial slots per block: the first
                // segment marks the unused slots empty (sum 0; the merge skips them)
                if (run_w == first_run) {
                    for (int pp = nseg_blk; pp < P.KS; ++pp) {
                        f32x4 none = {-SID_INF, 0.f, -SID_INF, SID_INF};
                        *(f32x4*)(P.stats + ((((size_t)blk * P.KS + pp) * BH + bh) * P.Nq + q) * 4) = none;
                    }
                }
            }
        }
    }
    stamp(5);
#ifdef SID_DIAG
    ++diag_seg;
    stamp(0);
#endif
    }   // segments
    wait_direct_loads();      // the pipelined loop leaves (redundant) stage copies in flight
    if (SKIP && tid == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((skipc >> (8 * i)) & 255u) atomicAdd(P.skip_count + i, (skipc >> (8 * i)) & 255u);
    }
}

template <int DKS, int DVB, bool X3, int NW, bool SHIFT, bool QK16, bool SKIP = false>
int launch_flash_inst(const FlashParams& fp, const Geometry& g, hipStream_t stream) {
    constexpr int KVS = stage_tiles(DKS, DVB, X3);
    constexpr int QT = query_tiles(DKS, DVB, X3);
    constexpr auto kern = flash_kernel<DKS, DVB, KVS, X3, QT, NW, SHIFT, QK16, SKIP>;
    constexpr size_t lds_need = SKIP ? skip_lds_bytes(DKS, DVB, X3)
                                     : (size_t)ring_stages(DKS, DVB, X3) * KVS * tile_bytes(DKS, DVB, X3);
    if (lds_need > 48 * 1024) {
        const int rc = sid::allow_dynamic_lds<kern>(lds_need);
        if (rc != SID_OK) return rc;
    }
    const long long grid = (long long)fp.nQG * fp.B * fp.H * fp.W;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), SKIP ? lds_need : g.lds, stream, fp);
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}

template <int DKS, int DVB, bool X3, bool SHIFT, bool QK16>
int launch_flash_nw(const FlashParams& fp, const Geometry& g, hipStream_t stream) {
    if constexpr (!X3) {
        // block skipping is only on for long key blocks, whose calls take workgroups of 8 or 4 waves
        if (fp.rowmap) {
            if (g.NW == 8) return launch_flash_inst<DKS, DVB, X3, 8, SHIFT, QK16, true>(fp, g, stream);
            if (g.NW == 4) return launch_flash_inst<DKS, DVB, X3, 4, SHIFT, QK16, true>(fp, g, stream);
            return sid::fail(SID_EINVAL, "internal: block skipping with %d waves per workgroup", g.NW);
        }
    }
    switch (g.NW) {
        case 8: return launch_flash_inst<DKS, DVB, X3, 8, SHIFT, QK16>(fp, g, stream);
        case 4: return launch_flash_inst<DKS, DVB, X3, 4, SHIFT, QK16>(fp, g, stream);
        case 2: return launch_flash_inst<DKS, DVB, X3, 2, SHIFT, QK16>(fp, g, stream);
        default: return launch_flash_inst<DKS, DVB, X3, 1, SHIFT, QK16>(fp, g, stream);
    }
}

// The half-precision logits mode is instantiated for every head-dim class the bf16 mode serves.
constexpr bool f16qk_instance(int DKS, int DVB) { return DKS > 0 && DVB > 0; }

template <bool X3, bool QK16 = false>
int launch_flash(const FlashParams& fp, const Geometry& g, hipStream_t stream) {
    // (DKS, DVB) pairs that contain a head_dim with head_dim % 16 == 8 also have a SHIFT instance
#define SID_CASE(a, b, has_shift)                                                                            \
    if constexpr (!QK16 || f16qk_instance(a, b)) {                                                           \
        if (g.DKS == a && g.DVB == b) {                                                                      \
            if constexpr (has_shift) {                                                                       \
                if (fp.shift_cols) return launch_flash_nw<a, b, X3, true, QK16>(fp, g, stream);              \
            }                                                                                                \
            return launch_flash_nw<a, b, X3, false, QK16>(fp, g, stream);                                    \
        }                                                                                                    \
    }
    SID_CASE(2, 1, true) SID_CASE(2, 2, false) SID_CASE(3, 2, true) SID_CASE(4, 2, true) SID_CASE(4, 3, false)
    SID_CASE(5, 3, true) SID_CASE(6, 4, false) SID_CASE(8, 5, false) SID_CASE(10, 6, false)
#undef SID_CASE
    return sid::fail(SID_ENOSYS, "no flash kernel for DKS=%d DVB=%d", g.DKS, g.DVB);
}

#endif  // __HIPCC__

}  // namespace sid_attn
