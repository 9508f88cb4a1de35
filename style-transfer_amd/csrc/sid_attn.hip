// Injected three-block attention and single-block attention for gfx950 (MI355X).
//
// Up to three kernels per call, all on the caller's stream:
//   1. pack   : f32/bf16 inputs -> bf16 MFMA operand fragments in the workspace.  The query blend
//               q = q_cnt*gamma + x_q*coef happens here; K is zero-padded to a multiple of 16 in
//               head_dim; V gets a column of ones (so P*V also yields the softmax row sums) and is
//               stored transposed in fragment order.  The fragment image of a 32-key tile is
//               exactly what a wave's lanes read, so the main kernel copies it to LDS verbatim
//               and reads it back with conflict-free lane-linear ds_read_b128.
//   2. flash  : the key stages of all blocks of a head form one sequence cut into equal runs; one
//               workgroup per (head, run, group of query tiles).  Each wave runs a flash-style pass
//               over the run's keys (sid_attn_kernel.hpp) for its query tiles and writes, per block
//               segment of the run, an un-normalised partial plus the raw row max/min of q.k that
//               the mask bias needs.
//   3. merge  : per query row, turns the statistics into the two bias terms
//               b = (min_j cc - max_j q.k_b) * (mask*ch_b) and combines the parts' partials into
//               the softmax over the concatenated keys.  Skipped when there is a single part
//               (the flash kernel then writes the final rows itself).
//
#include "sid_attn_kernel.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>

namespace {

using namespace sid_attn;

constexpr int kMaxJobs = 8;
constexpr float LOG2E = 1.4426950408889634f;

// The query images are multiplied by the logit scale, and the raw q.k extremes that the mask bias
// needs are recovered by dividing it out again, so it must not be zero.  A scale of zero (T = 0)
// becomes 2^-40: the logits it leaves (|q.k| * 1e-12) change no result, and a power of two keeps
// the recovered extremes exact.
inline float clamp_scale(float c) {
    const float tiny = 9.094947017729282e-13f;   // 2^-40
    return std::fabs(c) < tiny ? (std::signbit(c) ? -tiny : tiny) : c;
}

// SID_COMPUTE_F16QK keeps the scaled queries in half precision: the logit scale must leave them inside
// half's normal range (the short-sequence kernels scale after the product and have no such limit).
int check_f16qk(int DKS, int DVB, int d, const float* c2, int n) {
    if (!f16qk_instance(DKS, DVB))
        return sid::fail(SID_ENOSYS, "SID_COMPUTE_F16QK has no kernel for head_dim %d; use SID_COMPUTE_BF16", d);
    for (int i = 0; i < n; ++i)
        if (!(std::fabs(c2[i]) >= 0x1p-10f && std::fabs(c2[i]) <= 0x1p10f))
            return sid::fail(SID_EINVAL, "SID_COMPUTE_F16QK needs |logit scale * log2(e)| in [2^-10, 2^10] (got %g): "
                             "use SID_COMPUTE_BF16 for this call", (double)c2[i]);
    return SID_OK;
}

__device__ __forceinline__ u32x4 as_u32x4(bf16x8 v) { return __builtin_bit_cast(u32x4, v); }

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
    float post;         // then value *= post (query images carry the logit scale); 1 = none
    int ones_at;        // row fragments: columns ones_at, ones_at+1 of every valid row are 1 (-1: none)
    int kind;           // 0: row fragments (Q, K)   1: transposed V fragments with ones column
    int vec4;           // kind 0, f32: every source row chunk is 16-byte aligned
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
    int f16_rows;       // row fragments (Q, K) are written as IEEE half instead of bf16 (SID_COMPUTE_F16QK)
    // SID_FLAG_BLOCK_SKIP: the first rm_blocks blocks of the launch (one per image) regroup the tokens by
    // mask side instead of packing
    const float* rm_mask;   // (B, N)
    int* rm_out;            // [B][rm_NP]
    int* rm_groups;         // [B]
    unsigned* rm_counters;  // the call's four skip counters, zeroed here
    int rm_blocks, rm_N, rm_NP, rm_GR, rm_own_middle;
    float rm_ts;
};

template <bool BF16IN>
__device__ __forceinline__ float load_elem(const void* p, long long idx) {
    if (BF16IN) return (float)((const __bf16*)p)[idx];
    return ((const float*)p)[idx];
}

// Regrouping of one image's tokens by mask side: [m > thr | the rest | m < -thr], stable, -1 in the gaps.
// A region boundary cuts through the rows of the feature map, so in token order most workgroups straddle
// it; regrouped, none does, and a workgroup can drop the style block of the far side for all its rows.
// The rows in between (|m| <= thr: the mask's blurred boundary, a few dozen tokens of a real mask, for
// which no block can be dropped) go one of two ways (rm_own_middle):
//   1  every side starts on a multiple of GR positions (the queries of one workgroup): the rows in between
//      get a workgroup of their own, most of whose waves hold padding alone and compute nothing -- a cheap
//      workgroup, and no other one loses its dropped block.  For launches of one round of workgroups, where
//      the slowest workgroup sets the time.
//   0  they ride behind the first side (only the last side starts on a workgroup boundary): they cost that
//      side's last workgroup its dropped block instead of a workgroup of their own.  Least total work: for
//      launches that keep the GPU busy for several rounds.
// Only the order in which workgroups pick up queries changes: the images stay in token order.
__device__ void rowmap_block(const PackParams& P, int b) {
    constexpr int kThreads = 256, kWaves = kThreads / 64;
    __shared__ int wave_sum[3][kWaves];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = P.rm_N, GR = P.rm_GR;
    const float* m = P.rm_mask + (size_t)b * N;
    int* out = P.rm_out + (size_t)b * P.rm_NP;
    const float ts = P.rm_ts;
    auto cls = [&](float v) { return v > ts ? 0 : (v < -ts ? 2 : 1); };
    // Every wave owns a run of consecutive tokens and walks it 64 at a time (coalesced loads that do not
    // depend on each other, so many are in flight); a token's rank inside its side comes from the ballots.
    // Positions inside a side keep the token order.
    const int steps = (N + kThreads - 1) / kThreads;
    const int w0 = wv * steps * 64;
    constexpr int kBatch = 16;     // loads requested together
    // side of the token `lane` of step s_ (-1 past the end), for kBatch steps at once
    auto sides = [&](int s0, int (&c)[kBatch]) {
        float v[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int r = w0 + (s0 + j) * 64 + lane;
            v[j] = m[min(r, N - 1)];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int r = w0 + (s0 + j) * 64 + lane;
            c[j] = (s0 + j < steps && r < N) ? cls(v[j]) : -1;
        }
    };
    int cnt[3] = {0, 0, 0};
    for (int s0 = 0; s0 < steps; s0 += kBatch) {
        int c[kBatch];
        sides(s0, c);
#pragma unroll
        for (int j = 0; j < kBatch; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) cnt[k] += __popcll(__ballot(c[j] == k));
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) wave_sum[k][wv] = cnt[k];
    }
    __syncthreads();
    int first[3], total[3], base[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int lower = 0;
        total[k] = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const int n = wave_sum[k][w];
            lower += w < wv ? n : 0;
            total[k] += n;
        }
        base[k] = lower;
    }
    auto up = [&](int n) { return (n + GR - 1) / GR * GR; };
    first[0] = 0;
    first[1] = P.rm_own_middle ? up(total[0]) : total[0];
    first[2] = up(first[1] + total[1]);
    const int start = first[2] + up(total[2]);      // positions in use
#pragma unroll
    for (int k = 0; k < 3; ++k) base[k] += first[k];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int s0 = 0; s0 < steps; s0 += kBatch) {
        int c[kBatch];
        sides(s0, c);
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int r = w0 + (s0 + j) * 64 + lane;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned long long mk = __ballot(c[j] == k);
                if (c[j] == k) out[base[k] + __popcll(mk & below)] = r;
                base[k] += __popcll(mk);
            }
        }
    }
    for (int p = tid; p < P.rm_NP; p += kThreads) {
        const bool gap = (p >= total[0] && p < first[1]) || (p >= first[1] + total[1] && p < first[2]) || p >= first[2] + total[2];
        if (gap) out[p] = -1;
    }
    if (tid == 0) P.rm_groups[b] = start / GR;
    if (b == 0 && tid < 4) P.rm_counters[tid] = 0u;
}

// One thread per 16-byte fragment.  Row fragments (Q, K) are 8 consecutive elements of one source
// row, so threads are ordered (tile, row, chunk): consecutive lanes read consecutive 32-byte chunks of
// a row (coalesced, two 16-byte loads each when the rows allow it) and scatter their fragment to its
// place in the image.  V^T fragments gather 8 keys of one column; lanes run along the row there.
template <bool BF16IN>
__global__ void __launch_bounds__(256) pack_kernel(PackParams P) {
    if ((int)blockIdx.x < P.rm_blocks) {          // first in the launch: they take longest
        rowmap_block(P, (int)blockIdx.x);
        return;
    }
    int jb = 0;
#pragma unroll
    for (int i = 1; i < kMaxJobs; ++i)
        if (i < P.njobs && (int)blockIdx.x >= P.job[i].first_block) jb = i;
    const PackJob& J = P.job[jb];
    const long long gid = (long long)(blockIdx.x - J.first_block) * 256 + threadIdx.x;
    if (gid >= J.frags) return;
    float val[8];
    long long dst;
    if (J.kind == 0) {
        const int nchunk = 2 * P.DKS;
        const int chunk = (int)(gid % nchunk);
        const long long t1 = gid / nchunk;
        const int r = (int)(t1 & 31);
        const long long t2 = t1 >> 5;
        const int tile = (int)(t2 % J.NT);
        const long long bh = t2 / J.NT;
        const int b = (int)(bh / P.H), h = (int)(bh % P.H);
        const int row = tile * 32 + r;
        const int d0 = 8 * chunk;
        dst = ((bh * J.NT + tile) * P.DKS + (chunk >> 1)) * 64 + (chunk & 1) * 32 + r;
#pragma unroll
        for (int j = 0; j < 8; ++j) val[j] = 0.f;
        if (row < J.N && d0 < P.d) {
            const long long o1 = b * J.sb + h * J.sh + (long long)row * J.sn + d0;
            const long long o2 = J.src2 ? b * J.sb2 + h * J.sh2 + (long long)row * J.sn2 + d0 : 0;
            if (!BF16IN && J.vec4 && d0 + 8 <= P.d) {
                const f32x4* p1 = (const f32x4*)((const float*)J.src + o1);
                const f32x4 a0 = p1[0], a1 = p1[1];
#pragma unroll
                for (int j = 0; j < 4; ++j) { val[j] = a0[j]; val[4 + j] = a1[j]; }
                if (J.src2) {
                    const f32x4* p2 = (const f32x4*)((const float*)J.src2 + o2);
                    const f32x4 w0 = p2[0], w1 = p2[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        val[j] = __fadd_rn(__fmul_rn(val[j], J.c1), __fmul_rn(w0[j], J.c2));
                        val[4 + j] = __fadd_rn(__fmul_rn(val[4 + j], J.c1), __fmul_rn(w1[j], J.c2));
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (d0 + j < P.d) {
                        float v = load_elem<BF16IN>(J.src, o1 + j);
                        if (J.src2) v = __fadd_rn(__fmul_rn(v, J.c1), __fmul_rn(load_elem<BF16IN>(J.src2, o2 + j), J.c2));
                        val[j] = v;
                    }
                }
            }
            if (J.post != 1.f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) val[j] = __fmul_rn(val[j], J.post);
            }
        }
        if (row < J.N && d0 == J.ones_at) val[0] = val[1] = 1.f;
    } else {
        const int lane = (int)(gid & 63);
        const int hh = lane >> 5, r = lane & 31;
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
        dst = gid;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // loads are unconditional (clamped address) so that all eight are in flight together
            const int row = tile * 32 + pv_key(s2, hh, j);
            const float x = load_elem<BF16IN>(J.src, b * J.sb + h * J.sh + (long long)min(row, J.N - 1) * J.sn + min(dv, P.d - 1));
            val[j] = row < J.N ? (dv < P.d ? x : (dv == ones_col ? 1.f : 0.f)) : 0.f;
        }
    }
    if (P.f16_rows && J.kind == 0) {
        // half precision saturates instead of overflowing to infinity (its range ends at 65504)
        f16x8 hv;
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = (_Float16)fminf(fmaxf(val[j], -65504.f), 65504.f);
        J.dst[dst] = __builtin_bit_cast(u32x4, hv);
        return;
    }
    bf16x8 hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) hi[j] = (__bf16)val[j];
    J.dst[dst] = as_u32x4(hi);
    if (J.dst_lo) {
        bf16x8 lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) lo[j] = (__bf16)(val[j] - (float)hi[j]);
        J.dst_lo[dst] = as_u32x4(lo);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// (DKS, DVB) of the kernel instance that serves head_dim d: the smallest instantiated pair that holds
// ceil(d / 16) k-steps and ceil((d + 1) / 32) dv blocks.  A larger pair only adds zero padding (the
// ones column sits in the last dv block of the pair, wherever that is).
bool supported_dims(int d, int* DKS, int* DVB) {
    const int dks = (d + 15) / 16, dvb = (d + 1 + 31) / 32;
    static const int table[][2] = {{2, 1}, {2, 2}, {3, 2}, {4, 2}, {4, 3}, {5, 3}, {6, 4}, {8, 5}, {10, 6}};
    for (auto& t : table)
        if (t[0] >= dks && t[1] >= dvb) { *DKS = t[0]; *DVB = t[1]; return true; }
    return false;
}

// head dims whose two spare padding columns d, d+1 of the last k-step carry the softmax reference
// (the kShifted instances): the instance must not be a rounded-up one
bool has_shift_cols(int d, int DKS) { return d % 16 == 8 && d <= 72 && DKS == (d + 15) / 16; }

// Measurement and test overrides from the environment, validated on every read (values outside the
// ranges below are ignored): SID_TUNE_NW in {1,2,4,8}, SID_TUNE_KS in [1,8], SID_TUNE_W in [1,kMaxRuns],
// SID_TUNE_NO_SMALL=1.  Not for production use: the environment must not change while calls run.
struct Tune { int nw = 0, ks = 0, w = 0; bool no_small = false; };
Tune tune() {
    Tune r;
    auto num = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; };
    const int nw = num("SID_TUNE_NW");
    if (nw == 1 || nw == 2 || nw == 4 || nw == 8) r.nw = nw;
    const int ks = num("SID_TUNE_KS");
    if (ks >= 1 && ks <= 8) r.ks = ks;
    const int w = num("SID_TUNE_W");
    if (w >= 1 && w <= kMaxRuns) r.w = w;
    const char* e = getenv("SID_TUNE_NO_SMALL");
    r.no_small = e && e[0] == '1';
    return r;
}

// Decomposition of one call: waves per workgroup and key splits -- sized so the grid covers the
// 256 CUs with about two waves per SIMD while as many queries as possible share each K/V stage.
// SID_TUNE_NW / SID_TUNE_KS override the choice (measurement only).
// `skip`: block skipping is on; the run table of two blocks is filled in as well.
int make_geometry(int B, int H, int Nq, int Nk, int d, int nblk, bool x3, bool throughput, Geometry* g, bool skip = false) {
    if (!supported_dims(d, &g->DKS, &g->DVB))
        return sid::fail(SID_ENOSYS, "head_dim %d has no kernel instance (supported: 1-160)", d);
    g->KVS = stage_tiles(g->DKS, g->DVB, x3);
    g->QT = query_tiles(g->DKS, g->DVB, x3);
    g->NQT = (Nq + 31) / 32;
    const int nkt = (Nk + 31) / 32;
    const int stages = (nkt + g->KVS - 1) / g->KVS;
    const long long BH = (long long)B * H;
    const int qwaves = (g->NQT + g->QT - 1) / g->QT;      // waves that cover the queries of one (head, part)

    // waves per workgroup: as many as there are query waves, up to 4, so each staged K/V tile is
    // shared by up to 4*QT*32 queries; 8 (both waves of every SIMD from ONE workgroup, one workgroup
    // per CU) where that still leaves 8 query groups per head: the stage copy is shared by twice as
    // many waves and the two waves of a SIMD advance together instead of the older workgroup
    // starving the younger one (measured: 2-4 % at N >= 4096, a loss at N = 1024)
    int nw = qwaves >= 64 ? 8 : 4;
    while (nw > 1 && qwaves < nw) nw >>= 1;
    if (tune().nw) nw = tune().nw;
    g->NW = nw;
    g->nQG = (qwaves + nw - 1) / nw;
    // Runs per (head, query group).  The nblk * stages key stages of a head are cut into W equal runs
    // (a run may span a block boundary).  Cost model in units of one stage: a workgroup costs its
    // G stages plus a fixed ~4 (prologue, store); workgroups execute in rounds of `slots` (two waves
    // per SIMD), and every partial slot per block adds a little to the merge.  Runs shorter than 16
    // stages do not pay (measured), so they are not considered unless the whole sequence is shorter.
    // Concurrent callers (throughput) take one run per block: the fewest partials, no rounding loss
    // to recover because other kernels fill the machine.
    const int T = nblk * stages;
    SID_CHECK_ARG(T <= 65535, "%d key stages per head exceed the 65535 a run boundary can hold", T);
    auto slots_per_block = [&](int G_) {
        int most = 1;
        for (int b = 0; b < nblk; ++b) most = std::max(most, ((b + 1) * stages - 1) / G_ - (b * stages) / G_ + 1);
        return most;
    };
    int W = nblk;
    if (!throughput) {
        // (block skipping launches up to kSkipPadGroups more query groups: they must fit the same rounds)
        const long long pairs = BH * (g->nQG + (skip ? kSkipPadGroups : 0)), slots = 512LL * 4 / nw;      // two waves per SIMD
        double best = 1e30;
        for (int w = 1; w <= 8 * nblk && w <= T; ++w) {
            const int G_ = (T + w - 1) / w;
            if (G_ < std::min(16, stages)) break;
            const long long rounds = (pairs * w + slots - 1) / slots;
            // (with block skipping the runs also cut two-block sequences, which raises the slots per block:
            // there the slot term follows the rows the merge reads, 4096 x 8 being the calibration point)
            const double slot_cost = skip ? 2.0 * std::max(1.0, (double)(BH * Nq) / 32768.0) : 2.0;
            const double cost = (double)rounds * (G_ + 4) + slot_cost * slots_per_block(G_);
            if (cost < best - 1e-9) { best = cost; W = w; }
        }
    }
    if (tune().ks) W = nblk * tune().ks;       // the same number of runs per block
    if (tune().w) W = tune().w;
    W = std::max(1, std::min(std::min(W, T), kMaxRuns));
    g->SPB = stages;
    g->W = W;
    // Run boundaries: equal shares of the cost, where a run that continues into the next block pays
    // kSwitch stages for the change (store, new query image, first stage of the new block: measured
    // 6-7 us against 1.3 us per stage at head_dim 40).  c(s) = s + kSwitch * (block boundaries before
    // s) is cut at its W-quantiles; a cut that falls into the cost of a change sits on the boundary.
    constexpr double kSwitch = 5.0;
    g->KS = 1;
    auto cut = [&](int nb, unsigned short* bound, unsigned char* first_run, unsigned char* nseg) {
        const int Tn = nb * stages;
        auto place = [&](double sw) {
            const double ctot = Tn + sw * (nb - 1);
            bound[0] = 0;
            for (int k = 1; k <= W; ++k) {
                const double x = ctot * k / W;
                int s = Tn;
                for (int b = 0; b < nb; ++b) {
                    const double lo = b * (stages + sw);
                    if (x <= lo) { s = b * stages; break; }
                    if (x < lo + stages) { s = b * stages + (int)std::lround(x - lo); break; }
                }
                s = std::max(s, (int)bound[k - 1] + 1);       // every run holds a stage ...
                s = std::min(s, Tn - (W - k));                  // ... and leaves one for each run after it
                bound[k] = (unsigned short)s;
            }
        };
        place(kSwitch);
        // where every block boundary came out as a run boundary no run changes block, so nobody pays for a
        // change: the shares are then equal shares of the stages alone
        auto crossing = [&]() {
            bool c = false;
            for (int k = 0; k < W; ++k) c |= bound[k] / stages != (bound[k + 1] - 1) / stages;
            return c;
        };
        if (nb > 1 && !crossing()) {
            place(0.0);
            if (crossing()) place(kSwitch);
        }
        for (int b = 0; b < nb; ++b) {
            int first = -1, n = 0;
            for (int k = 0; k < W; ++k)
                if (bound[k] < (b + 1) * stages && bound[k + 1] > b * stages) { if (first < 0) first = k; ++n; }
            first_run[b] = (unsigned char)first;
            nseg[b] = (unsigned char)n;
            g->KS = std::max(g->KS, n);
        }
    };
    for (int b = 0; b < kBlocks; ++b) { g->first_run[b] = 0; g->nseg[b] = 0; }
    cut(nblk, g->bound, g->first_run, g->nseg);
    g->G = 0;
    for (int k = 0; k < W; ++k) g->G = std::max(g->G, (int)g->bound[k + 1] - (int)g->bound[k]);
    // block skipping: a query group whose one style block is dead runs the other two blocks with the same
    // W workgroups (W <= 2 * stages because W <= T and skipping is only on for long blocks)
    for (int k = 0; k <= kMaxRuns; ++k) g->bound2[k] = 0;
    for (int b = 0; b < 2; ++b) { g->first_run2[b] = 0; g->nseg2[b] = 0; }
    if (skip && nblk == kBlocks && W <= 2 * stages) cut(2, g->bound2, g->first_run2, g->nseg2);
    // (content block cached: the style block that is left takes all W runs, so a block has W partial slots)
    if (skip && nblk == 2 && W <= stages) cut(1, g->bound2, g->first_run2, g->nseg2);
    g->NKT = stages * g->KVS;                             // whole stages; surplus tiles are zero
    g->lds = (size_t)ring_stages(g->DKS, g->DVB, x3) * g->KVS * tile_bytes(g->DKS, g->DVB, x3);
    g->q_units = (size_t)BH * g->NQT * g->DKS * 64;
    g->k_units = (size_t)BH * g->NKT * g->DKS * 64;
    g->v_units = (size_t)BH * g->NKT * 2 * g->DVB * 64;
    return SID_OK;
}

// positions per image in the regrouped token order: the launch's query groups, padding included
size_t skip_positions(const Geometry& g) { return (size_t)(g.nQG + kSkipPadGroups) * g.NW * g.QT * 32; }

struct Workspace {
    u32x4* q[2][2];              // [which: 0 blended, 1 content][part: hi, lo]
    u32x4* k[kBlocks][2];
    u32x4* v[kBlocks][2];
    float* opart;                // [parts][BH][Nq][d]
    float* stats;                // [parts][BH][Nq][4]
    int* rowmap;                 // block skipping: [B][NP] token order, [B] groups in use, 4 counters
    int* rowmap_groups;
    unsigned* skip_count;
    size_t skip_count_off;
    size_t bytes;
};

// `with_kv` false: the K/V fragment images live outside the workspace (pre-packed by sid_pack_kv)
void carve(const Geometry& g, int nblk, int nq, bool x3, long long rows, int d, void* base, Workspace* w, bool with_kv = true) {
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = base ? (char*)base + off : nullptr; off += sid::align_up(bytes, 256); return p; };
    const int parts = x3 ? 2 : 1;
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 2; ++p) w->q[i][p] = nullptr;
    for (int i = 0; i < kBlocks; ++i) for (int p = 0; p < 2; ++p) { w->k[i][p] = nullptr; w->v[i][p] = nullptr; }
    for (int i = 0; i < nq; ++i) for (int p = 0; p < parts; ++p) w->q[i][p] = (u32x4*)take(g.q_units * 16);
    if (with_kv) {
        for (int i = 0; i < nblk; ++i) for (int p = 0; p < parts; ++p) w->k[i][p] = (u32x4*)take(g.k_units * 16);
        for (int i = 0; i < nblk; ++i) for (int p = 0; p < parts; ++p) w->v[i][p] = (u32x4*)take(g.v_units * 16);
    }
    const int np = nblk * g.KS;                     // rows = B*H*Nq
    w->opart = w->stats = nullptr;
    if (np > 1) {
        w->opart = (float*)take((size_t)np * rows * d * (x3 ? 4 : 2));      // halves unless f32-grade
        w->stats = (float*)take((size_t)np * rows * 16);
    }
    w->rowmap = w->rowmap_groups = nullptr; w->skip_count = nullptr; w->skip_count_off = 0;
    if (nblk >= 2) {
        const size_t BH = g.q_units / ((size_t)g.DKS * 64) / g.NQT;
        w->skip_count_off = off;
        w->skip_count = (unsigned*)take(256);
        w->rowmap_groups = (int*)take(BH * 4);                       // B entries would do; B is not known here
        w->rowmap = (int*)take(BH * skip_positions(g) * 4);
    }
    w->bytes = off;
}

// A pre-packed K/V pair (sid_pack_kv): [K hi | K lo | V hi | V lo], every part 256-byte aligned, the lo
// parts only in the f32-grade mode.  The layout depends on (B*H, N, d, compute) only.
struct KvImage { u32x4* k[2]; u32x4* v[2]; size_t bytes; };
void carve_kv(const Geometry& g, bool x3, const void* base, KvImage* im) {
    size_t off = 0;
    auto take = [&](size_t bytes) { u32x4* p = base ? (u32x4*)((char*)base + off) : nullptr; off += sid::align_up(bytes, 256); return p; };
    im->k[1] = im->v[1] = nullptr;
    im->k[0] = take(g.k_units * 16);
    if (x3) im->k[1] = take(g.k_units * 16);
    im->v[0] = take(g.v_units * 16);
    if (x3) im->v[1] = take(g.v_units * 16);
    im->bytes = off;
}

// A cached content block (sid_content_partial): one partial slot kept by the caller,
// [rows x d output rows: halves, f32 in the f32-grade mode | rows x 4 f32 statistics], rows = B*H*N.
struct PartialImage { void* rows; float* stats; size_t bytes; };
void carve_partial(long long rows, int d, bool x3, const void* base, PartialImage* im) {
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = base ? (char*)base + off : nullptr; off += sid::align_up(bytes, 256); return p; };
    im->rows = take((size_t)rows * d * (x3 ? 4 : 2));
    im->stats = (float*)take((size_t)rows * 16);
    im->bytes = off;
}

// element strides {batch, head, token} of a layout
void layout_strides(int layout, int H, int N, int d, long long s[3]) {
    if (layout == SID_LAYOUT_BHND) { s[0] = (long long)H * N * d; s[1] = (long long)N * d; s[2] = d; }
    else { s[0] = (long long)N * H * d; s[1] = d; s[2] = (long long)H * d; }
}

void set_strides(PackJob& j, int layout, int H, int N, int d, bool second) {
    long long s[3];
    layout_strides(layout, H, N, d, s);
    if (second) { j.sb2 = s[0]; j.sh2 = s[1]; j.sn2 = s[2]; } else { j.sb = s[0]; j.sh = s[1]; j.sn = s[2]; }
}

// Short key sequences take the single-launch kernel; SID_TUNE_NO_SMALL=1 forces the general path
// (measurement and tests only).
bool use_small_path(int Nk, int d) {
    return small_attn_supported(Nk, d) && !tune().no_small;
}

int launch_pack(PackParams& pp, const long long* frags, hipStream_t stream, int rm_images = 0) {
    pp.rm_blocks = pp.rm_out ? rm_images : 0;
    int blocks = pp.rm_blocks;
    for (int i = 0; i < pp.njobs; ++i) {
        PackJob& j = pp.job[i];
        j.first_block = blocks;
        j.frags = frags[i];
        auto al4 = [](const void* p, long long a, long long b, long long c) {
            return ((uintptr_t)p & 15) == 0 && a % 4 == 0 && b % 4 == 0 && c % 4 == 0;
        };
        j.vec4 = j.kind == 0 && !pp.in_bf16 && al4(j.src, j.sb, j.sh, j.sn) && (!j.src2 || al4(j.src2, j.sb2, j.sh2, j.sn2));
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
    SID_CHECK_ARG(compute == SID_COMPUTE_BF16 || compute == SID_COMPUTE_F32X3 || compute == SID_COMPUTE_F16QK,
                  "unknown compute mode %d", compute);
    SID_CHECK_ARG((long long)B * H * (long long)std::max(Nq, Nk) * d < (1LL << 31), "tensor too large for 32-bit tile indexing");
    return SID_OK;
}

// flash + merge stages shared by the two entry points
// `keep_partial`: the call ends with its partial slots (a content block computed for later calls): nothing
// is normalised or merged.  `cached`: the content block of an injected call comes from such a partial.
int run_flash_merge(FlashParams& fp, const Geometry& g, const Workspace& w, const Stages& stg, const float* mask,
                    void* out, float scale, int B, int H, int Nq, int Nk, int d, int nblk, int dtype, bool x3,
                    hipStream_t stream, bool qk16 = false, bool keep_partial = false, const PartialImage* cached = nullptr) {
    fp.opart = w.opart; fp.stats = w.stats; fp.out = out;
    fp.part_f16 = !x3;
    fp.B = B; fp.H = H; fp.Nq = Nq; fp.Nk = Nk; fp.d = d; fp.NQT = g.NQT; fp.NKT = g.NKT;
    fp.nQG = g.nQG + (fp.rowmap ? kSkipPadGroups : 0);
    fp.KS = g.KS; fp.SPB = g.SPB; fp.W = g.W; fp.nblk = nblk;
    for (int k = 0; k <= g.W; ++k) fp.bound[k] = g.bound[k];
    for (int b = 0; b < kBlocks; ++b) { fp.first_run[b] = g.first_run[b]; fp.nseg[b] = g.nseg[b]; }
    fp.direct = !keep_partial && nblk * g.KS == 1; fp.out_bf16 = dtype == SID_DTYPE_BF16;
#ifdef SID_DIAG
    fp.diag = (unsigned long long*)(getenv("SID_DIAG_PTR") ? strtoull(getenv("SID_DIAG_PTR"), nullptr, 0) : 0ull);
#endif
    if (stg.flash) {
        const int rc = x3 ? launch_flash_x3(fp, g, stream)
                          : (qk16 ? launch_flash_f16qk(fp, g, stream) : launch_flash_bf16(fp, g, stream));
        if (rc != SID_OK) return rc;
    }
    if (!stg.merge || fp.direct || keep_partial) return SID_OK;
    MergeParams mp = {};
    if (cached) { mp.cpart = cached->rows; mp.cstats = cached->stats; }
    mp.opart = w.opart; mp.stats = w.stats; mp.mask = mask; mp.out = out; mp.scale = scale;
    mp.part_f16 = fp.part_f16;
    for (int b = 0; b < kBlocks; ++b) mp.c2[b] = fp.c2[b];
    mp.B = B; mp.H = H; mp.Nq = Nq; mp.d = d; mp.KS = g.KS; mp.nblk = nblk; mp.out_bf16 = fp.out_bf16;
    return launch_merge(mp, stream);
}

}  // namespace

extern "C" size_t sid_injected_attn_workspace_bytes(int B, int H, int N, int d, int compute) {
    Geometry g;
    if (check_common(B, H, N, N, d, SID_DTYPE_F32, compute) != SID_OK) return 0;
    const bool x3 = compute == SID_COMPUTE_F32X3;
    size_t bytes = 0;
    for (int v = 0; v < 4; ++v) {                // enough for any decomposition (SID_FLAG_THROUGHPUT, SID_FLAG_BLOCK_SKIP)
        if (make_geometry(B, H, N, N, d, kBlocks, x3, (v & 1) != 0, &g, (v & 2) != 0) != SID_OK) return 0;
        Workspace w;
        carve(g, kBlocks, 2, x3, (long long)B * H * N, d, nullptr, &w);
        bytes = std::max(bytes, w.bytes);
    }
    return bytes;
}

// Scratch of the entry point that allocates nothing (sid_injected_attn_fwd_static): one array per
// GPU, part of the code object, held from the moment the library is loaded on a device until the
// process ends.  In host code the name below is only a handle of the symbol, never a device address:
// the address the kernels and memsets use comes from hipGetSymbolAddress, once per device (every GPU
// has its own copy at its own address).
static __device__ __align__(256) unsigned char scratch[SID_STATIC_SCRATCH_BYTES];

namespace {

constexpr int kMaxDevices = 64;
std::atomic<void*> g_scratch_ptr[kMaxDevices];

int static_scratch(void** ptr) {
    int dev = 0;
    SID_CHECK_HIP(hipGetDevice(&dev));
    SID_CHECK_ARG(dev >= 0 && dev < kMaxDevices, "device ordinal %d is beyond the %d this library tracks", dev, kMaxDevices);
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        SID_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        SID_CHECK_ARG(p != nullptr && ((uintptr_t)p & 255) == 0, "the static scratch array has no usable device address");
        g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *ptr = p;
    return SID_OK;
}

// SID_FLAG_BLOCK_SKIP takes effect where its bounds hold and where it can pay: positive logit scales (the
// bounds are monotone only then), the f32 row statistics of the bf16 / f16 kernels, and key blocks of at
// least 2048 keys (below that the bounds cost more than the stages they can save).
bool skip_applies(int flags, int compute, int N, int d, float T, float scale) {
    // (SID_TUNE_SKIP_MIN_N moves the length limit, for measurement: at N = 1024, head_dim 80 the flag costs
    // 5 us at one image and returns 11 % at sixteen)
    const char* e = getenv("SID_TUNE_SKIP_MIN_N");
    const int min_n = e && atoi(e) >= 256 ? atoi(e) : 2048;
    return (flags & SID_FLAG_BLOCK_SKIP) && compute != SID_COMPUTE_F32X3 && T > 0.f && scale > 0.f && N >= min_n &&
           T * scale * LOG2E >= 0x1p-30f && scale * LOG2E >= 0x1p-30f;
}
// Margin in log2 units: a block of N keys, every logit `gap` below the row's largest, weighs at most
// N * 2^-gap of the row sum; kDeadGapBase keeps that under half an ulp of f32 with a factor 4 to spare.
// SID_TUNE_DEAD_GAP overrides it (tests: a huge value drops nothing with the same decomposition).
float dead_gap(int N) {
    if (const char* e = getenv("SID_TUNE_DEAD_GAP")) {
        const float v = (float)atof(e);
        if (v > 0.f) return v;
    }
    return kDeadGapBase + std::ceil(std::log2((float)N));
}

// Block skipping, shared by the three-block call and the call with a cached content block: the token
// regrouping rides in the pack launch (one extra workgroup per image), the flash kernel gets the map, the
// sampling plan, the margin and the run table of the shortened sequence.
void set_rowmap_job(PackParams& pp, const Geometry& g, const Workspace& w, const float* mask_signed, int B, int H, int N,
                    float T, float scale) {
    pp.rm_mask = mask_signed; pp.rm_out = w.rowmap; pp.rm_groups = w.rowmap_groups; pp.rm_counters = w.skip_count;
    pp.rm_N = N; pp.rm_NP = (int)skip_positions(g); pp.rm_GR = g.NW * g.QT * 32;
    pp.rm_ts = std::max(T * scale, 0.5f);    // |m| <= 1/2: the blurred boundary of a +-1 mask (its values are +-1, +-1/2, 0)
    // one round of workgroups (one per CU at 8 waves, two at 4): the slowest workgroup sets the time
    const long long wgs = (long long)B * H * (g.nQG + kSkipPadGroups) * g.W;
    pp.rm_own_middle = wgs <= 256LL * (g.NW == 8 ? 1 : 2);
}
void set_skip_params(FlashParams& fp, const Geometry& g, const Workspace& w, const float* mask_signed, int N, float T, float scale) {
    fp.rowmap = w.rowmap; fp.rowmap_groups = w.rowmap_groups; fp.mask = mask_signed; fp.skip_count = w.skip_count;
    fp.NP = (int)skip_positions(g);
    const int full_tiles = N / 32;                      // key tiles without padding
    fp.nsamp = std::min(kProbeTiles, full_tiles);
    fp.samp_stride = full_tiles / fp.nsamp;
    fp.ts = T * scale;
    fp.dead_gap = dead_gap(N);
    for (int k = 0; k <= g.W; ++k) fp.bound2[k] = g.bound2[k];
    for (int b = 0; b < 2; ++b) { fp.first_run2[b] = g.first_run2[b]; fp.nseg2[b] = g.nseg2[b]; }
}

int injected_attn_impl(const void* x_q, const void* q_cnt, const void* k1, const void* v1,
                       const void* k2, const void* v2, const void* k_cnt, const void* v_cnt,
                       const float* mask_signed, void* out, int B, int H, int N, int d,
                       float gamma, float q_self_coef, float T, float scale, int dtype,
                       int compute, int flags, void* workspace, size_t workspace_bytes,
                       bool static_ws, hipStream_t stream, const void* const* kv_images = nullptr) {
    const Stages stg(flags);
    int rc = check_common(B, H, N, N, d, dtype, compute);
    if (rc != SID_OK) return rc;
    if (kv_images) {
        SID_CHECK_ARG(x_q && q_cnt && kv_images[0] && kv_images[1] && kv_images[2] && mask_signed && out, "NULL tensor pointer");
        for (int b = 0; b < kBlocks; ++b)
            SID_CHECK_ARG(((uintptr_t)kv_images[b] & 255) == 0, "K/V images must be 256-byte aligned");
    } else {
        SID_CHECK_ARG(x_q && q_cnt && k1 && v1 && k2 && v2 && k_cnt && v_cnt && mask_signed && out, "NULL tensor pointer");
    }
    SID_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace must be non-NULL and 256-byte aligned");
    SID_CHECK_ARG(((uintptr_t)out & 15) == 0, "out must be 16-byte aligned");
    SID_CHECK_ARG(std::isfinite(gamma) && std::isfinite(q_self_coef) && std::isfinite(T) && std::isfinite(scale),
                  "gamma, q_self_coef, T and scale must be finite");
    const bool x3 = compute == SID_COMPUTE_F32X3;
    Geometry g;
    // block skipping needs positive logit scales (the bounds are monotone only then), the f32 row statistics
    // of the bf16 / f16 kernels, and 64 real keys to sample
    const bool skip = skip_applies(flags, compute, N, d, T, scale);
    rc = make_geometry(B, H, N, N, d, kBlocks, x3, (flags & SID_FLAG_THROUGHPUT) != 0, &g, skip);
    if (rc != SID_OK) return rc;
    Workspace w;
    carve(g, kBlocks, 2, x3, (long long)B * H * N, d, workspace, &w, kv_images == nullptr);
    if (kv_images) {
        for (int b = 0; b < kBlocks; ++b) {
            KvImage im;
            carve_kv(g, x3, kv_images[b], &im);
            for (int p = 0; p < 2; ++p) { w.k[b][p] = im.k[p]; w.v[b][p] = im.v[p]; }
        }
    }
    if (w.bytes > workspace_bytes) {
        if (static_ws)
            return sid::fail(SID_ENOMEM, "B=%d H=%d N=%d d=%d needs %zu bytes of scratch; the static buffer of "
                             "sid_injected_attn_fwd_static holds %zu (use sid_injected_attn_fwd with a caller workspace)",
                             B, H, N, d, w.bytes, workspace_bytes);
        return sid::fail(SID_ENOMEM, "workspace of %zu bytes is smaller than the %zu required", workspace_bytes, w.bytes);
    }

    // logit scales in the log2 domain; applied to the query images by the pack kernel
    const float c2s = clamp_scale(T * scale * LOG2E), c2c = clamp_scale(scale * LOG2E);
    const bool qk16 = compute == SID_COMPUTE_F16QK;
    if (qk16) {
        const float cs[2] = {c2s, c2c};
        rc = check_f16qk(g.DKS, g.DVB, d, cs, 2);
        if (rc != SID_OK) return rc;
    }
    const bool shift_cols = has_shift_cols(d, g.DKS);    // head dims with a shifted-reference instance

    // ---- pack
    if (stg.pack) {
        PackParams pp = {};
        pp.njobs = kv_images ? 2 : 8; pp.H = H; pp.d = d; pp.DKS = g.DKS; pp.DVB = g.DVB; pp.in_bf16 = dtype == SID_DTYPE_BF16;
        pp.f16_rows = qk16;
        long long frags[kMaxJobs];
        auto rowjob = [&](int i, const void* src, u32x4* dst, u32x4* dst_lo, int NT, size_t units) {
            PackJob& j = pp.job[i];
            j = PackJob{};
            j.src = src; j.dst = dst; j.dst_lo = dst_lo; j.kind = 0; j.N = N; j.NT = NT;
            j.post = 1.f; j.ones_at = -1;
            set_strides(j, SID_LAYOUT_BHND, H, N, d, false);
            frags[i] = (long long)units;
        };
        rowjob(0, q_cnt, w.q[0][0], w.q[0][1], g.NQT, g.q_units);      // blended query, in log2-logit units
        pp.job[0].src2 = x_q; pp.job[0].c1 = gamma; pp.job[0].c2 = q_self_coef; pp.job[0].post = c2s;
        set_strides(pp.job[0], SID_LAYOUT_BNHD, H, N, d, true);
        rowjob(1, q_cnt, w.q[1][0], w.q[1][1], g.NQT, g.q_units);      // content query
        pp.job[1].post = c2c;
        if (skip) set_rowmap_job(pp, g, w, mask_signed, B, H, N, T, scale);
        const void* ks[kBlocks] = {k1, k2, k_cnt};
        const void* vs[kBlocks] = {v1, v2, v_cnt};
        for (int b = 0; b < kBlocks && !kv_images; ++b) {
            rowjob(2 + b, ks[b], w.k[b][0], w.k[b][1], g.NKT, g.k_units);
            pp.job[2 + b].ones_at = shift_cols ? d : -1;
            rowjob(5 + b, vs[b], w.v[b][0], w.v[b][1], g.NKT, g.v_units);
            pp.job[5 + b].kind = 1;
        }
        rc = launch_pack(pp, frags, stream, skip ? B : 0);
        if (rc != SID_OK) return rc;
    }
    // ---- flash
    FlashParams fp = {};
    fp.shift_cols = shift_cols;
    for (int b = 0; b < kBlocks; ++b) {
        const int qi = (b == kBlocks - 1) ? 1 : 0;
        fp.q[b] = w.q[qi][0]; fp.q_lo[b] = w.q[qi][1];
        fp.k[b] = w.k[b][0]; fp.k_lo[b] = w.k[b][1];
        fp.v[b] = w.v[b][0]; fp.v_lo[b] = w.v[b][1];
    }
    fp.c2[0] = fp.c2[1] = c2s;
    fp.c2[2] = c2c;
    if (skip) {
        if (g.bound2[g.W] != 2 * g.SPB) return sid::fail(SID_EINVAL, "internal: no two-block run table for W=%d SPB=%d", g.W, g.SPB);
        set_skip_params(fp, g, w, mask_signed, N, T, scale);
    }
    return run_flash_merge(fp, g, w, stg, mask_signed, out, scale, B, H, N, N, d, kBlocks, dtype, x3, stream, qk16);
}

}  // namespace

extern "C" int sid_injected_attn_fwd(const void* x_q, const void* q_cnt, const void* k1, const void* v1,
                                     const void* k2, const void* v2, const void* k_cnt, const void* v_cnt,
                                     const float* mask_signed, void* out, int B, int H, int N, int d,
                                     float gamma, float q_self_coef, float T, float scale, int dtype,
                                     int compute, int flags, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    return injected_attn_impl(x_q, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt, mask_signed, out, B, H, N, d, gamma,
                              q_self_coef, T, scale, dtype, compute, flags, workspace, workspace_bytes, false,
                              (hipStream_t)stream);
}

extern "C" int sid_injected_attn_fwd_static(const void* x_q, const void* q_cnt, const void* k1, const void* v1,
                                            const void* k2, const void* v2, const void* k_cnt, const void* v_cnt,
                                            const float* mask_signed, void* out, int B, int H, int N, int d,
                                            float gamma, float q_self_coef, float T, float scale, int dtype,
                                            int compute, int flags, void* stream) {
    void* ws = nullptr;
    const int rc = static_scratch(&ws);
    if (rc != SID_OK) return rc;
    return injected_attn_impl(x_q, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt, mask_signed, out, B, H, N, d, gamma,
                              q_self_coef, T, scale, dtype, compute, flags, ws, (size_t)SID_STATIC_SCRATCH_BYTES, true,
                              (hipStream_t)stream);
}

extern "C" size_t sid_kv_image_bytes(int B, int H, int N, int d, int compute) {
    Geometry g;
    if (check_common(B, H, N, N, d, SID_DTYPE_F32, compute) != SID_OK) return 0;
    const bool x3 = compute == SID_COMPUTE_F32X3;
    if (make_geometry(B, H, N, N, d, kBlocks, x3, false, &g) != SID_OK) return 0;
    KvImage im;
    carve_kv(g, x3, nullptr, &im);
    return im.bytes;
}

extern "C" int sid_pack_kv(const void* k, const void* v, void* image, size_t image_bytes, int B, int H, int N, int d,
                           int dtype, int compute, void* stream) {
    int rc = check_common(B, H, N, N, d, dtype, compute);
    if (rc != SID_OK) return rc;
    SID_CHECK_ARG(k && v && image, "NULL tensor pointer");
    SID_CHECK_ARG(((uintptr_t)image & 255) == 0, "the image must be 256-byte aligned");
    const bool x3 = compute == SID_COMPUTE_F32X3;
    Geometry g;
    rc = make_geometry(B, H, N, N, d, kBlocks, x3, false, &g);
    if (rc != SID_OK) return rc;
    KvImage im;
    carve_kv(g, x3, image, &im);
    if (im.bytes > image_bytes)
        return sid::fail(SID_ENOMEM, "image of %zu bytes is smaller than the %zu required", image_bytes, im.bytes);
    if (compute == SID_COMPUTE_F16QK && !f16qk_instance(g.DKS, g.DVB))
        return sid::fail(SID_ENOSYS, "SID_COMPUTE_F16QK has no kernel for head_dim %d; use SID_COMPUTE_BF16", d);
    PackParams pp = {};
    pp.njobs = 2; pp.H = H; pp.d = d; pp.DKS = g.DKS; pp.DVB = g.DVB; pp.in_bf16 = dtype == SID_DTYPE_BF16;
    pp.f16_rows = compute == SID_COMPUTE_F16QK;
    long long frags[kMaxJobs];
    for (int i = 0; i < 2; ++i) {
        PackJob& j = pp.job[i];
        j = PackJob{};
        j.src = i == 0 ? k : v; j.dst = i == 0 ? im.k[0] : im.v[0]; j.dst_lo = i == 0 ? im.k[1] : im.v[1];
        j.kind = i; j.N = N; j.NT = g.NKT; j.post = 1.f;
        j.ones_at = (i == 0 && has_shift_cols(d, g.DKS)) ? d : -1;
        set_strides(j, SID_LAYOUT_BHND, H, N, d, false);
        frags[i] = (long long)(i == 0 ? g.k_units : g.v_units);
    }
    return launch_pack(pp, frags, (hipStream_t)stream);
}

extern "C" size_t sid_injected_attn_packed_workspace_bytes(int B, int H, int N, int d, int compute) {
    Geometry g;
    if (check_common(B, H, N, N, d, SID_DTYPE_F32, compute) != SID_OK) return 0;
    const bool x3 = compute == SID_COMPUTE_F32X3;
    size_t bytes = 0;
    for (int v = 0; v < 4; ++v) {
        if (make_geometry(B, H, N, N, d, kBlocks, x3, (v & 1) != 0, &g, (v & 2) != 0) != SID_OK) return 0;
        Workspace w;
        carve(g, kBlocks, 2, x3, (long long)B * H * N, d, nullptr, &w, false);
        bytes = std::max(bytes, w.bytes);
    }
    return bytes;
}

extern "C" int sid_injected_attn_fwd_packed(const void* x_q, const void* q_cnt, const void* kv1_image,
                                            const void* kv2_image, const void* kv_cnt_image,
                                            const float* mask_signed, void* out, int B, int H, int N, int d,
                                            float gamma, float q_self_coef, float T, float scale, int dtype,
                                            int compute, int flags, void* workspace, size_t workspace_bytes,
                                            void* stream) {
    const void* images[kBlocks] = {kv1_image, kv2_image, kv_cnt_image};
    return injected_attn_impl(x_q, q_cnt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, mask_signed, out, B, H, N, d,
                              gamma, q_self_coef, T, scale, dtype, compute, flags, workspace, workspace_bytes, false,
                              (hipStream_t)stream, images);
}

// ---- cached content block ---------------------------------------------------------------------------
extern "C" size_t sid_content_partial_bytes(int B, int H, int N, int d, int compute) {
    if (check_common(B, H, N, N, d, SID_DTYPE_F32, compute) != SID_OK) return 0;
    PartialImage im;
    carve_partial((long long)B * H * N, d, compute == SID_COMPUTE_F32X3, nullptr, &im);
    return im.bytes;
}

namespace {
// the content block alone: one run per (head, query group), so that it fills exactly one partial slot
int content_geometry(int B, int H, int N, int d, bool x3, Geometry* g) {
    const int rc = make_geometry(B, H, N, N, d, 1, x3, true, g);
    if (rc != SID_OK) return rc;
    if (g->W != 1 || g->KS != 1)
        return sid::fail(SID_EINVAL, "the content partial is one run per head (SID_TUNE_W / SID_TUNE_KS must not be set)");
    return SID_OK;
}
}  // namespace

extern "C" size_t sid_content_partial_workspace_bytes(int B, int H, int N, int d, int compute) {
    Geometry g;
    if (check_common(B, H, N, N, d, SID_DTYPE_F32, compute) != SID_OK) return 0;
    const bool x3 = compute == SID_COMPUTE_F32X3;
    if (content_geometry(B, H, N, d, x3, &g) != SID_OK) return 0;
    Workspace w;
    carve(g, 1, 1, x3, (long long)B * H * N, d, nullptr, &w, false);
    return w.bytes;
}

extern "C" int sid_content_partial(const void* q_cnt, const void* kv_cnt_image, void* partial, size_t partial_bytes,
                                   int B, int H, int N, int d, float scale, int dtype, int compute,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_common(B, H, N, N, d, dtype, compute);
    if (rc != SID_OK) return rc;
    SID_CHECK_ARG(q_cnt && kv_cnt_image && partial, "NULL tensor pointer");
    SID_CHECK_ARG(((uintptr_t)kv_cnt_image & 255) == 0 && ((uintptr_t)partial & 255) == 0,
                  "the K/V image and the partial must be 256-byte aligned");
    SID_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace must be non-NULL and 256-byte aligned");
    SID_CHECK_ARG(std::isfinite(scale), "scale must be finite");
    const bool x3 = compute == SID_COMPUTE_F32X3;
    Geometry g;
    rc = content_geometry(B, H, N, d, x3, &g);
    if (rc != SID_OK) return rc;
    const long long rows = (long long)B * H * N;
    PartialImage im;
    carve_partial(rows, d, x3, partial, &im);
    if (im.bytes > partial_bytes)
        return sid::fail(SID_ENOMEM, "partial of %zu bytes is smaller than the %zu required", partial_bytes, im.bytes);
    Workspace w;
    carve(g, 1, 1, x3, rows, d, workspace, &w, false);
    if (w.bytes > workspace_bytes)
        return sid::fail(SID_ENOMEM, "workspace of %zu bytes is smaller than the %zu required", workspace_bytes, w.bytes);
    KvImage kim;
    carve_kv(g, x3, kv_cnt_image, &kim);
    const float c2c = clamp_scale(scale * LOG2E);
    const bool qk16 = compute == SID_COMPUTE_F16QK;
    if (qk16) {
        rc = check_f16qk(g.DKS, g.DVB, d, &c2c, 1);
        if (rc != SID_OK) return rc;
    }
    PackParams pp = {};
    pp.njobs = 1; pp.H = H; pp.d = d; pp.DKS = g.DKS; pp.DVB = g.DVB; pp.in_bf16 = dtype == SID_DTYPE_BF16;
    pp.f16_rows = qk16;
    long long frags[kMaxJobs];
    {
        PackJob& j = pp.job[0];
        j = PackJob{};
        j.src = q_cnt; j.dst = w.q[0][0]; j.dst_lo = w.q[0][1]; j.kind = 0; j.N = N; j.NT = g.NQT;
        j.post = c2c; j.ones_at = -1;
        set_strides(j, SID_LAYOUT_BHND, H, N, d, false);
        frags[0] = (long long)g.q_units;
    }
    rc = launch_pack(pp, frags, stream);
    if (rc != SID_OK) return rc;
    FlashParams fp = {};
    fp.shift_cols = has_shift_cols(d, g.DKS);
    fp.q[0] = w.q[0][0]; fp.q_lo[0] = w.q[0][1];
    fp.k[0] = kim.k[0]; fp.k_lo[0] = kim.k[1];
    fp.v[0] = kim.v[0]; fp.v_lo[0] = kim.v[1];
    fp.c2[0] = c2c;
    fp.content_only = 1;
    Workspace wp = w;               // the one partial slot of the call is the caller's buffer
    wp.opart = (float*)im.rows; wp.stats = im.stats;
    return run_flash_merge(fp, g, wp, Stages(0), nullptr, nullptr, scale, B, H, N, N, d, 1, dtype, x3, stream, qk16, true);
}

extern "C" size_t sid_injected_attn_cached_workspace_bytes(int B, int H, int N, int d, int compute) {
    Geometry g;
    if (check_common(B, H, N, N, d, SID_DTYPE_F32, compute) != SID_OK) return 0;
    const bool x3 = compute == SID_COMPUTE_F32X3;
    size_t bytes = 0;
    for (int v = 0; v < 4; ++v) {                // enough for any decomposition (SID_FLAG_THROUGHPUT, SID_FLAG_BLOCK_SKIP)
        if (make_geometry(B, H, N, N, d, 2, x3, (v & 1) != 0, &g, (v & 2) != 0) != SID_OK) return 0;
        Workspace w;
        carve(g, 2, 1, x3, (long long)B * H * N, d, nullptr, &w, false);
        bytes = std::max(bytes, w.bytes);
    }
    return bytes;
}

extern "C" int sid_injected_attn_fwd_cached(const void* x_q, const void* q_cnt, const void* kv1_image,
                                            const void* kv2_image, const void* content_partial,
                                            const float* mask_signed, void* out, int B, int H, int N, int d,
                                            float gamma, float q_self_coef, float T, float scale, int dtype,
                                            int compute, int flags, void* workspace, size_t workspace_bytes,
                                            void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const Stages stg(flags);
    int rc = check_common(B, H, N, N, d, dtype, compute);
    if (rc != SID_OK) return rc;
    SID_CHECK_ARG(x_q && q_cnt && kv1_image && kv2_image && content_partial && mask_signed && out, "NULL tensor pointer");
    SID_CHECK_ARG(((uintptr_t)kv1_image & 255) == 0 && ((uintptr_t)kv2_image & 255) == 0 && ((uintptr_t)content_partial & 255) == 0,
                  "K/V images and the content partial must be 256-byte aligned");
    SID_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace must be non-NULL and 256-byte aligned");
    SID_CHECK_ARG(((uintptr_t)out & 15) == 0, "out must be 16-byte aligned");
    SID_CHECK_ARG(std::isfinite(gamma) && std::isfinite(q_self_coef) && std::isfinite(T) && std::isfinite(scale),
                  "gamma, q_self_coef, T and scale must be finite");
    const bool x3 = compute == SID_COMPUTE_F32X3;
    Geometry g;
    const bool skip = skip_applies(flags, compute, N, d, T, scale);
    rc = make_geometry(B, H, N, N, d, 2, x3, (flags & SID_FLAG_THROUGHPUT) != 0, &g, skip);
    if (rc != SID_OK) return rc;
    const long long rows = (long long)B * H * N;
    Workspace w;
    carve(g, 2, 1, x3, rows, d, workspace, &w, false);
    if (w.bytes > workspace_bytes)
        return sid::fail(SID_ENOMEM, "workspace of %zu bytes is smaller than the %zu required", workspace_bytes, w.bytes);
    const void* images[2] = {kv1_image, kv2_image};
    for (int b = 0; b < 2; ++b) {
        KvImage im;
        carve_kv(g, x3, images[b], &im);
        for (int p = 0; p < 2; ++p) { w.k[b][p] = im.k[p]; w.v[b][p] = im.v[p]; }
    }
    PartialImage cached;
    carve_partial(rows, d, x3, content_partial, &cached);
    const float c2s = clamp_scale(T * scale * LOG2E), c2c = clamp_scale(scale * LOG2E);
    const bool qk16 = compute == SID_COMPUTE_F16QK;
    if (qk16) {
        rc = check_f16qk(g.DKS, g.DVB, d, &c2s, 1);
        if (rc != SID_OK) return rc;
    }
    if (stg.pack) {                 // the blended query is the only image a call makes
        PackParams pp = {};
        pp.njobs = 1; pp.H = H; pp.d = d; pp.DKS = g.DKS; pp.DVB = g.DVB; pp.in_bf16 = dtype == SID_DTYPE_BF16;
        pp.f16_rows = qk16;
        long long frags[kMaxJobs];
        PackJob& j = pp.job[0];
        j = PackJob{};
        j.src = q_cnt; j.dst = w.q[0][0]; j.dst_lo = w.q[0][1]; j.kind = 0; j.N = N; j.NT = g.NQT;
        j.ones_at = -1;
        set_strides(j, SID_LAYOUT_BHND, H, N, d, false);
        j.src2 = x_q; j.c1 = gamma; j.c2 = q_self_coef; j.post = c2s;
        set_strides(j, SID_LAYOUT_BNHD, H, N, d, true);
        frags[0] = (long long)g.q_units;
        if (skip) set_rowmap_job(pp, g, w, mask_signed, B, H, N, T, scale);
        rc = launch_pack(pp, frags, stream, skip ? B : 0);
        if (rc != SID_OK) return rc;
    }
    FlashParams fp = {};
    fp.shift_cols = has_shift_cols(d, g.DKS);
    if (skip) {
        if (g.bound2[g.W] != g.SPB) return sid::fail(SID_EINVAL, "internal: no one-block run table for W=%d SPB=%d", g.W, g.SPB);
        set_skip_params(fp, g, w, mask_signed, N, T, scale);
        fp.cstats = cached.stats;
    }
    for (int b = 0; b < 2; ++b) {
        fp.q[b] = w.q[0][0]; fp.q_lo[b] = w.q[0][1];
        fp.k[b] = w.k[b][0]; fp.k_lo[b] = w.k[b][1];
        fp.v[b] = w.v[b][0]; fp.v_lo[b] = w.v[b][1];
    }
    fp.c2[0] = fp.c2[1] = c2s;
    fp.c2[2] = c2c;
    return run_flash_merge(fp, g, w, stg, mask_signed, out, scale, B, H, N, N, d, 2, dtype, x3, stream, qk16, false, &cached);
}

extern "C" long long sid_injected_attn_skip_counts_offset(int B, int H, int N, int d, int compute, int flags, int packed) {
    Geometry g;
    if (check_common(B, H, N, N, d, SID_DTYPE_F32, compute) != SID_OK) return -1;
    const bool x3 = compute == SID_COMPUTE_F32X3;
    if (!skip_applies(flags, compute, N, d, 1.f, 1.f)) return -1;
    const int nblk = packed == 2 ? 2 : kBlocks;          // 2: the workspace of sid_injected_attn_fwd_cached
    if (make_geometry(B, H, N, N, d, nblk, x3, (flags & SID_FLAG_THROUGHPUT) != 0, &g, true) != SID_OK) return -1;
    Workspace w;
    carve(g, nblk, nblk == 2 ? 1 : 2, x3, (long long)B * H * N, d, nullptr, &w, packed == 0);
    return (long long)w.skip_count_off;
}

extern "C" int sid_attn_plan(int blocks, int B, int H, int Nq, int Nk, int d, int compute, int flags, int plan[40]) {
    int rc = check_common(B, H, Nq, Nk, d, SID_DTYPE_F32, compute);
    if (rc != SID_OK) return rc;
    SID_CHECK_ARG(blocks >= 1 && blocks <= kBlocks, "blocks must be 1, 2 or %d (got %d)", kBlocks, blocks);
    SID_CHECK_ARG(blocks == 1 || Nq == Nk, "the injected attention has Nq == Nk");
    SID_CHECK_ARG(plan != nullptr, "NULL plan");
    for (int i = 0; i < 40; ++i) plan[i] = 0;
    if (blocks == 1 && use_small_path(Nk, d)) return SID_OK;
    Geometry g;
    rc = make_geometry(B, H, Nq, Nk, d, blocks, compute == SID_COMPUTE_F32X3, (flags & SID_FLAG_THROUGHPUT) != 0, &g,
                       blocks >= 2 && skip_applies(flags, compute, Nq, d, 1.f, 1.f));
    if (rc != SID_OK) return rc;
    plan[0] = g.W; plan[1] = g.G; plan[2] = g.SPB; plan[3] = g.KS; plan[4] = g.NW; plan[5] = g.QT; plan[6] = g.nQG;
    plan[7] = g.KVS * 32;
    for (int k = 0; k <= g.W; ++k) plan[8 + k] = g.bound[k];
    return SID_OK;
}

extern "C" size_t sid_plain_attn_workspace_bytes(int B, int H, int Nq, int Nk, int d, int compute) {
    Geometry g;
    if (check_common(B, H, Nq, Nk, d, SID_DTYPE_F32, compute) != SID_OK) return 0;
    if (use_small_path(Nk, d)) return 256;       // the single-launch path keeps nothing in the workspace
    const bool x3 = compute == SID_COMPUTE_F32X3;
    size_t bytes = 0;
    for (int thr = 0; thr < 2; ++thr) {
        if (make_geometry(B, H, Nq, Nk, d, 1, x3, thr != 0, &g) != SID_OK) return 0;
        Workspace w;
        carve(g, 1, 1, x3, (long long)B * H * Nq, d, nullptr, &w);
        bytes = std::max(bytes, w.bytes);
    }
    return bytes;
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
    const bool x3 = compute == SID_COMPUTE_F32X3;
    constexpr float LOG2E = 1.4426950408889634f;
    if (use_small_path(Nk, d)) {
        if (!stg.flash) return SID_OK;
        long long qs[3], ks[3];
        layout_strides(q_layout, H, Nq, d, qs);
        layout_strides(kv_layout, H, Nk, d, ks);
        return launch_small_attn(q, k, v, out, B, H, Nq, Nk, d, logit_scale * LOG2E, qs, ks, dtype == SID_DTYPE_BF16, x3, stream,
                                 (flags & SID_FLAG_THROUGHPUT) != 0, compute == SID_COMPUTE_F16QK);
    }
    Geometry g;
    rc = make_geometry(B, H, Nq, Nk, d, 1, x3, (flags & SID_FLAG_THROUGHPUT) != 0, &g);
    if (rc != SID_OK) return rc;
    Workspace w;
    carve(g, 1, 1, x3, (long long)B * H * Nq, d, workspace, &w);
    const float c2p = clamp_scale(logit_scale * LOG2E);
    const bool qk16 = compute == SID_COMPUTE_F16QK;
    if (qk16) {
        rc = check_f16qk(g.DKS, g.DVB, d, &c2p, 1);
        if (rc != SID_OK) return rc;
    }
    if (w.bytes > workspace_bytes)
        return sid::fail(SID_ENOMEM, "workspace of %zu bytes is smaller than the %zu required", workspace_bytes, w.bytes);

    if (stg.pack) {
        PackParams pp = {};
        pp.njobs = 3; pp.H = H; pp.d = d; pp.DKS = g.DKS; pp.DVB = g.DVB; pp.in_bf16 = dtype == SID_DTYPE_BF16;
        pp.f16_rows = qk16;
        long long frags[kMaxJobs];
        auto job = [&](int i, const void* src, u32x4* dst, u32x4* dst_lo, int kind, int N, int NT, size_t units, int layout) {
            PackJob& j = pp.job[i];
            j = PackJob{};
            j.src = src; j.dst = dst; j.dst_lo = dst_lo; j.kind = kind; j.N = N; j.NT = NT;
            j.post = 1.f; j.ones_at = -1;
            set_strides(j, layout, H, N, d, false);
            frags[i] = (long long)units;
        };
        job(0, q, w.q[0][0], w.q[0][1], 0, Nq, g.NQT, g.q_units, q_layout);
        pp.job[0].post = c2p;
        job(1, k, w.k[0][0], w.k[0][1], 0, Nk, g.NKT, g.k_units, kv_layout);
        pp.job[1].ones_at = has_shift_cols(d, g.DKS) ? d : -1;
        job(2, v, w.v[0][0], w.v[0][1], 1, Nk, g.NKT, g.v_units, kv_layout);
        rc = launch_pack(pp, frags, stream);
        if (rc != SID_OK) return rc;
    }
    FlashParams fp = {};
    fp.shift_cols = has_shift_cols(d, g.DKS);
    fp.q[0] = w.q[0][0]; fp.q_lo[0] = w.q[0][1];
    fp.k[0] = w.k[0][0]; fp.k_lo[0] = w.k[0][1];
    fp.v[0] = w.v[0][0]; fp.v_lo[0] = w.v[0][1];
    fp.c2[0] = c2p;
    return run_flash_merge(fp, g, w, stg, nullptr, out, 1.f, B, H, Nq, Nk, d, 1, dtype, x3, stream, qk16);
}

// ------------------------------------------------------------------------------------------
// merge: per query row, statistics -> mask bias, then the softmax over the concatenated parts
// ------------------------------------------------------------------------------------------
namespace sid_attn {

// One thread per (row, VEC output columns), thread order (head, query, column): the reads, nparts
// times the volume of the writes, are contiguous across the wave.  NP > 0 fixes the number of parts
// at compile time: all statistics and all partial rows are then requested before the first one is
// used (one memory round trip per thread instead of one per part).
// PART16: the partial rows are halves, normalised by their row sum (a part then weighs w * l).
// CACHED: the call computed the two style blocks (nblk == 2); the content block is one more part, the
// last one, read from a partial kept outside the workspace (P.cpart / P.cstats).
template <int VEC, int NP, bool PART16, bool CACHED>
__global__ void __launch_bounds__(256) merge_kernel(MergeParams P) {
    const int dpv = P.d / VEC;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)P.B * P.H * P.Nq * dpv;
    if (gid >= total) return;
    const int c = (int)(gid % dpv);
    const long long t = gid / dpv;
    const int q = (int)(t % P.Nq);
    const int bh = (int)(t / P.Nq);
    const int b = bh / P.H, h = bh % P.H;
    const int BH = P.B * P.H;
    const int nparts = NP > 0 ? NP : P.nblk * P.KS + (CACHED ? 1 : 0);
    constexpr int CAP = NP > 0 ? NP : 1;
    constexpr float LOG2E = 1.4426950408889634f;
    auto cached_part = [&](int p) { return CACHED && p == nparts - 1; };
    auto stat_ptr = [&](int p) {
        const float* base = cached_part(p) ? P.cstats : P.stats + (size_t)p * BH * P.Nq * 4;
        return (const f32x4*)(base + ((size_t)bh * P.Nq + q) * 4);
    };
    const size_t part_off = ((size_t)bh * P.Nq + q) * P.d + c * VEC, part_stride = (size_t)BH * P.Nq * P.d;
    // VEC values of part p as floats
    auto load_part = [&](int p, float (&o)[VEC]) {
        if (PART16) {
            const _Float16* hp = (cached_part(p) ? (const _Float16*)P.cpart : (const _Float16*)P.opart + p * part_stride) + part_off;
            if (VEC == 4) {
                typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                const f16x4 v = *(const f16x4*)hp;
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = (float)v[e % 4];
            } else {
                o[0] = (float)hp[0];
            }
        } else {
            const float* fp = (cached_part(p) ? (const float*)P.cpart : P.opart + p * part_stride) + part_off;
            if (VEC == 4) {
                const f32x4 v = *(const f32x4*)fp;
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] = v[e % 4];
            } else {
                o[0] = fp[0];
            }
        }
    };

    f32x4 st_[CAP];
    float o_[CAP][VEC];
    if (NP > 0) {
#pragma unroll
        for (int p = 0; p < CAP; ++p) st_[p] = *stat_ptr(p);
#pragma unroll
        for (int p = 0; p < CAP; ++p) load_part(p, o_[p]);
    }
    auto stat = [&](int p) { return NP > 0 ? st_[NP > 0 ? p : 0] : *stat_ptr(p); };

    float bias2[kBlocks] = {0.f, 0.f, 0.f};
    if (P.nblk == kBlocks || CACHED) {
        // parts per block: the workspace holds kBlocks blocks, or (CACHED) the two style blocks
        const int KS = CACHED ? (nparts - 1) / 2 : (nparts >= kBlocks ? nparts / kBlocks : 1);
        float bmax[kBlocks], cmin = SID_INF;
#pragma unroll
        for (int blk = 0; blk < kBlocks; ++blk) bmax[blk] = -SID_INF;
        if (NP > 0) {
#pragma unroll
            for (int p = 0; p < CAP; ++p) {
                constexpr int KSC = CACHED ? (CAP > 2 ? (CAP - 1) / 2 : 1) : (CAP / kBlocks > 0 ? CAP / kBlocks : 1);
                const int blk = (CACHED && p == CAP - 1) ? kBlocks - 1 : p / KSC;
                const f32x4 st = stat(p);
#pragma unroll
                for (int k = 0; k < kBlocks; ++k) bmax[k] = (k == blk) ? fmaxf(bmax[k], st[2]) : bmax[k];
                if (blk == kBlocks - 1) cmin = fminf(cmin, st[3]);
            }
        } else {
            for (int p = 0; p < nparts; ++p) {
                const int blk = cached_part(p) ? kBlocks - 1 : p / KS;
                const f32x4 st = stat(p);
#pragma unroll
                for (int k = 0; k < kBlocks; ++k) bmax[k] = (k == blk) ? fmaxf(bmax[k], st[2]) : bmax[k];
                if (blk == kBlocks - 1) cmin = fminf(cmin, st[3]);
            }
        }
        // min_j (q_cnt.k_cnt * scale): scaling by a constant is monotone, so it commutes with min
        const float min_cc = (P.scale >= 0.f) ? __fmul_rn(cmin, P.scale) : __fmul_rn(bmax[2], P.scale);
        const float m = P.mask[(size_t)b * P.Nq + q];
        bias2[0] = __fmul_rn(__fsub_rn(min_cc, bmax[0]), __fmul_rn(m, -1.0f)) * LOG2E;
        bias2[1] = __fmul_rn(__fsub_rn(min_cc, bmax[1]), __fmul_rn(m, 1.0f)) * LOG2E;
    }
    const int KSr = P.KS;
    auto blk_of = [&](int p) { return cached_part(p) ? kBlocks - 1 : p / KSr; };
    auto bias_of = [&](int p) { const int k = blk_of(p); return k == 0 ? bias2[0] : (k == 1 ? bias2[1] : bias2[2]); };
    // reference of the combined softmax: the largest log2 of a part's biased weight sum, m2 + log2(l)
    // (not the parts' own references m2, which a logit may exceed by up to 2^kOverflowThr)
    float M2 = -SID_INF;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    float den = 0.f;
    if (NP > 0) {
#pragma unroll
        for (int p = 0; p < CAP; ++p) {
            const f32x4 st = stat(p);
            if (st[1] > 0.f) M2 = fmaxf(M2, st[0] + __builtin_amdgcn_logf(st[1]) + bias_of(p));
        }
#pragma unroll
        for (int p = 0; p < CAP; ++p) {
            const f32x4 st = stat(p);
            const float w = st[1] > 0.f ? __builtin_amdgcn_exp2f(st[0] + bias_of(p) - M2) : 0.f;
            const float wl = w * (st[1] > 0.f ? st[1] : 0.f);
            den += wl;
            const float wo = PART16 ? wl : w;
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += st[1] > 0.f ? wo * o_[p][e] : 0.f;    // an unused slot holds no data
        }
    } else {
        for (int p = 0; p < nparts; ++p) {
            const f32x4 st = stat(p);
            if (st[1] > 0.f) M2 = fmaxf(M2, st[0] + __builtin_amdgcn_logf(st[1]) + bias_of(p));
        }
        for (int p = 0; p < nparts; ++p) {
            const f32x4 st = stat(p);
            if (!(st[1] > 0.f)) continue;
            const float w = __builtin_amdgcn_exp2f(st[0] + bias_of(p) - M2);
            const float wl = w * st[1];
            den += wl;
            const float wo = PART16 ? wl : w;
            float o[VEC];
            load_part(p, o);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += wo * o[e];
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
            f32x4 r = {acc[0] * inv, acc[VEC > 1 ? 1 : 0] * inv, acc[VEC > 2 ? 2 : 0] * inv, acc[VEC > 3 ? 3 : 0] * inv};
            *(f32x4*)o = r;
        } else {
            o[0] = acc[0] * inv;
        }
    }
}

template <int VEC, bool PART16>
void launch_merge_np(const MergeParams& mp, unsigned blocks, hipStream_t stream) {
    if (mp.cpart) {                 // two style blocks in the workspace + the cached content part
        switch (mp.nblk * mp.KS + 1) {
            case 3: hipLaunchKernelGGL((merge_kernel<VEC, 3, PART16, true>), dim3(blocks), dim3(256), 0, stream, mp); break;
            case 5: hipLaunchKernelGGL((merge_kernel<VEC, 5, PART16, true>), dim3(blocks), dim3(256), 0, stream, mp); break;
            case 7: hipLaunchKernelGGL((merge_kernel<VEC, 7, PART16, true>), dim3(blocks), dim3(256), 0, stream, mp); break;
            default: hipLaunchKernelGGL((merge_kernel<VEC, 0, PART16, true>), dim3(blocks), dim3(256), 0, stream, mp); break;
        }
        return;
    }
    switch (mp.nblk * mp.KS) {
        case 2: hipLaunchKernelGGL((merge_kernel<VEC, 2, PART16, false>), dim3(blocks), dim3(256), 0, stream, mp); break;
        case 3: hipLaunchKernelGGL((merge_kernel<VEC, 3, PART16, false>), dim3(blocks), dim3(256), 0, stream, mp); break;
        case 4: hipLaunchKernelGGL((merge_kernel<VEC, 4, PART16, false>), dim3(blocks), dim3(256), 0, stream, mp); break;
        case 6: hipLaunchKernelGGL((merge_kernel<VEC, 6, PART16, false>), dim3(blocks), dim3(256), 0, stream, mp); break;
        case 9: hipLaunchKernelGGL((merge_kernel<VEC, 9, PART16, false>), dim3(blocks), dim3(256), 0, stream, mp); break;
        case 12: hipLaunchKernelGGL((merge_kernel<VEC, 12, PART16, false>), dim3(blocks), dim3(256), 0, stream, mp); break;
        default: hipLaunchKernelGGL((merge_kernel<VEC, 0, PART16, false>), dim3(blocks), dim3(256), 0, stream, mp); break;
    }
}

int launch_merge(const MergeParams& mp, hipStream_t stream) {
    const int vec = (mp.d & 3) == 0 ? 4 : 1;
    const long long total = (long long)mp.B * mp.H * mp.Nq * (mp.d / vec);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (mp.part_f16) {
        if (vec == 4) launch_merge_np<4, true>(mp, blocks, stream);
        else launch_merge_np<1, true>(mp, blocks, stream);
    } else {
        if (vec == 4) launch_merge_np<4, false>(mp, blocks, stream);
        else launch_merge_np<1, false>(mp, blocks, stream);
    }
    SID_CHECK_HIP(hipGetLastError());
    return SID_OK;
}

}  // namespace sid_attn
