/*
 * sid.h -- C ABI of libsid.so: the mask-based multi-style injection path on MI355X (gfx950).
 *
 * The reference implementation is pure Python/PyTorch and has no FFI; each entry point below
 * replaces the tensor arithmetic of one reference call site (cited per function, paths
 * relative to the reference repository root).  A binding needs only plain pointers and sizes.
 *
 * Conventions
 *   - Every tensor pointer is DEVICE memory on the current HIP device; the caller owns it.
 *   - Inputs are never written.  Work is enqueued on `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream); no entry point synchronises, allocates or frees.
 *   - Return value: SID_OK, or a negative errno-style code; sid_last_error() then returns a
 *     thread-local description.  Nothing is enqueued when an argument check fails.
 *   - Scratch memory is supplied by the caller: ask sid_*_workspace_bytes() and pass a
 *     256-byte aligned buffer at least that large.
 */
#ifndef SID_H
#define SID_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SID_VERSION 100 /* 0.1.0 */

/* return codes */
#define SID_OK 0
#define SID_EINVAL (-22) /* bad argument (shape, dtype, alignment, NULL) */
#define SID_ENOMEM (-12) /* workspace too small */
#define SID_ENOSYS (-38) /* valid request this build has no kernel for */
#define SID_EIO (-5)     /* HIP runtime error at launch */

/* storage type of the q/k/v/out tensors */
#define SID_DTYPE_F32 0
#define SID_DTYPE_BF16 1

/* arithmetic of the two matrix products inside the attention kernels */
#define SID_COMPUTE_BF16 0  /* bf16 MFMA operands, f32 accumulate: <= 2e-2 relative at unit-variance q, k (see F16QK) */
#define SID_COMPUTE_F32X3 1 /* split-bf16 (hi+lo) 3-product MFMA, f32-grade: <= 1e-3 relative */
/* Half-precision (f16) operands for the logits product q k^T, bf16 for P V, f32 accumulate: the speed of
 * SID_COMPUTE_BF16 with eight times finer logits.  The error of a bf16-operand logit grows with |q||k|:
 * <= 2e-2 relative at unit-variance features, 2.5-3.2e-2 at three times that (feature-sized logits), for
 * any implementation that rounds q and k to bf16; this mode stays <= 1e-2 at every scale tested and is what
 * the Python layer selects by default.  Scaled queries and keys beyond +-65504 saturate.  Every head dim and
 * sequence length the bf16 mode serves; tensors stored as bf16 carry no more than bf16 holds, so for them the
 * mode equals SID_COMPUTE_BF16.  More than 256 keys (the long-sequence kernels scale the queries before the
 * product): |logit scale * log2(e)| must lie in [2^-10, 2^10], else SID_EINVAL. */
#define SID_COMPUTE_F16QK 2

/* memory layout of a (batch, heads, tokens, head_dim) tensor; head_dim is always contiguous */
#define SID_LAYOUT_BHND 0 /* (B*H, N, d) contiguous: the precomputed feature lists            */
#define SID_LAYOUT_BNHD 1 /* (B, N, H*d) contiguous: the output of a Linear projection        */

/* flags (bit mask) */
/* Injected attention only: leave out a key block for a workgroup of queries when its whole softmax mass is
 * provably below half an ulp of the f32 row sums.  The mask bias (min cc - max q.k) * (-+m) is built from
 * raw products: at feature-sized logits it sinks the style block of the far mask side below f32
 * resolution on every row with |mask| > T * scale, and often the content block with it.
 * With the flag the queries of an image are regrouped by mask side (workgroups then sit on one side
 * whatever the shape of the region), every workgroup bounds its rows' logits from sampled key tiles and
 *   - drops the far style block for all its rows when the bounds prove it dead (the two remaining
 *     blocks are then cut into the same number of runs, each a third shorter);
 *   - reduces a content block it predicts dead to a pass over its keys alone (the mask bias needs
 *     min cc exactly) and verifies the prediction with the exact maximum, redoing the segment if wrong.
 * Nothing is dropped on belief: results differ from the unflagged call only the way two decompositions of
 * one call differ (runs round their probabilities to bf16 against their own reference).
 * Takes effect with SID_COMPUTE_BF16 / F16QK, T > 0, scale > 0 and N >= 2048; elsewhere the flag is ignored.
 * Measured on (B,8,4096,40), disc and real content masks, features two to three times unit variance:
 * 15-33 % less time per call at B = 16 (a shared GPU), 7-12 % at one image; about 15-22 % MORE when no block can be dropped
 * (the regrouping pads every mask side to a whole workgroup, and the bounds cost a few microseconds), so
 * set it only for feature-sized logits.  sid_injected_attn_skip_counts_offset tells what was dropped. */
#define SID_FLAG_BLOCK_SKIP 1
/* The call shares the GPU with other work of the caller (other streams): choose the decomposition
 * with the least total work (no key-range splits, fewer partial results) instead of the one that
 * fills an otherwise idle GPU. */
#define SID_FLAG_THROUGHPUT 2
/* Stage selection for measurement: an attention call is pack -> flash -> merge.  With none of
 * these bits set all three run.  A call that leaves out an earlier stage reuses what a previous
 * call of the SAME entry point with the same shape arguments, mode and flags left in the same workspace
 * (every entry point lays the workspace out its own way); after anything else the results are garbage
 * (never addresses: what a kernel reads from the workspace is range-checked before it is used as an index). */
#define SID_FLAG_STAGE_PACK 0x10
#define SID_FLAG_STAGE_FLASH 0x20
#define SID_FLAG_STAGE_MERGE 0x40

const char* sid_last_error(void);
int sid_version(void);

/*
 * Injected three-block self-attention.
 * Replaces ldm/modules/attention.py:370-381 (query blend), :201-207 (content logits),
 * :210-329 x2 (style logits + per-row mask bias), :538-546 (concat softmax, PV, head merge)
 * of CrossAttention.forward; to_q before and to_out after stay with the caller.
 *
 *   x_q         (B, N, H*d)   to_q(x) of the current latent            [SID_LAYOUT_BNHD]
 *   q_cnt       (B*H, N, d)   content query        (q_injected)
 *   k1, v1      (B*H, N, d)   character style      (k_injected, v_injected)
 *   k2, v2      (B*H, N, d)   background style     (sty2_k_injected, sty2_v_injected)
 *   k_cnt,v_cnt (B*H, N, d)   content              (cnt_k_injected, cnt_v_injected)
 *   mask_signed (B, N) f32    per-token mask from sid_mask_downsample with ch=+1:
 *                             +1 character ... -1 background
 *   out         (B, N, H*d)   'h (b n) d -> b n (h d)' result, input of to_out
 *
 *   q      = q_cnt*gamma + x_q*q_self_coef          (reference: q_self_coef is the literal 0.5)
 *   logits = [ (q k1^T)*T*scale + b1 | (q k2^T)*T*scale + b2 | (q_cnt k_cnt^T)*scale ]
 *   b1 = (min_j cc - max_j q k1^T) * (-mask),  b2 = (min_j cc - max_j q k2^T) * (+mask)
 *   out = softmax(logits) [v1; v2; v_cnt]
 * The N x 3N logits are never written to memory.
 */
size_t sid_injected_attn_workspace_bytes(int B, int H, int N, int d, int compute);

int sid_injected_attn_fwd(const void* x_q, const void* q_cnt,
                          const void* k1, const void* v1,
                          const void* k2, const void* v2,
                          const void* k_cnt, const void* v_cnt,
                          const float* mask_signed, void* out,
                          int B, int H, int N, int d,
                          float gamma, float q_self_coef, float T, float scale,
                          int dtype, int compute, int flags,
                          void* workspace, size_t workspace_bytes, void* stream);

/*
 * Pre-packed K/V: the feature store's side of the injected attention.
 * The key/value tensors of a call come from the precomputed feature files (run_ori.py:168-222 loads
 * them, run_ori.py:36-75 re-keys them) and never change between images, so their conversion to the
 * kernels' operand layout ("fragment image": bf16 MFMA operand order, head_dim padded, V transposed
 * with a ones column) can happen once, when a feature file is loaded, instead of in every call.
 * No counterpart in the reference, which keeps the tensors in fp32 and re-reads them per call.
 *
 *   sid_kv_image_bytes   size of the image of one (k, v) pair of shape (B*H, N, d)
 *   sid_pack_kv          k, v (B*H, N, d) in `dtype` -> image (256-byte aligned, caller-owned)
 *   sid_injected_attn_fwd_packed
 *                        sid_injected_attn_fwd with the three (k, v) pairs given as images made with the
 *                        same B, H, N, d and compute; x_q, q_cnt and out are tensors of `dtype` as before.
 *                        Same results bit for bit; the workspace is smaller
 *                        (sid_injected_attn_packed_workspace_bytes).
 */
size_t sid_kv_image_bytes(int B, int H, int N, int d, int compute);

int sid_pack_kv(const void* k, const void* v, void* image, size_t image_bytes,
                int B, int H, int N, int d, int dtype, int compute, void* stream);

size_t sid_injected_attn_packed_workspace_bytes(int B, int H, int N, int d, int compute);

int sid_injected_attn_fwd_packed(const void* x_q, const void* q_cnt,
                                 const void* kv1_image, const void* kv2_image, const void* kv_cnt_image,
                                 const float* mask_signed, void* out,
                                 int B, int H, int N, int d,
                                 float gamma, float q_self_coef, float T, float scale,
                                 int dtype, int compute, int flags,
                                 void* workspace, size_t workspace_bytes, void* stream);

/*
 * Cached content block: the feature store's other side of the injected attention.
 * The third key block of a call, softmax-partial of (q_cnt k_cnt^T)*scale over v_cnt
 * (ldm/modules/attention.py:201-207 and the content third of :538-546), reads nothing but the content
 * image's feature file: q_cnt, k_cnt, v_cnt of one (timestep, layer).  It does not depend on the latent,
 * on the styles, on gamma or on T, so a content image that is stylised with several style pairs can
 * have it computed once per (timestep, layer) and reused: a third of the call's matrix work.
 * No counterpart in the reference, which recomputes it in every call.
 *
 *   sid_content_partial_bytes   size of the partial of one (q_cnt, k_cnt, v_cnt) triple
 *   sid_content_partial         q_cnt (B*H, N, d) in `dtype` and the content pair's image (sid_pack_kv,
 *                               same B, H, N, d, compute) -> partial (256-byte aligned, caller-owned):
 *                               per (head, row) the block's normalised output row (halves; f32 with
 *                               SID_COMPUTE_F32X3), its softmax reference and row sum, and the extremes
 *                               of q_cnt.k_cnt that the mask bias needs.  Workspace:
 *                               sid_content_partial_workspace_bytes.
 *   sid_injected_attn_fwd_cached
 *                               sid_injected_attn_fwd_packed with the content pair's image replaced by
 *                               the partial made with the same B, H, N, d, compute, scale and q_cnt.
 *                               Only the two style blocks are computed.  Results equal those of the
 *                               other entry points the way two decompositions of one call are equal
 *                               (each part rounds its probabilities against its own reference).
 *                               SID_FLAG_BLOCK_SKIP applies under the same conditions as in the other
 *                               entry points: the bounds take every row's smallest content logit from
 *                               the partial (exact, no sampling), and a workgroup whose far style block
 *                               is dead runs the near one alone.  Workspace:
 *                               sid_injected_attn_cached_workspace_bytes.
 */
size_t sid_content_partial_bytes(int B, int H, int N, int d, int compute);

size_t sid_content_partial_workspace_bytes(int B, int H, int N, int d, int compute);

int sid_content_partial(const void* q_cnt, const void* kv_cnt_image, void* partial, size_t partial_bytes,
                        int B, int H, int N, int d, float scale, int dtype, int compute,
                        void* workspace, size_t workspace_bytes, void* stream);

size_t sid_injected_attn_cached_workspace_bytes(int B, int H, int N, int d, int compute);

int sid_injected_attn_fwd_cached(const void* x_q, const void* q_cnt,
                                 const void* kv1_image, const void* kv2_image, const void* content_partial,
                                 const float* mask_signed, void* out,
                                 int B, int H, int N, int d,
                                 float gamma, float q_self_coef, float T, float scale,
                                 int dtype, int compute, int flags,
                                 void* workspace, size_t workspace_bytes, void* stream);

/*
 * sid_injected_attn_fwd for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Same arguments and results as sid_injected_attn_fwd without the workspace pair: the scratch is a
 * static device array of SID_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the
 * process ends.  The call allocates and frees nothing.  Its FIRST call on a device looks the array's
 * address up (hipGetSymbolAddress, a runtime call that is not a stream operation): make that first call
 * eagerly, outside HIP graph capture; every later call on the device only enqueues kernels and may be
 * captured.  A shape whose scratch need for this call's decomposition (at most
 * sid_injected_attn_workspace_bytes, which covers every decomposition of the shape) exceeds the array
 * is refused with SID_ENOMEM; 64 MiB covers B = 1, H = 8 at N = 4096, d = 40 and N = 1024, d = 80
 * in SID_COMPUTE_BF16 and SID_COMPUTE_F16QK.  All calls on one GPU share the array: calls that overlap in time on one GPU
 * (different streams or threads) race; serialise them, or give each stream its own workspace through
 * sid_injected_attn_fwd.
 */
#define SID_STATIC_SCRATCH_BYTES (64u << 20)

int sid_injected_attn_fwd_static(const void* x_q, const void* q_cnt,
                                 const void* k1, const void* v1,
                                 const void* k2, const void* v2,
                                 const void* k_cnt, const void* v_cnt,
                                 const float* mask_signed, void* out,
                                 int B, int H, int N, int d,
                                 float gamma, float q_self_coef, float T, float scale,
                                 int dtype, int compute, int flags, void* stream);

/*
 * Where the four block-skipping counters of the last sid_injected_attn_fwd[_packed|_cached] call with these
 * arguments sit in its workspace (byte offset of 4 x uint32: key-block segments dropped, content segments
 * reduced to their logits, such segments redone in full, segments seen), or -1 when the flag has no effect
 * for the shape and mode.  A call also ignores the flag, and then leaves the counters unwritten, when its
 * T or scale is not positive or T * scale * log2(e) < 2^-30: read the counters only after calls outside
 * that range.  `packed`: 0 = sid_injected_attn_fwd (the workspace holds the K/V images), 1 =
 * sid_injected_attn_fwd_packed, 2 = sid_injected_attn_fwd_cached.  No counterpart in the reference:
 * measurement and tests.
 */
long long sid_injected_attn_skip_counts_offset(int B, int H, int N, int d, int compute, int flags, int packed);

/*
 * How a call would be decomposed (host arithmetic only, nothing is launched): for
 * sid_injected_attn_fwd (blocks = 3, Nq = Nk = N), sid_injected_attn_fwd_cached (blocks = 2: the two
 * style blocks; the content block comes from a partial) or sid_plain_attn_fwd (blocks = 1) with the
 * given shape, compute mode and flags, fills
 *   plan[0] runs per (head, query group) over the blocks' key stages   plan[1] stages of the longest run
 *   plan[2] stages per block          plan[3] partial slots per block   plan[4] waves per workgroup
 *   plan[5] query tiles per wave      plan[6] query groups              plan[7] keys per stage
 *   plan[8 .. 8 + plan[0]]  run boundaries: run w covers stages [plan[8 + w], plan[9 + w]) of the
 *                           sequence of blocks * plan[2] stages (at most 24 runs)
 * or plan[0] = 0 when the short-sequence kernel takes the call (it has no decomposition).
 * No counterpart in the reference: a planning query for tests and for callers that size their own
 * stream pools.
 */
int sid_attn_plan(int blocks, int B, int H, int Nq, int Nk, int d, int compute, int flags, int plan[40]);

/*
 * Single-block attention  out = softmax((q k^T) * logit_scale) v.
 * Replaces ldm/modules/attention.py:556-566 (mask present, not injected: logit_scale = T*scale),
 * :569-579 (no mask file / cross-attention: T*scale if anything was injected, else scale) and
 * :584-604 without the boolean `mask` fill (gen_pkl: scale).
 *   q (B,H,Nq,d), k,v (B,H,Nk,d) in q_layout / kv_layout;  out (B, Nq, H*d).
 */
size_t sid_plain_attn_workspace_bytes(int B, int H, int Nq, int Nk, int d, int compute);

int sid_plain_attn_fwd(const void* q, const void* k, const void* v, void* out,
                       int B, int H, int Nq, int Nk, int d, float logit_scale,
                       int q_layout, int kv_layout,
                       int dtype, int compute, int flags,
                       void* workspace, size_t workspace_bytes, void* stream);

/*
 * Content mask -> signed per-token mask.
 * Replaces ldm/modules/attention.py:270-284: threshold (<0.5 -> -1, >0.5 -> +1, ==0.5 kept),
 * multiply by ch, bilinear resize (align_corners=False, no antialias) to (Hout, Wout).
 *   mask (Hin, Win) f32 in [0,1];  out (Hout*Wout) f32.  Bit-exact with the reference for
 *   its inputs (binary 512x512 masks, power-of-two outputs).
 */
int sid_mask_downsample(const float* mask, int Hin, int Win, int Hout, int Wout, float ch,
                        float* out, void* stream);

/*
 * High-frequency radius filter, optionally fused with the patched ResBlock sum.
 * Replaces high_frequency_final.py:4-36 (fft2 ortho, shift, zero the disc
 * (y-H/2)^2+(x-W/2)^2 < int(min(H,W)*ratio)^2, unshift, ifft2, real part) and :82-88
 * (out = out_skip + out_h + HF(h_cnt)).
 *   h_cnt (B,C,H,W); skip, h (B,C,H,W) or both NULL for the bare filter; out (B,C,H,W).
 *   H and W powers of two in [8, 128].
 */
int sid_hf_filter(const void* h_cnt, const void* skip, const void* h, void* out,
                  int B, int C, int H, int W, float ratio, int dtype, void* stream);

/* Same, with the disc radius given as the integer the reference computes on the host,
 * int(min(H,W) * radius_ratio) (high_frequency_final.py:18), so that a binding that holds the
 * ratio as a double truncates exactly as the reference does. */
int sid_hf_filter_radius(const void* h_cnt, const void* skip, const void* h, void* out,
                         int B, int C, int H, int W, int radius, int dtype, void* stream);

/*
 * The patched ResBlock sum with HF(h_cnt) made earlier:  out = (skip + h) + hf  over n elements.
 * Replaces high_frequency_final.py:82-88 for callers that keep the filtered content residual of a
 * (timestep, block) -- sid_hf_filter with skip = h = NULL -- because it depends on the content image and
 * the ratio only (high_frequency_final.py:70-81): one content image stylised with several style pairs
 * filters once.  Same additions in the same order as the fused form of sid_hf_filter: same bits.
 *   skip, h, hf, out: n elements of `dtype`, aligned to four elements.
 */
int sid_residual_sum(const void* skip, const void* h, const void* hf, void* out, long long n, int dtype, void* stream);

/*
 * Masked AdaIN initial latent.
 * Replaces run_ori.py:77-82 (adain: per-channel mean / unbiased std over (batch, H, W)) and
 * run_ori.py:228-232 (bilinear resize of the raw [0,1] mask to the latent size, no threshold;
 * out = m * adain(cnt, char) + (1 - m) * adain(cnt, back)).
 *   cnt, sty_char, sty_back, out (B, C, H, W);  mask (Hin, Win) f32 on the device.
 *   mask == NULL and sty_back == NULL: out = adain(cnt, sty_char), the single-style form.
 */
int sid_masked_adain(const void* cnt, const void* sty_char, const void* sty_back, const float* mask,
                     int Hin, int Win, void* out, int B, int C, int H, int W, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SID_H */
