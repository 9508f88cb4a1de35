"""Tensor-level entry points: torch device tensors in, HIP kernels enqueued on the current stream.

`compute` defaults to "f16" (SID_COMPUTE_F16QK: half-precision query/key operands, bf16 probabilities and
values), the arithmetic that holds the path's 2e-2 bar at feature-sized logits; "bf16" and "fp32" select
the other two.  The long-sequence kernels refuse "f16" for logit scales outside [2^-10, 2^10] / log2(e)
(SID_EINVAL); style_transfer_amd.attention.resolve_compute picks "bf16" for such calls.

PyTorch is used for device memory and streams only; all arithmetic of the path happens in
libsid.so.  Every function raises if the library is missing or rejects the call.
"""
import threading

import torch

from . import lib as _lib

_workspaces = {}

# Held by a thread while it captures a HIP graph (graphed.GraphedUNet) and, in short pieces, by the feature
# loader's worker thread around its GPU calls.  The two must not overlap: with the capturing thread in the
# thread-local capture mode the runtime accepts another thread's allocations and copies during the capture, but
# a blocking copy from pageable memory that races with it fails now and then with "operation not permitted when
# stream is capturing" (measured: once in a few thousand copies), and a failed call invalidates nothing but
# itself, so the loader would lose a file.
capture_lock = threading.RLock()


def _dtype_code(t):
    if t.dtype == torch.float32:
        return _lib.DTYPE_F32
    if t.dtype == torch.bfloat16:
        return _lib.DTYPE_BF16
    raise TypeError("tensors must be float32 or bfloat16, got %s" % t.dtype)


def _compute_code(compute):
    if compute in ("bf16", _lib.COMPUTE_BF16):
        return _lib.COMPUTE_BF16
    if compute in ("fp32", "f32", "f32x3", _lib.COMPUTE_F32X3):
        return _lib.COMPUTE_F32X3
    if compute in ("f16", "f16qk", _lib.COMPUTE_F16QK):
        return _lib.COMPUTE_F16QK
    raise ValueError("compute must be 'bf16', 'f16' or 'fp32', got %r" % (compute,))


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


_retired = []        # superseded workspaces stay alive: a captured HIP graph may hold their address


def _workspace(device, nbytes):
    """One growing scratch buffer per (device, stream): calls on one stream run in order, so
    they can share it; allocation happens here, never inside the library.  A HIP graph captured on
    the stream bakes the buffer's address in, so the buffer must not grow during capture (warm up
    eagerly with the largest shape first) and a buffer that is outgrown later is kept, not freed."""
    key = (device.index, _stream(device))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("libsid workspace of %d bytes would have to be allocated during graph capture: run the "
                               "call sequence once eagerly on this stream first" % nbytes)
        if ws is not None:
            _retired.append(ws)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _need_cuda(*ts):
    dev = ts[0].device
    for t in ts:
        if t.device.type != "cuda":
            raise RuntimeError("libsid kernels need tensors in GPU memory (got %s)" % t.device)
        if t.device != dev:
            raise RuntimeError("all tensors must be on the same device")
        if not t.is_contiguous():
            raise RuntimeError("tensors must be contiguous")
    return dev


def injected_attention(x_q, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt, mask_signed, gamma, T, scale,
                       heads, q_self_coef=0.5, compute="f16", out=None, flags=0, static_scratch=False):
    """x_q (B,N,H*d); q_cnt,k*,v* (B*H,N,d); mask_signed (B,N) or (N,) float32 -> (B,N,H*d).
    static_scratch=True goes through sid_injected_attn_fwd_static (pass `out` to keep the call free of
    allocations altogether); calls on one GPU must then not overlap."""
    L = _lib.load()
    B, N, C = x_q.shape
    H = heads
    d = C // H
    ts = (x_q, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt)
    dev = _need_cuda(*ts, mask_signed)
    for t in ts[1:]:
        if tuple(t.shape) != (B * H, N, d):
            raise ValueError("feature tensors must be (B*H, N, d) = %s, got %s" % ((B * H, N, d), tuple(t.shape)))
        if t.dtype != x_q.dtype:
            raise TypeError("all attention tensors must share one dtype")
    if mask_signed.dtype != torch.float32 or mask_signed.numel() != B * N:
        raise ValueError("mask_signed must be float32 with B*N elements")
    dt = _dtype_code(x_q)
    cc = _compute_code(compute)
    if out is None:
        out = torch.empty((B, N, C), dtype=x_q.dtype, device=dev)
    if static_scratch:
        # the entry point that allocates nothing: scratch is the library's static device array
        with torch.cuda.device(dev):
            _lib.check(L.sid_injected_attn_fwd_static(
                x_q.data_ptr(), q_cnt.data_ptr(), k1.data_ptr(), v1.data_ptr(), k2.data_ptr(), v2.data_ptr(),
                k_cnt.data_ptr(), v_cnt.data_ptr(), mask_signed.data_ptr(), out.data_ptr(),
                B, H, N, d, float(gamma), float(q_self_coef), float(T), float(scale), dt, cc, int(flags),
                _stream(dev)))
        return out
    nbytes = L.sid_injected_attn_workspace_bytes(B, H, N, d, cc)
    if nbytes == 0:
        raise _lib.SidError(-22, L.sid_last_error().decode())
    ws = _workspace(dev, nbytes)
    with torch.cuda.device(dev):
        _lib.check(L.sid_injected_attn_fwd(
            x_q.data_ptr(), q_cnt.data_ptr(), k1.data_ptr(), v1.data_ptr(), k2.data_ptr(), v2.data_ptr(),
            k_cnt.data_ptr(), v_cnt.data_ptr(), mask_signed.data_ptr(), out.data_ptr(),
            B, H, N, d, float(gamma), float(q_self_coef), float(T), float(scale), dt, cc, int(flags),
            ws.data_ptr(), ws.numel(), _stream(dev)))
    return out


FLAG_THROUGHPUT = 2      # SID_FLAG_THROUGHPUT: the caller keeps the GPU busy with other streams
FLAG_BLOCK_SKIP = 1      # SID_FLAG_BLOCK_SKIP: drop key blocks whose softmax mass is provably nil


def skip_counts(x_q, heads, compute="f16", flags=FLAG_BLOCK_SKIP, packed=False, T=None, scale=None):
    """The block-skipping counters of the last injected-attention call made on the current stream with this
    shape, mode, flags and entry point: (query groups with a dead style block, content segments reduced to
    their logits, of those redone in full, query groups), or None where the flag has no effect (then the
    call ran as if it were not set).  `packed`: False = injected_attention, True = injected_attention_packed,
    "cached" = injected_attention_cached.  Pass the call's T and scale: the library ignores the flag for
    non-positive or vanishing logit scales, and the counters of such a call are not written.  Synchronises."""
    L = _lib.load()
    B, N, C = x_q.shape
    if T is not None and scale is not None:
        log2e = 1.4426950408889634
        if not (T > 0 and scale > 0 and T * scale * log2e >= 2.0 ** -30 and scale * log2e >= 2.0 ** -30):
            return None
    off = L.sid_injected_attn_skip_counts_offset(B, heads, N, C // heads, _compute_code(compute), int(flags),
                                                 2 if packed == "cached" else int(bool(packed)))
    if off < 0:
        return None
    ws = _workspaces.get((x_q.device.index, _stream(x_q.device)))
    if ws is None or off + 16 > ws.numel():
        raise RuntimeError("no injected-attention call with this shape has run on the current stream")
    return tuple(int(v) for v in ws[off:off + 16].view(torch.int32).cpu())


class PackedKV:
    """The key/value pair of one feature-list entry in the kernels' operand layout (sid_pack_kv): what a
    feature store keeps in HBM instead of the two (B*H, N, d) tensors.  Stands in for BOTH tensors in
    the injected-attention call (pass the same object as k and v)."""

    def __init__(self, image, B, H, N, d, compute):
        self.image, self.B, self.H, self.N, self.d, self.compute = image, B, H, N, d, compute
        self.partial = None     # a content pair may carry its cached content block (ContentPartial)

    @property
    def device(self):
        return self.image.device

    def nbytes(self):
        return self.image.numel()


def pack_kv(k, v, heads, compute="f16"):
    """k, v (B*H, N, d) float32 / bfloat16 -> PackedKV (one launch on the current stream)."""
    L = _lib.load()
    dev = _need_cuda(k, v)
    if k.shape != v.shape or k.dtype != v.dtype or k.dim() != 3 or k.shape[0] % heads:
        raise ValueError("k and v must be two (B*H, N, d) tensors of one dtype")
    BH, N, d = k.shape
    B = BH // heads
    cc = _compute_code(compute)
    nbytes = L.sid_kv_image_bytes(B, heads, N, d, cc)
    if nbytes == 0:
        raise _lib.SidError(-22, L.sid_last_error().decode())
    image = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.sid_pack_kv(k.data_ptr(), v.data_ptr(), image.data_ptr(), nbytes, B, heads, N, d, _dtype_code(k), cc,
                                 _stream(dev)))
    return PackedKV(image, B, heads, N, d, cc)


def injected_attention_packed(x_q, q_cnt, kv1, kv2, kv_cnt, mask_signed, gamma, T, scale, heads, q_self_coef=0.5,
                              out=None, flags=0):
    """injected_attention with the three key/value pairs pre-packed (PackedKV); same results."""
    L = _lib.load()
    B, N, C = x_q.shape
    H = heads
    d = C // H
    dev = _need_cuda(x_q, q_cnt, mask_signed, kv1.image, kv2.image, kv_cnt.image)
    if tuple(q_cnt.shape) != (B * H, N, d) or q_cnt.dtype != x_q.dtype:
        raise ValueError("q_cnt must be (B*H, N, d) = %s of x_q's dtype" % ((B * H, N, d),))
    cc = kv1.compute
    for kv in (kv1, kv2, kv_cnt):
        if (kv.B, kv.H, kv.N, kv.d, kv.compute) != (B, H, N, d, cc):
            raise ValueError("packed K/V was made for (B,H,N,d,compute) = %s, the call has %s"
                             % ((kv.B, kv.H, kv.N, kv.d, kv.compute), (B, H, N, d, cc)))
    if mask_signed.dtype != torch.float32 or mask_signed.numel() != B * N:
        raise ValueError("mask_signed must be float32 with B*N elements")
    if out is None:
        out = torch.empty((B, N, C), dtype=x_q.dtype, device=dev)
    nbytes = L.sid_injected_attn_packed_workspace_bytes(B, H, N, d, cc)
    if nbytes == 0:
        raise _lib.SidError(-22, L.sid_last_error().decode())
    ws = _workspace(dev, nbytes)
    with torch.cuda.device(dev):
        _lib.check(L.sid_injected_attn_fwd_packed(
            x_q.data_ptr(), q_cnt.data_ptr(), kv1.image.data_ptr(), kv2.image.data_ptr(), kv_cnt.image.data_ptr(),
            mask_signed.data_ptr(), out.data_ptr(), B, H, N, d, float(gamma), float(q_self_coef), float(T), float(scale),
            _dtype_code(x_q), cc, int(flags), ws.data_ptr(), ws.numel(), _stream(dev)))
    return out


class ContentPartial:
    """The content block of the injected attention for one (timestep, layer) of a content feature file
    (sid_content_partial): per (head, row) the block's normalised output row, softmax reference and row
    sum, and the q_cnt.k_cnt extremes the mask bias needs.  It depends on q_cnt, k_cnt, v_cnt and `scale`
    only, so one content image stylised with several style pairs computes it once."""

    def __init__(self, data, B, H, N, d, compute, scale, q_ptr):
        self.data, self.B, self.H, self.N, self.d, self.compute, self.scale = data, B, H, N, d, compute, scale
        self.q_ptr = q_ptr          # address of the q_cnt tensor it was made from (calls check they pass the same one)

    @property
    def device(self):
        return self.data.device

    def nbytes(self):
        return self.data.numel()


def content_partial(q_cnt, kv_cnt, scale, heads):
    """q_cnt (B*H, N, d) and the content pair's PackedKV -> ContentPartial (two launches on the current stream)."""
    L = _lib.load()
    dev = _need_cuda(q_cnt, kv_cnt.image)
    B, H, N, d, cc = kv_cnt.B, kv_cnt.H, kv_cnt.N, kv_cnt.d, kv_cnt.compute
    if H != heads or tuple(q_cnt.shape) != (B * H, N, d):
        raise ValueError("q_cnt must be (B*H, N, d) = %s, got %s" % ((B * H, N, d), tuple(q_cnt.shape)))
    nbytes = L.sid_content_partial_bytes(B, H, N, d, cc)
    wbytes = L.sid_content_partial_workspace_bytes(B, H, N, d, cc)
    if nbytes == 0 or wbytes == 0:
        raise _lib.SidError(-22, L.sid_last_error().decode())
    data = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ws = _workspace(dev, wbytes)
    with torch.cuda.device(dev):
        _lib.check(L.sid_content_partial(q_cnt.data_ptr(), kv_cnt.image.data_ptr(), data.data_ptr(), nbytes, B, H, N, d,
                                         float(scale), _dtype_code(q_cnt), cc, ws.data_ptr(), ws.numel(), _stream(dev)))
    return ContentPartial(data, B, H, N, d, cc, float(scale), q_cnt.data_ptr())


def content_partials(jobs, heads, streams=4):
    """ContentPartial of every (q_cnt, PackedKV, scale) in `jobs`, in order.  One call is one run per (head,
    query group), 64 workgroups at SD-v1 sizes, a quarter of the GPU, so the calls go round-robin over
    `streams` side streams (each with its own workspace) and the current stream waits for all of them."""
    if not jobs:
        return []
    dev = jobs[0][0].device
    cur = torch.cuda.current_stream(dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(max(1, min(streams, len(jobs))))]
    out = []
    for st in side:
        st.wait_stream(cur)
    for i, (q_cnt, kv, scale) in enumerate(jobs):
        with torch.cuda.stream(side[i % len(side)]):
            out.append(content_partial(q_cnt, kv, scale, heads))
    for st in side:
        cur.wait_stream(st)
    for p in out:                       # made on a side stream, used on this one from here on
        p.data.record_stream(cur)
    return out


def injected_attention_cached(x_q, q_cnt, kv1, kv2, partial, mask_signed, gamma, T, scale, heads, q_self_coef=0.5,
                              out=None, flags=0):
    """injected_attention_packed with the content pair replaced by its ContentPartial: only the two style
    blocks are computed.  `scale` and `q_cnt` must be the ones the partial was made with."""
    L = _lib.load()
    B, N, C = x_q.shape
    H = heads
    d = C // H
    dev = _need_cuda(x_q, q_cnt, mask_signed, kv1.image, kv2.image, partial.data)
    if tuple(q_cnt.shape) != (B * H, N, d) or q_cnt.dtype != x_q.dtype:
        raise ValueError("q_cnt must be (B*H, N, d) = %s of x_q's dtype" % ((B * H, N, d),))
    cc = kv1.compute
    for kv in (kv1, kv2, partial):
        if (kv.B, kv.H, kv.N, kv.d, kv.compute) != (B, H, N, d, cc):
            raise ValueError("packed K/V or content partial was made for (B,H,N,d,compute) = %s, the call has %s"
                             % ((kv.B, kv.H, kv.N, kv.d, kv.compute), (B, H, N, d, cc)))
    if partial.scale != float(scale):
        raise ValueError("the content partial was made with scale %r, the call has %r" % (partial.scale, float(scale)))
    if partial.q_ptr != q_cnt.data_ptr():
        raise ValueError("the content partial was made from another q_cnt tensor")
    if mask_signed.dtype != torch.float32 or mask_signed.numel() != B * N:
        raise ValueError("mask_signed must be float32 with B*N elements")
    if out is None:
        out = torch.empty((B, N, C), dtype=x_q.dtype, device=dev)
    nbytes = L.sid_injected_attn_cached_workspace_bytes(B, H, N, d, cc)
    if nbytes == 0:
        raise _lib.SidError(-22, L.sid_last_error().decode())
    ws = _workspace(dev, nbytes)
    with torch.cuda.device(dev):
        _lib.check(L.sid_injected_attn_fwd_cached(
            x_q.data_ptr(), q_cnt.data_ptr(), kv1.image.data_ptr(), kv2.image.data_ptr(), partial.data.data_ptr(),
            mask_signed.data_ptr(), out.data_ptr(), B, H, N, d, float(gamma), float(q_self_coef), float(T), float(scale),
            _dtype_code(x_q), cc, int(flags), ws.data_ptr(), ws.numel(), _stream(dev)))
    return out


def plain_attention(q, k, v, logit_scale, heads, q_layout="bnhd", kv_layout="bnhd", compute="f16", out=None, flags=0):
    """softmax((q k^T) * logit_scale) v.  Layout 'bnhd': (B,N,H*d); 'bhnd': (B*H,N,d).
    Returns (B, Nq, H*d)."""
    L = _lib.load()
    H = heads
    lay = {"bnhd": _lib.LAYOUT_BNHD, "bhnd": _lib.LAYOUT_BHND}
    dev = _need_cuda(q, k, v)

    def dims(t, layout):
        if layout == "bnhd":
            b, n, c = t.shape
            return b, n, c // H
        bh, n, d_ = t.shape
        return bh // H, n, d_

    B, Nq, d = dims(q, q_layout)
    Bk, Nk, dk = dims(k, kv_layout)
    if (Bk, dk) != (B, d) or tuple(v.shape) != tuple(k.shape):
        raise ValueError("q/k/v shapes disagree: %s %s %s" % (tuple(q.shape), tuple(k.shape), tuple(v.shape)))
    if not (q.dtype == k.dtype == v.dtype):
        raise TypeError("q, k, v must share one dtype")
    dt = _dtype_code(q)
    cc = _compute_code(compute)
    if out is None:
        out = torch.empty((B, Nq, H * d), dtype=q.dtype, device=dev)
    nbytes = L.sid_plain_attn_workspace_bytes(B, H, Nq, Nk, d, cc)
    if nbytes == 0:
        raise _lib.SidError(-22, L.sid_last_error().decode())
    ws = _workspace(dev, nbytes)
    with torch.cuda.device(dev):
        _lib.check(L.sid_plain_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, Nq, Nk, d,
                                        float(logit_scale), lay[q_layout], lay[kv_layout], dt, cc, int(flags),
                                        ws.data_ptr(), ws.numel(), _stream(dev)))
    return out


def mask_downsample(mask, out_h, out_w, ch=1.0):
    """(Hin, Win) float32 mask in [0,1] on the GPU -> signed (out_h*out_w,) float32."""
    L = _lib.load()
    dev = _need_cuda(mask)
    if mask.dtype != torch.float32 or mask.dim() != 2:
        raise ValueError("mask must be a 2-D float32 tensor")
    out = torch.empty(out_h * out_w, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.sid_mask_downsample(mask.data_ptr(), mask.shape[0], mask.shape[1], out_h, out_w, float(ch),
                                         out.data_ptr(), _stream(dev)))
    return out


def hf_filter(h_cnt, ratio=0.5, skip=None, h=None, out=None):
    """HF(h_cnt), or skip + h + HF(h_cnt) when skip and h are given.  (B,C,H,W) tensors."""
    L = _lib.load()
    ts = [h_cnt] + ([skip, h] if skip is not None else [])
    dev = _need_cuda(*ts)
    if (skip is None) != (h is None):
        raise ValueError("skip and h must be given together")
    for t in ts:
        if t.shape != h_cnt.shape or t.dtype != h_cnt.dtype:
            raise ValueError("skip, h and h_cnt must share shape and dtype")
    B, C, Hh, Ww = h_cnt.shape
    radius = int(min(Hh, Ww) * ratio)          # high_frequency_final.py:18, in double like the reference
    if out is None:
        out = torch.empty_like(h_cnt)
    with torch.cuda.device(dev):
        _lib.check(L.sid_hf_filter_radius(h_cnt.data_ptr(), skip.data_ptr() if skip is not None else None,
                                          h.data_ptr() if h is not None else None, out.data_ptr(),
                                          B, C, Hh, Ww, radius, _dtype_code(h_cnt), _stream(dev)))
    return out


def residual_sum(skip, h, hf, out=None):
    """(skip + h) + hf: the patched ResBlock sum with HF(h_cnt) filtered earlier (hf_filter without skip / h);
    the same bits as the fused hf_filter(h_cnt, ratio, skip, h)."""
    L = _lib.load()
    dev = _need_cuda(skip, h, hf)
    for t in (h, hf):
        if t.shape != skip.shape or t.dtype != skip.dtype:
            raise ValueError("skip, h and hf must share shape and dtype")
    if out is None:
        out = torch.empty_like(skip)
    with torch.cuda.device(dev):
        _lib.check(L.sid_residual_sum(skip.data_ptr(), h.data_ptr(), hf.data_ptr(), out.data_ptr(), skip.numel(),
                                      _dtype_code(skip), _stream(dev)))
    return out


def masked_adain(cnt, sty_char, sty_back=None, mask=None, out=None):
    """m * adain(cnt, sty_char) + (1 - m) * adain(cnt, sty_back) with m the raw (Hin, Win) float32 mask
    resized bilinearly to the latent; with sty_back and mask None: adain(cnt, sty_char).  (B,C,H,W)."""
    L = _lib.load()
    ts = [cnt, sty_char] + ([sty_back] if sty_back is not None else [])
    dev = _need_cuda(*ts)
    if (sty_back is None) != (mask is None):
        raise ValueError("sty_back and mask go together")
    for t in ts:
        if tuple(t.shape) != tuple(cnt.shape) or t.dtype != cnt.dtype or t.dim() != 4:
            raise ValueError("latents must share one (B,C,H,W) shape and dtype")
    if mask is not None:
        _need_cuda(mask)
        if mask.dtype != torch.float32 or mask.dim() != 2 or mask.device != dev:
            raise ValueError("mask must be a 2-D float32 tensor on the latents' device")
    B, C, H, W = cnt.shape
    if out is None:
        out = torch.empty_like(cnt)
    with torch.cuda.device(dev):
        _lib.check(L.sid_masked_adain(cnt.data_ptr(), sty_char.data_ptr(),
                                      sty_back.data_ptr() if sty_back is not None else None,
                                      mask.data_ptr() if mask is not None else None,
                                      mask.shape[0] if mask is not None else 0, mask.shape[1] if mask is not None else 0,
                                      out.data_ptr(), B, C, H, W, _dtype_code(cnt), _stream(dev)))
    return out
