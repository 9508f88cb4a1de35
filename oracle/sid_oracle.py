"""CPU restatement (numpy) of the mask-based multi-style injection path.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, and only as the
checker.  The product path is the HIP library and fails loudly without it.

Each function restates one piece of the reference (paths relative to the reference
repository root) and is pinned by tests/golden/*.npz, which were produced by running the
reference's own modules (tests/golden/make_goldens.py).

Array conventions: H heads, N tokens, d head_dim; per-head tensors are (H, N, d) float32.
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------
# A4  mask threshold + bilinear downsample      ldm/modules/attention.py:270-284
# --------------------------------------------------------------------------------------
def threshold_mask(mask, ch=1.0):
    """`mask[mask<0.5]=-1; mask[mask>0.5]=1; mask*=ch` (ldm/modules/attention.py:270-273).

    A value of exactly 0.5 is left at 0.5 by both comparisons.
    """
    m = np.array(mask, dtype=F32, copy=True)
    lo = m < F32(0.5)
    hi = m > F32(0.5)
    m[lo] = F32(-1.0)
    m[hi] = F32(1.0)
    return (m * F32(ch)).astype(F32)


def _linear_taps(in_size, out_size):
    """Source taps of torch's bilinear resize with align_corners=False, antialias off.

    ldm/modules/attention.py:281-282 calls F.interpolate(mode='bilinear',
    align_corners=False); its float32 arithmetic is: scale = in/out,
    src = scale*(dst+0.5)-0.5 clamped at 0, i0 = int(src), i1 = i0 + (i0 < in-1),
    w1 = src - i0, w0 = 1 - w1.
    """
    scale = F32(in_size) / F32(out_size)
    dst = np.arange(out_size, dtype=F32)
    src = (scale * (dst + F32(0.5)) - F32(0.5)).astype(F32)
    src = np.maximum(src, F32(0.0))
    i0 = src.astype(np.int64)
    i0 = np.minimum(i0, in_size - 1)
    i1 = i0 + (i0 < in_size - 1)
    w1 = (src - i0.astype(F32)).astype(F32)
    w1 = np.clip(w1, F32(0.0), F32(1.0))
    w0 = (F32(1.0) - w1).astype(F32)
    return i0, i1, w0, w1


def bilinear_resize(img, out_h, out_w):
    """float32 bilinear resize of a 2-D array, same operation order as torch's CPU kernel:
    out = w0h*(w0w*a + w1w*b) + w1h*(w0w*c + w1w*d), every product and sum rounded to f32."""
    img = np.asarray(img, dtype=F32)
    ih, iw = img.shape
    y0, y1, wy0, wy1 = _linear_taps(ih, out_h)
    x0, x1, wx0, wx1 = _linear_taps(iw, out_w)
    top = (wx0[None, :] * img[y0][:, x0]).astype(F32) + (wx1[None, :] * img[y0][:, x1]).astype(F32)
    bot = (wx0[None, :] * img[y1][:, x0]).astype(F32) + (wx1[None, :] * img[y1][:, x1]).astype(F32)
    return ((wy0[:, None] * top.astype(F32)).astype(F32) + (wy1[:, None] * bot.astype(F32)).astype(F32)).astype(F32)


def mask_downsample(mask, out_h, out_w, ch=1.0):
    """Signed per-token mask m (out_h*out_w,), +ch on the character region, -ch elsewhere
    (ldm/modules/attention.py:270-284): threshold, multiply by ch, bilinear resize."""
    return bilinear_resize(threshold_mask(mask, ch), out_h, out_w).reshape(-1)


def mask_downsample_pow2(mask, r, ch=1.0):
    """Closed form for a square 2^k mask reduced to r x r: the mean of the 2x2 centre pixels
    of each cell, exact in float32 (equals mask_downsample bit for bit)."""
    m = threshold_mask(mask, ch)
    k = m.shape[0] // r
    c = k // 2
    if k == 1:
        return m.reshape(-1)
    s = m[c - 1::k, c - 1::k] + m[c - 1::k, c::k] + m[c::k, c - 1::k] + m[c::k, c::k]
    return (F32(0.25) * s).astype(F32).reshape(-1)


# --------------------------------------------------------------------------------------
# A1-A5  injected three-block attention        ldm/modules/attention.py:370-546
# --------------------------------------------------------------------------------------
def split_heads(x, heads):
    """'b n (h d) -> (b h) n d' for b = 1 (ldm/modules/attention.py:372,378)."""
    n, c = x.shape[-2], x.shape[-1]
    return np.ascontiguousarray(x.reshape(n, heads, c // heads).transpose(1, 0, 2))


def merge_heads(o):
    """'h (b n) d -> b n (h d)' for b = 1 (ldm/modules/attention.py:546)."""
    h, n, d = o.shape
    return np.ascontiguousarray(o.transpose(1, 0, 2).reshape(n, h * d))


def _softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def injected_attention(xq, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt, m, gamma, T, scale,
                       q_self_coef=0.5, dtype=F32, return_stats=False):
    """One softmax over [char-style | back-style | content] keys with a per-row mask bias.

    xq    (N, H*d)  query projection of the current latent, to_q(x)      attention.py:377
    q_cnt (H, N, d) content query from the feature list                  attention.py:375
    k1,v1 char style; k2,v2 back style; k_cnt,v_cnt content              attention.py:385-438
    m     (N,)      signed mask, +1 character ... -1 background (ch=+1)
    Returns (N, H*d), the tensor handed to to_out                        attention.py:546

    q      = q_cnt*gamma + split_heads(xq)*q_self_coef                   attention.py:381
    CC     = (q_cnt @ k_cnt^T) * scale                                   attention.py:206,440
    R1,R2  = q @ k1^T, q @ k2^T (raw)                                    attention.py:218,236
    bias_b = (min_j CC - max_j R_b) * (m * ch_b), ch_1=-1, ch_2=+1       attention.py:239-241,273,285-288
    L      = cat[(R1*T)*scale + bias_1, (R2*T)*scale + bias_2, CC]       attention.py:220-221,538
    out    = softmax(L) @ cat[v1, v2, v_cnt]                             attention.py:540-544
    """
    H, N, d = q_cnt.shape
    cast = lambda a: np.asarray(a, dtype=dtype)
    xq, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt = map(cast, (xq, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt))
    m = cast(m).reshape(1, N, 1)
    gamma, T, scale, q_self_coef = dtype(gamma), dtype(T), dtype(scale), dtype(q_self_coef)
    q_ = split_heads(xq, H)
    q = q_cnt * gamma + q_ * q_self_coef
    cc = np.einsum("hid,hjd->hij", q_cnt, k_cnt) * scale
    min_cc = cc.min(axis=-1, keepdims=True)
    outs = np.empty((H, N, d), dtype=dtype)
    r1 = np.einsum("hid,hjd->hij", q, k1)
    max1 = r1.max(axis=-1, keepdims=True)
    s1 = r1 * T
    s1 *= scale
    s1 += (min_cc - max1) * (m * dtype(-1.0))
    del r1
    r2 = np.einsum("hid,hjd->hij", q, k2)
    max2 = r2.max(axis=-1, keepdims=True)
    s2 = r2 * T
    s2 *= scale
    s2 += (min_cc - max2) * (m * dtype(1.0))
    del r2
    for h in range(H):
        logits = np.concatenate([s1[h], s2[h], cc[h]], axis=1)
        p = _softmax(logits)
        outs[h] = p @ np.concatenate([v1[h], v2[h], v_cnt[h]], axis=0)
    out = merge_heads(outs)
    if return_stats:
        return out, dict(min_cc=min_cc[..., 0], max1=max1[..., 0], max2=max2[..., 0])
    return out


def injected_attention_rows(rows, xq, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt, m, gamma, T, scale,
                            q_self_coef=0.5, dtype=np.float64):
    """injected_attention restricted to the query rows `rows` (all keys); (len(rows), H*d).
    Lets a test check sampled rows of a large problem without the N x 3N logits."""
    return _rows_impl(np.asarray(rows), xq, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt, m, gamma, T,
                      scale, q_self_coef, dtype)


def _rows_impl(rows, xq, q_cnt, k1, v1, k2, v2, k_cnt, v_cnt, m, gamma, T, scale, q_self_coef, dtype):
    H, N, d = q_cnt.shape
    cast = lambda a: np.asarray(a, dtype=dtype)
    k1, v1, k2, v2, k_cnt, v_cnt = map(cast, (k1, v1, k2, v2, k_cnt, v_cnt))
    qc = cast(np.asarray(q_cnt)[:, rows])
    q_ = split_heads(cast(np.asarray(xq)[rows]), H)
    mm = cast(np.asarray(m).reshape(-1)[rows]).reshape(1, -1, 1)
    q = qc * dtype(gamma) + q_ * dtype(q_self_coef)
    cc = np.einsum("hid,hjd->hij", qc, k_cnt) * dtype(scale)
    min_cc = cc.min(axis=-1, keepdims=True)
    r1 = np.einsum("hid,hjd->hij", q, k1)
    r2 = np.einsum("hid,hjd->hij", q, k2)
    s1 = r1 * dtype(T) * dtype(scale) + (min_cc - r1.max(-1, keepdims=True)) * (mm * dtype(-1.0))
    s2 = r2 * dtype(T) * dtype(scale) + (min_cc - r2.max(-1, keepdims=True)) * (mm * dtype(1.0))
    p = _softmax(np.concatenate([s1, s2, cc], axis=2))
    o = np.einsum("hij,hjd->hid", p, np.concatenate([v1, v2, v_cnt], axis=1))
    return merge_heads(o)


# --------------------------------------------------------------------------------------
# A7-A9  single-block attention                 ldm/modules/attention.py:556-604
# --------------------------------------------------------------------------------------
def plain_attention(q, k, v, scale, T=1.0, dtype=F32):
    """softmax(((q k^T) * T) * scale) v, merged heads.  q (H,Nq,d), k,v (H,Nk,d) -> (Nq, H*d).

    T is the temperature of the mask-present non-injected branch (attention.py:557-560),
    of the no-mask StyleID branch (attention.py:570-575) and 1 for the vanilla branch
    (attention.py:586-602).
    """
    q, k, v = (np.asarray(a, dtype=dtype) for a in (q, k, v))
    sim = np.einsum("hid,hjd->hij", q, k)
    sim *= dtype(T)
    sim *= dtype(scale)
    return merge_heads(np.einsum("hij,hjd->hid", _softmax(sim), v))


def plain_attention_rows(rows, q, k, v, scale, T=1.0, dtype=np.float64):
    """plain_attention restricted to the query rows `rows` (all keys); (len(rows), H*d)."""
    q, k, v = (np.asarray(a, dtype=dtype) for a in (np.asarray(q)[:, np.asarray(rows)], k, v))
    sim = np.einsum("hid,hjd->hij", q, k) * dtype(T) * dtype(scale)
    return merge_heads(np.einsum("hij,hjd->hid", _softmax(sim), v))


def blend_query(xq, q_cnt, gamma, q_self_coef=0.5):
    """q = q_cnt*gamma + split_heads(to_q(x))*0.5; the 0.5 is a literal in the reference
    (ldm/modules/attention.py:381)."""
    H = q_cnt.shape[0]
    return (np.asarray(q_cnt, F32) * F32(gamma) + split_heads(np.asarray(xq, F32), H) * F32(q_self_coef)).astype(F32)


# --------------------------------------------------------------------------------------
# A10-A12  high-frequency radius filter         high_frequency_final.py:4-43,62-89
# --------------------------------------------------------------------------------------
def hf_keep_mask(H, W, ratio):
    """(H, W) float32 keep-mask in UNSHIFTED bin order.

    high_frequency_final.py:17-25 zeroes shifted bins with (y-H//2)^2+(x-W//2)^2 <
    int(min(H,W)*ratio)^2; shifted bin y holds signed frequency y-H//2, so in unshifted
    order bin k has f = k if k < ceil(H/2) else k-H.
    """
    radius = int(min(H, W) * ratio)
    fy = np.fft.ifftshift(np.arange(H) - H // 2)
    fx = np.fft.ifftshift(np.arange(W) - W // 2)
    dist = fy[:, None] ** 2 + fx[None, :] ** 2
    return (dist >= radius ** 2).astype(F32)


def high_freq_filter(h, ratio=0.5):
    """fft2(ortho) -> fftshift -> zero the disc -> ifftshift -> ifft2(ortho) -> real
    (high_frequency_final.py:4-36), restated literally with numpy on (B, C, H, W)."""
    h = np.asarray(h)
    orig = h.dtype
    x = h.astype(F32)
    B, C, H, W = x.shape
    f = np.fft.fft2(x.astype(np.float64), norm="ortho")
    fs = np.fft.fftshift(f)                       # no axes given: rolls all four dims (line 14)
    cy, cx = H // 2, W // 2
    radius = int(min(H, W) * ratio)
    y = np.arange(H).reshape(-1, 1)
    xx = np.arange(W).reshape(1, -1)
    dist = (y - cy) ** 2 + (xx - cx) ** 2
    mask = np.ones((H, W))
    mask[dist < radius ** 2] = 0
    fs = fs * mask[None, None]
    f2 = np.fft.ifftshift(fs)
    out = np.fft.ifft2(f2, norm="ortho")
    return out.real.astype(F32).astype(orig)


def high_freq_filter_unshifted(h, ratio=0.5):
    """Shift-free form: ifft2(fft2(h) * keep), keep from hf_keep_mask.  Same result."""
    x = np.asarray(h, dtype=F32).astype(np.float64)
    H, W = x.shape[-2:]
    keep = hf_keep_mask(H, W, ratio)
    return np.fft.ifft2(np.fft.fft2(x, norm="ortho") * keep, norm="ortho").real.astype(F32)


def resblock_hf_merge(skip, h, h_cnt, ratio=0.5):
    """out = skip + h + HF(h_cnt)  (high_frequency_final.py:82-88), left-to-right in f32."""
    hf = high_freq_filter(h_cnt, ratio)
    return ((np.asarray(skip, F32) + np.asarray(h, F32)).astype(F32) + hf).astype(F32)


def make_content_injection_schedule(ddim_timesteps, alpha=0.4):
    """First int(alpha*T) entries of the ascending timestep list (high_frequency_final.py:38-43)."""
    T = len(ddim_timesteps)
    return ddim_timesteps[: int(alpha * T)]


def make_ddim_timesteps_uniform(num_ddim, num_ddpm=1000):
    """ldm/modules/diffusionmodules/util.py:46-60, 'uniform' method: arange(num_ddim)*c + 1."""
    c = num_ddpm // num_ddim
    return (np.arange(0, num_ddim) * c).astype(int) + 1


# --------------------------------------------------------------------------------------
# Masked AdaIN initial latent                      run_ori.py:77-82, 228-232
# --------------------------------------------------------------------------------------
def adain(cnt, sty):
    """((cnt - mean_c)/std_c)*std_s + mean_s, statistics per channel over (batch, H, W), std unbiased
    (torch.std default)                                                   run_ori.py:77-82"""
    c, s = np.asarray(cnt, F32), np.asarray(sty, F32)
    ax = (0, 2, 3)
    cm, sm_ = c.mean(ax, keepdims=True, dtype=np.float64), s.mean(ax, keepdims=True, dtype=np.float64)
    cs, ss = c.std(ax, keepdims=True, ddof=1, dtype=np.float64), s.std(ax, keepdims=True, ddof=1, dtype=np.float64)
    cm, sm_, cs, ss = (a.astype(F32) for a in (cm, sm_, cs, ss))
    return ((((c - cm).astype(F32) / cs).astype(F32) * ss).astype(F32) + sm_).astype(F32)


def masked_adain_init(cnt, char, back, mask):
    """m*adain(cnt,char) + (1-m)*adain(cnt,back), m = bilinear resize of the RAW mask (no threshold)
    to the latent size                                                    run_ori.py:228-232"""
    H, W = cnt.shape[2], cnt.shape[3]
    m = bilinear_resize(np.asarray(mask, F32), H, W)[None, None]
    a, b = adain(cnt, char), adain(cnt, back)
    return ((m * a).astype(F32) + ((F32(1.0) - m).astype(F32) * b).astype(F32)).astype(F32)
