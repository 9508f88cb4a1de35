"""Drop-in `CrossAttention` for the SD-v1.4 UNet with the injected-features call signature.

Same constructor, attributes and `forward` keywords as the reference module
(ldm/modules/attention.py:161-199, :332-346), so `BasicTransformerBlock` can call it unchanged
(:671-686).  The projections to_q/to_k/to_v/to_out stay torch Linear layers; everything between
them runs in libsid.so:

  branch of the reference forward                      kernel
  -------------------------------------------------   ---------------------------------------
  mask file present, q/k injected      (:425-546)      sid_injected_attn_fwd (3 key blocks + bias)
  mask file present, nothing injected  (:556-566)      sid_plain_attn_fwd, logit scale T*scale
  no mask file, or cross-attention     (:569-579)      sid_plain_attn_fwd, T only if q or k injected
  gen_pkl (feature extraction)         (:584-604)      torch (it must expose qk_sim / attn)
"""
import math
import os
import warnings

import torch
from torch import nn

from . import ops
from .mask import default_cache, mask_path_for

_default_compute = "f16"


def set_default_compute(mode):
    """'f16' (the default): half-precision query/key operands in the logits product, bf16 probabilities and
    values (<= 1e-2 rel. also at feature-scale magnitudes); 'bf16': bf16 operands throughout (<= 2e-2 at
    unit-variance features, up to 3.2e-2 at three times that; 3 % faster on the longest layers);
    'fp32': split-bf16 f32-grade products (<= 1e-3)."""
    global _default_compute
    if mode not in ("bf16", "f16", "fp32"):
        raise ValueError("compute mode must be 'bf16', 'f16' or 'fp32'")
    _default_compute = mode


def resolve_compute(mode, n_keys, head_dim, logit_scale):
    """The arithmetic a call runs in: 'f16' falls back to 'bf16' where SID_COMPUTE_F16QK refuses the call
    (include/sid.h: more than 256 keys and |scale * log2(e)| outside [2^-10, 2^10]: the long-sequence
    kernels keep the scaled queries in half precision)."""
    if mode != "f16":
        return mode
    c2 = abs(logit_scale) * 1.4426950408889634
    served = n_keys <= 256 or 2.0 ** -10 <= c2 <= 2.0 ** 10
    return "f16" if served else "bf16"


def resolve_injected_compute(mode, n_keys, head_dim, T, scale):
    """resolve_compute for the three-block call: its style blocks scale by T * scale, its content block by scale."""
    mode = resolve_compute(mode, n_keys, head_dim, scale)
    return resolve_compute(mode, n_keys, head_dim, T * scale)


class CrossAttention(nn.Module):
    _warned_bypass = False

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))
        self.attn = None
        self.k = None
        self.v = None
        self.qk_sim = None
        self.gen_pkl = False
        self.sty_name = None
        self.cnt_name = None
        self.target_t_list = None
        self.layer_id = None
        self.compute = None          # None: follow set_default_compute()
        self.kernel_flags = 0        # e.g. ops.FLAG_THROUGHPUT when several images share the GPU
        # ops.FLAG_BLOCK_SKIP pays only where the logits are large enough for blocks to die.  "auto": after the
        # layer's next flagged call the counters are read once (one synchronisation) and the flag is dropped
        # for this layer unless at least half of the query groups lost a style block; "fixed": leave it alone.
        self.skip_policy = "fixed"
        self.mask_cache = default_cache
        self._q = None
        self._q_lazy = None

    # `q` is the (blended) query in (b*heads, n, d) layout, as the reference stores it (:404).
    # On the injected branch the blend happens inside the kernel, so the tensor is only
    # materialised if somebody reads the attribute.
    @property
    def q(self):
        if self._q is None and self._q_lazy is not None:
            self._q = self._q_lazy()
            self._q_lazy = None
        return self._q

    @q.setter
    def q(self, value):
        self._q = value
        self._q_lazy = None

    def _split(self, t):
        b, n, c = t.shape
        h = self.heads
        return t.reshape(b, n, h, c // h).permute(0, 2, 1, 3).reshape(b * h, n, c // h)

    def forward(self, x, context=None, mask=None, q_injected=None, k_injected=None, v_injected=None,
                cnt_k_injected=None, sty_q_injected=None, cnt_v_injected=None, sty2_q_injected=None,
                sty2_k_injected=None, sty2_v_injected=None, injection_config=None):
        self.attn = None
        h = self.heads
        b = x.shape[0]
        compute = self.compute or _default_compute
        T = 1.0
        gamma = 0.
        is_cross = context is not None
        if injection_config is not None:
            T = injection_config['T']
            gamma = injection_config['gamma']

        xq = self.to_q(x)                                  # (b, n, h*d)
        context = x if context is None else context

        if self.gen_pkl:
            return self._forward_gen_pkl(xq, context, mask)

        base_name, _ = os.path.splitext(self.cnt_name)     # TypeError when cnt_name is None (:410)
        mask_path = base_name + "_mask.npy"
        use_mask = (self.cnt_name is not None) and (not is_cross) and os.path.exists(mask_path)

        if use_mask and q_injected is not None and k_injected is not None:
            if b != 1:
                raise ValueError("injected attention is defined for batch size 1 only: the reference folds the "
                                 "batch into the token axis (ldm/modules/attention.py:203-204, :226)")
            n = x.shape[1]
            side = int(math.sqrt(n))
            if side * side != n:
                raise ValueError("token count %d is not a square grid" % n)
            m = self.mask_cache.signed_mask(mask_path, side, side, x.device)
            if isinstance(k_injected, ops.PackedKV):
                # the feature store keeps the key/value pairs in the kernels' operand layout
                # (FeatureStore.pack): one PackedKV stands for a block's k and v
                q_cnt = self._as(q_injected, xq)
                self.k, self.v = k_injected, v_injected
                self._q, self._q_lazy = None, (lambda: q_cnt * gamma + self._split(xq) * (1. - 0.5))
                partial = getattr(cnt_k_injected, "partial", None)
                if partial is not None and partial.q_ptr != q_cnt.data_ptr() and not CrossAttention._warned_bypass:
                    CrossAttention._warned_bypass = True
                    warnings.warn("a cached content block is not used: the call's content query is not the tensor the block "
                                  "was made from (a dtype or device copy: store the features in the model's dtype); "
                                  "computing all three blocks")
                if partial is not None and partial.q_ptr == q_cnt.data_ptr() and partial.scale == float(self.scale):
                    # the store computed the content block of this (timestep, layer) when it was loaded
                    # (FeatureStore.pack(cache_content=True)): only the two style blocks are left
                    out = ops.injected_attention_cached(xq.contiguous(), q_cnt, k_injected, sty2_k_injected, partial, m,
                                                        gamma, T, self.scale, self.heads, q_self_coef=0.5,
                                                        flags=self.kernel_flags)
                    self._review_block_skip(xq, k_injected.compute, "cached", T)
                    return self.to_out(out)
                out = ops.injected_attention_packed(xq.contiguous(), q_cnt, k_injected, sty2_k_injected, cnt_k_injected, m,
                                                    gamma, T, self.scale, self.heads, q_self_coef=0.5,
                                                    flags=self.kernel_flags)
                self._review_block_skip(xq, k_injected.compute, True, T)
                return self.to_out(out)
            feats = [q_injected, k_injected, v_injected, sty2_k_injected, sty2_v_injected, cnt_k_injected,
                     cnt_v_injected]
            feats = [self._as(f, xq) for f in feats]
            self.k, self.v = feats[1], feats[2]
            q_cnt = feats[0]
            self._q, self._q_lazy = None, (lambda: q_cnt * gamma + self._split(xq) * (1. - 0.5))
            out = self._injected(xq.contiguous(), feats, m, gamma, T, compute)
            return self.to_out(out)

        # single-block attention
        if isinstance(k_injected, ops.PackedKV):
            raise TypeError("pre-packed key/value features serve the mask-based injected attention only; this call "
                            "takes the single-block path (no mask file for %r): load the feature store unpacked"
                            % (self.cnt_name,))
        if q_injected is None:
            q, q_layout = xq, "bnhd"
            self._q, self._q_lazy = None, (lambda: self._split(xq))
        else:
            q = torch.cat([q_injected] * b) * gamma + self._split(xq) * (1. - 0.5)     # (:376-381)
            q_layout = "bhnd"
            self.q = q
        if k_injected is None:
            k, v = self.to_k(context), self.to_v(context)
            kv_layout = "bnhd"
            self.k, self.v = None, None
        else:
            k = torch.cat([k_injected] * b, dim=0)
            v = torch.cat([v_injected] * b, dim=0)
            kv_layout = "bhnd"
            self.k, self.v = k, v
        if use_mask:
            logit_scale = T * self.scale                   # (:557-559)
        else:
            logit_scale = (T if (q_injected is not None or k_injected is not None) else 1.0) * self.scale
        out = self._plain(q.contiguous(), self._as(k, q), self._as(v, q), logit_scale, q_layout, kv_layout, compute)
        return self.to_out(out)

    # The two kernel calls of the module.  (A measurement harness may subclass and override them to
    # time another implementation inside the same network; the package itself has no other path.)
    def _injected(self, xq, feats, m, gamma, T, compute):
        d = xq.shape[-1] // self.heads
        compute = resolve_injected_compute(compute, xq.shape[1], d, T, self.scale)
        out = ops.injected_attention(xq, *feats, m, gamma, T, self.scale, self.heads, q_self_coef=0.5, compute=compute,
                                     flags=self.kernel_flags)
        self._review_block_skip(xq, compute, False, T)
        return out

    def _review_block_skip(self, xq, compute, packed, T):
        """skip_policy "auto": keep ops.FLAG_BLOCK_SKIP on this layer only if the call just made dropped a style
        block for at least half of its query groups (where nothing can be dropped the flag costs about a
        fifth of the call).  Decided once per arming (run.py arms it per image); never during graph capture."""
        if self.skip_policy != "auto" or not (self.kernel_flags & ops.FLAG_BLOCK_SKIP):
            return
        if torch.cuda.is_current_stream_capturing():
            return
        counts = ops.skip_counts(xq, self.heads, compute, self.kernel_flags, packed, T=T, scale=self.scale)
        if counts is None or counts[3] == 0 or 2 * counts[0] < counts[3]:
            self.kernel_flags &= ~ops.FLAG_BLOCK_SKIP
        self.skip_policy = "fixed"

    def _plain(self, q, k, v, logit_scale, q_layout, kv_layout, compute):
        n_keys = k.shape[1]
        d = (k.shape[-1] // self.heads) if kv_layout == "bnhd" else k.shape[-1]
        compute = resolve_compute(compute, n_keys, d, logit_scale)
        return ops.plain_attention(q, k, v, logit_scale, self.heads, q_layout=q_layout, kv_layout=kv_layout,
                                   compute=compute, flags=self.kernel_flags)

    @staticmethod
    def _as(t, like):
        if t.dtype != like.dtype or t.device != like.device:
            t = t.to(device=like.device, dtype=like.dtype)
        return t.contiguous()

    def _forward_gen_pkl(self, xq, context, mask):
        """Feature-extraction mode (ldm/modules/attention.py:584-604): vanilla attention that
        leaves q, k, v, qk_sim and attn on the module for the DDIM-inversion callback to copy."""
        h = self.heads
        q = self._split(xq)
        k = self._split(self.to_k(context))
        v = self._split(self.to_v(context))
        self.q, self.k, self.v = q, k, v
        sim = torch.einsum('bid,bjd->bij', q, k) * self.scale
        self.qk_sim = sim
        if mask is not None:
            keep = mask.reshape(mask.shape[0], -1)
            keep = keep[:, None, :].repeat_interleave(h, dim=0)
            sim.masked_fill_(~keep, -torch.finfo(sim.dtype).max)
        attn = sim.softmax(dim=-1)
        self.attn = attn
        out = torch.einsum('bij,bjd->bid', attn, v)
        bh, n, d = out.shape
        out = out.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, h * d)
        return self.to_out(out)
