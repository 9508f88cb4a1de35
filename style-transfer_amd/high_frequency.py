"""High-frequency injection: filter, schedule and decoder ResBlock patch.

Same names and arguments as the reference module high_frequency_final.py; the arithmetic of
`high_freq_filter` and of the patched ResBlock sum runs in the sid_hf_filter kernel.
"""
import types

import torch

from . import ops

_KERNEL_SIZES = (8, 16, 32, 64, 128)


def _kernel_ok(t):
    return (t.dim() == 4 and t.is_cuda and t.shape[-2] in _KERNEL_SIZES and t.shape[-1] in _KERNEL_SIZES)


def high_freq_filter(h, radius_ratio=0.5):
    """Zero the centred low-frequency disc of radius int(min(H,W)*radius_ratio) of the 2-D spectrum
    of a (B,C,H,W) map and return the real part of the inverse transform, in h's dtype
    (high_frequency_final.py:4-36)."""
    if not _kernel_ok(h):
        raise RuntimeError("high_freq_filter: needs a (B,C,H,W) GPU tensor with H, W powers of two in [8,128]; "
                           "got %s on %s (there is no fallback path)" % (tuple(h.shape), h.device))
    if h.dtype in (torch.float32, torch.bfloat16):
        return ops.hf_filter(h.contiguous(), radius_ratio)
    return ops.hf_filter(h.to(torch.float32).contiguous(), radius_ratio).to(h.dtype)


def make_content_injection_schedule(ddim_timesteps, alpha=0.4):
    """The first int(alpha*T) entries of the (ascending) DDIM timestep list
    (high_frequency_final.py:38-43)."""
    T = len(ddim_timesteps)
    return ddim_timesteps[:int(alpha * T)]


def resblock_hf_sum(out_skip, out_h, h_cnt, ratio=0.5):
    """out_skip + out_h + HF(h_cnt) in one kernel (high_frequency_final.py:82-88)."""
    h_cnt = h_cnt.to(device=out_skip.device, dtype=out_skip.dtype)
    if not _kernel_ok(h_cnt) or out_skip.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("resblock_hf_sum: unsupported map %s %s" % (tuple(h_cnt.shape), h_cnt.dtype))
    return ops.hf_filter(h_cnt.contiguous(), ratio, skip=out_skip.contiguous(), h=out_h.contiguous())


def resblock_cached_hf_sum(out_skip, out_h, h_cnt, ratio, cache, key):
    """resblock_hf_sum with HF(h_cnt) kept in `cache[key]`: filtered at the first use, added by
    sid_residual_sum from then on (the same bits: the fused kernel adds in the same order)."""
    hf = cache.get(key)
    if hf is None or hf.dtype != out_skip.dtype or hf.device != out_skip.device:
        hf = cache[key] = high_freq_filter(h_cnt.to(device=out_skip.device, dtype=out_skip.dtype), ratio)
    return ops.residual_sum(out_skip.contiguous(), out_h.contiguous(), hf)


def patch_decoder_resblocks_h_and_cnt_hf(unet, schedule, residuals_all, ratio=0.5, hf_cache=None):
    """Rebind `_forward` of the last ResBlock of output blocks 6..11 so that, on scheduled
    timesteps with a stored content residual, it returns out_skip + out_h + HF(h_cnt)
    (high_frequency_final.py:46-92).  The block must expose `out_skip` / `out_h` after its
    original `_forward`, and the sampler sets `ri_timestep` on it.
    hf_cache: a dict the caller keeps per (content image, ratio).  HF(h_cnt) depends on nothing else, so
    with it every (timestep, block) is filtered once and reused for every style pair of that content."""
    for block_id in range(6, 12):
        if block_id >= len(unet.output_blocks):
            break
        for module in reversed(unet.output_blocks[block_id]):
            if not module.__class__.__name__.endswith("ResBlock"):
                continue
            module.block_id = block_id
            # patching again (next image, other residuals) replaces the previous patch, never nests
            inner = getattr(module, "_unpatched_forward", None) or module._forward
            module._unpatched_forward = inner

            def patched(self, x, emb, out_layers_injected=None, _inner=inner):
                out = _inner(x, emb, out_layers_injected)
                t = getattr(self, "ri_timestep", None)
                if t in schedule and t in residuals_all:
                    h_cnt = residuals_all[t].get("output_block_%d_cnt_h" % self.block_id, None)
                    if h_cnt is not None and hf_cache is not None and _kernel_ok(h_cnt.to(self.out_skip.device)):
                        out = resblock_cached_hf_sum(self.out_skip, self.out_h, h_cnt, ratio, hf_cache, (t, self.block_id))
                    elif h_cnt is not None:
                        out = resblock_hf_sum(self.out_skip, self.out_h, h_cnt, ratio)
                return out

            module._forward = types.MethodType(patched, module)
            break


def unpatch_decoder_resblocks(unet):
    """Undo patch_decoder_resblocks_h_and_cnt_hf (the reference has no such call: it patches once per
    process; a sweep over images with and without residual files needs it)."""
    for block in unet.output_blocks:
        for module in block:
            inner = getattr(module, "_unpatched_forward", None)
            if inner is not None:
                module._forward = inner
                module._unpatched_forward = None
                module.ri_timestep = None
