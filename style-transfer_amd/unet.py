"""SD-v1 topology UNet that carries the injected attention: the host network around the hot path.

Everything here is stock PyTorch (convolutions, GroupNorm, LayerNorm, GEGLU feed-forward) except
the attention itself, which is `style_transfer_amd.attention.CrossAttention` (the HIP kernels).
Parameter names follow the Stable Diffusion v1 checkpoint layout (`input_blocks.4.1.
transformer_blocks.0.attn1.to_q.weight`, ...), so an SD-v1.4 state dict loads unchanged; the
configuration is `models/ldm/stable-diffusion-v1/v1-inference.yaml:29-44` of the reference.

What this module owns beyond plumbing is the per-output-block feature routing of the reference's
`UNetModel.forward` (ldm/modules/diffusionmodules/openaimodel.py:813-912): for output block i the
keys `output_block_{i}_self_attn_{q,k,v}`, `..._k_cnt`, `..._v_cnt`, `..._q_sty`, `..._q_sty2`,
`..._k_sty2`, `..._v_sty2`, `output_block_{i}_out_layers` are looked up in the step's feature
dict, `config` is handed to EVERY output block (so the 16x16 blocks 3-5 run their plain
self-attention with the temperature T), and `output_block_{i}_self_attn_s` overrides `config['T']`
in place, as the reference does (the override therefore also reaches the blocks after i).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .attention import CrossAttention

_ATTN_KEYS = (("q_injected", "q"), ("k_injected", "k"), ("v_injected", "v"), ("cnt_k_injected", "k_cnt"),
              ("sty_q_injected", "q_sty"), ("cnt_v_injected", "v_cnt"), ("sty2_q_injected", "q_sty2"),
              ("sty2_k_injected", "k_sty2"), ("sty2_v_injected", "v_sty2"))


def route_injected(injected_features, block_idx):
    """Arguments of output block `block_idx` out of one step's feature dict
    (openaimodel.py:833-883): (attention keyword dict, out_layers tensor or None)."""
    if injected_features is None:
        return {}, None
    pre = "output_block_%d_self_attn_" % block_idx
    kw = {arg: injected_features.get(pre + suffix) for arg, suffix in _ATTN_KEYS}
    cfg = injected_features["config"]
    if pre + "s" in injected_features:
        cfg["T"] = injected_features[pre + "s"]
    kw["injection_config"] = cfg
    return kw, injected_features.get("output_block_%d_out_layers" % block_idx)


_FREQS = {}


def timestep_embedding(timesteps, dim, max_period=10000):
    """[cos | sin] sinusoidal embedding (ldm/modules/diffusionmodules/util.py:151-171)."""
    half = dim // 2
    key = (dim, max_period, str(timesteps.device))
    freqs = _FREQS.get(key)
    if freqs is None:
        # computed on the CPU as the reference does, then kept on the device (no host copy per call, so an
        # evaluation can be captured in a HIP graph)
        freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(timesteps.device)
        _FREQS[key] = freqs
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


class ResBlock(nn.Module):
    """GroupNorm-SiLU-conv twice with the timestep embedding added in between.  Keeps `out_skip` and
    `out_h` of the last call: the high-frequency patch (high_frequency.py) rebuilds the block output
    from them (openaimodel.py:315-328)."""

    def __init__(self, channels, emb_channels, out_channels=None, dropout=0.0):
        super().__init__()
        out_channels = out_channels or channels
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(32, out_channels), nn.SiLU(), nn.Dropout(dropout),
                                        nn.Conv2d(out_channels, out_channels, 3, padding=1))
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)
        self.out_skip = self.out_h = None

    def forward(self, x, emb, out_layers_injected=None):
        return self._forward(x, emb, out_layers_injected)

    def _forward(self, x, emb, out_layers_injected=None):
        h = self.in_layers(x)
        if out_layers_injected is not None:
            h = torch.cat([out_layers_injected] * x.shape[0])
        else:
            h = self.out_layers(h + self.emb_layers(emb).type(h.dtype)[..., None, None])
        self.out_skip = self.skip_connection(x)
        self.out_h = h
        return self.out_skip + h


class Downsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.op = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.op(x)


class Upsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode="nearest"))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim))

    def forward(self, x):
        return self.net(x)


class BasicTransformerBlock(nn.Module):
    """Self-attention (the injected one), cross-attention on the text context, feed-forward; each with
    a pre-LayerNorm and a residual (ldm/modules/attention.py:671-689)."""
    attention_cls = CrossAttention

    def __init__(self, dim, n_heads, d_head, context_dim=None):
        super().__init__()
        self.attn1 = self.attention_cls(query_dim=dim, heads=n_heads, dim_head=d_head)
        self.ff = FeedForward(dim)
        self.attn2 = self.attention_cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    def forward(self, x, context=None, **attn_kw):
        x = self.attn1(self.norm1(x), **attn_kw) + x
        x = self.attn2(self.norm2(x), context=context) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, context_dim=None):
        super().__init__()
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, context_dim) for _ in range(depth)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, context=None, **attn_kw):
        b, c, h, w = x.shape
        y = self.proj_in(self.norm(x)).reshape(b, -1, h * w).transpose(1, 2)       # b (h w) c
        for blk in self.transformer_blocks:
            y = blk(y, context=context, **attn_kw)
        y = y.transpose(1, 2).reshape(b, -1, h, w)
        return self.proj_out(y) + x


class Stage(nn.Sequential):
    """One UNet block: its layers in order, each given the arguments it takes."""

    def forward(self, x, emb, context=None, out_layers_injected=None, **attn_kw):
        for layer in self:
            if isinstance(layer, ResBlock):
                x = layer(x, emb, out_layers_injected=out_layers_injected)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context, **attn_kw)
            else:
                x = layer(x)
        return x


class UNetModel(nn.Module):
    """Defaults are the SD-v1 configuration: 320 channels, multipliers (1,2,4,4), two ResBlocks per
    level, transformers at downsampling 1, 2 and 4 with 8 heads, 768-wide text context."""

    def __init__(self, in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2,
                 attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768,
                 transformer_depth=1):
        super().__init__()
        self.model_channels = model_channels
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))

        def transformer(ch):
            return SpatialTransformer(ch, num_heads, ch // num_heads, transformer_depth, context_dim)

        self.input_blocks = nn.ModuleList([Stage(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(transformer(ch))
                self.input_blocks.append(Stage(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(Stage(Downsample(ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = Stage(ResBlock(ch, ted), transformer(ch), ResBlock(ch, ted))
        self.output_blocks = nn.ModuleList()
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [ResBlock(ch + chans.pop(), ted, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(transformer(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch))
                    ds //= 2
                self.output_blocks.append(Stage(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))

    def attention_modules(self):
        return [m for m in self.modules() if isinstance(m, BasicTransformerBlock.attention_cls)]

    def set_content_name(self, cnt_name):
        """What run_ori.py:159-161 does per image: the attention derives the mask path from it."""
        for m in self.attention_modules():
            m.cnt_name = cnt_name

    def forward(self, x, timesteps=None, context=None, injected_features=None):
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels))
        hs, h = [], x
        for blk in self.input_blocks:
            h = blk(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        for i, blk in enumerate(self.output_blocks):
            kw, out_layers = route_injected(injected_features, i)
            h = blk(torch.cat([h, hs.pop()], dim=1), emb, context, out_layers_injected=out_layers, **kw)
        return self.out(h)
