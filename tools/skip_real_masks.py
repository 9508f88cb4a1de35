"""SID_FLAG_BLOCK_SKIP on the reference's own content masks (downsampled to 64x64, with their fractional
boundary rows): how many query groups drop a style block, per mask and feature scale.
    python tools/skip_real_masks.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "tests")]
import numpy as np, torch
from helpers import load_masks
from style_transfer_amd import ops
H, N, d = 8, 4096, 40
dev = torch.device("cuda:0")
masks = load_masks()
for sigma in (1.5, 2.0, 3.0):
    g = torch.Generator(device=dev); g.manual_seed(int(sigma * 10))
    rn = lambda *s: torch.randn(*s, device=dev, generator=g) * sigma
    xq, q_cnt = rn(1, N, H * d), rn(H, N, d)
    kv = [ops.pack_kv(rn(H, N, d), rn(H, N, d), H, "bf16") for _ in range(3)]
    for name, m512 in sorted(masks.items()):
        m = ops.mask_downsample(torch.from_numpy(m512).to(dev), 64, 64, 1.0).reshape(1, -1).contiguous()
        frac = int(((m.abs() < 1) & (m.abs() > 0)).sum().item()); zero = int((m == 0).sum().item())
        plain = ops.injected_attention_packed(xq, q_cnt, *kv, m, 0.5, 1.5, d ** -0.5, H)
        sk = ops.injected_attention_packed(xq, q_cnt, *kv, m, 0.5, 1.5, d ** -0.5, H, flags=ops.FLAG_BLOCK_SKIP)
        c = ops.skip_counts(xq, H, "bf16", packed=True)
        diff = (plain - sk).abs().max().item() / plain.abs().max().item()
        print("sigma %.1f %-28s rows +%d -%d half %d zero %d: groups %d, dead %d, content logits-only %d, diff %.1e" % (
            sigma, name, int((m > 0).sum()), int((m < 0).sum()), frac, zero, c[3], c[0], c[1], diff))
