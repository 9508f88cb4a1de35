"""End-to-end stylisation timing with the host UNet (SD-v1 topology, random weights: no checkpoint
is available offline) on synthetic feature lists in the precomputed_feats_k layout:

    python tools/e2e.py [--steps 50] [--images 2] [--compute bf16] [--hf]

Runs run_ori.py's per-image sequence — masked AdaIN init, feat_merge_2sty, 49 UNet evaluations under
DDIM — twice: with the HIP attention kernels and with the eager PyTorch attention in the same
network, and reports images/s of both and the PSNR between their final latents."""
import argparse
import json
import os
import sys
import tempfile
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

from style_transfer_amd import synth, unet as U
from style_transfer_amd.features import feat_merge_2sty, masked_adain_init
from style_transfer_amd.sampler import DDIMSampler
from torch_reference import TorchCrossAttention
from style_transfer_amd.attention import CrossAttention


def build(attn_cls, dev):
    U.BasicTransformerBlock.attention_cls = attn_cls
    try:
        torch.manual_seed(0)
        m = U.UNetModel().eval().to(dev)
    finally:
        U.BasicTransformerBlock.attention_cls = CrossAttention
    return m


def feature_list(dev, seed, steps):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    out = []
    for _ in range(steps):
        e = {"config": {"T": 1.5}, "z_enc": torch.randn(1, 4, 64, 64, device=dev, generator=g)}
        for b in range(6, 12):
            n, d = (1024, 80) if b < 9 else (4096, 40)
            for t in "qkv":
                e["output_block_%d_self_attn_%s" % (b, t)] = torch.randn(8, n, d, device=dev, generator=g)
        out.append(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--compute", default="bf16")
    ap.add_argument("--skip-torch", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cnt, char, back = (feature_list(dev, s, a.steps) for s in (1, 2, 3))
    opt = types.SimpleNamespace(gamma=0.5, T=1.5)
    merged = feat_merge_2sty(opt, cnt, char, back, start_step=a.steps - 1) if a.steps == 50 else None
    if merged is None:       # shorter runs: same merge, list padded to the 50 entries feat_merge_2sty walks
        pad = lambda f: f + [f[-1]] * (50 - len(f))
        merged = feat_merge_2sty(opt, pad(cnt), pad(char), pad(back), start_step=49)[:a.steps]
    mask = synth.disc_mask()
    ctx = torch.randn(1, 77, 768, device=dev)
    td = tempfile.mkdtemp()
    np.save(os.path.join(td, "img_mask.npy"), mask)
    res = {}
    for name, cls in (("hip", CrossAttention), ("torch", TorchCrossAttention)):
        if name == "torch" and a.skip_torch:
            continue
        model = build(cls, dev)
        model.set_content_name(os.path.join(td, "img.jpg"))
        for m in model.attention_modules():
            m.compute = a.compute
        s = DDIMSampler(model)
        s.make_schedule(a.steps, 0.0)

        def one():
            x_T = masked_adain_init(cnt[0]["z_enc"], char[0]["z_enc"], back[0]["z_enc"], torch.from_numpy(mask).to(dev))
            return s.sample(x_T, ctx, injected_features=merged, start_step=a.steps - 1)

        out = one()                      # warm-up (MIOpen picks its convolution kernels here)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.images):
            out = one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.images
        res[name] = {"s_per_image": dt, "imgs_per_s": 1.0 / dt, "latent": out.float().cpu().numpy()}
        del model, s
        torch.cuda.empty_cache()
    line = {k: {kk: vv for kk, vv in v.items() if kk != "latent"} for k, v in res.items()}
    if "torch" in res:
        ref, got = res["torch"]["latent"], res["hip"]["latent"]
        mse = float(((ref - got) ** 2).mean())
        line["psnr_db_hip_vs_torch"] = float(10 * np.log10((ref.max() - ref.min()) ** 2 / mse))
        line["speedup"] = res["torch"]["s_per_image"] / res["hip"]["s_per_image"]
    line["config"] = {"steps": a.steps, "unet_evals": a.steps - 1, "compute": a.compute, "unet": "SD-v1 topology, random weights, fp32"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
