"""Where the bf16-operand error of the injected attention sits: per-row error distribution against the
float64 oracle and against the bf16-operand error model, for unit-scale and feature-like inputs, hard
and real (soft-edged) masks.  GPU box:  python tools/err_dist.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bf16_model  # noqa: E402
import sid_oracle as orc  # noqa: E402
from helpers import load_masks  # noqa: E402
from style_transfer_amd import ops, synth  # noqa: E402

H = 8
dev = torch.device("cuda:0")


def main():
    for (N, d) in ((4096, 40), (1024, 80)):
        side = int(N ** 0.5)
        mask = orc.mask_downsample(load_masks()["content_flower_mask"], side, side, 1.0)
        frac = np.nonzero(np.abs(mask) != 1.0)[0]
        rows = np.unique(np.concatenate([frac, (synth.hash_u64(5, 200) % np.uint64(N)).astype(np.int64)]))
        isfrac = np.isin(rows, frac)
        for sigma in (1.0, 2.0, 3.0):
            t = synth.attention_inputs(int(700 + sigma), H, N, d, sigma, 1.0)
            a = (t["xq"][0], t["q_cnt"], t["k1"], t["v1"], t["k2"], t["v2"], t["k_cnt"], t["v_cnt"], mask, 0.5, 1.5, d ** -0.5)
            ref = orc.injected_attention_rows(rows, *a)
            models = {"bf16": bf16_model.injected_attention_rows(rows, *a), "f16": bf16_model.injected_attention_rows(rows, *a, qk="f16")}
            g = {k: torch.from_numpy(v).to(dev) for k, v in t.items()}
            for compute in ("bf16", "f16", "fp32"):
                model = models.get(compute, models["bf16"])
                out = ops.injected_attention(g["xq"], g["q_cnt"], g["k1"], g["v1"], g["k2"], g["v2"], g["k_cnt"], g["v_cnt"],
                                             torch.from_numpy(mask).to(dev), 0.5, 1.5, d ** -0.5, H, compute=compute)
                got = out[0].cpu().numpy().astype(np.float64)[rows]
                den = np.abs(ref).max()
                e_ref = np.abs(got - ref).max(1) / den                       # per row, max-normalised
                e_row = np.abs(got - ref).max(1) / np.abs(ref).max(1)        # per row, row-normalised
                e_mod = np.abs(got - model).max(1) / den
                m_ref = np.abs(model - ref).max(1) / den
                pc = lambda x: "p50 %.1e p90 %.1e p99 %.1e max %.1e" % tuple(np.percentile(x, [50, 90, 99, 100]))
                print("N=%d d=%d sigma=%.0f %s | vs ref: %s | row-norm: %s" % (N, d, sigma, compute, pc(e_ref), pc(e_row)))
                print("      frac rows max %.1e, hard rows max %.1e | gpu vs bf16 model: %s | model vs ref: %s"
                      % (e_ref[isfrac].max(), e_ref[~isfrac].max(), pc(e_mod), pc(m_ref)), flush=True)


if __name__ == "__main__":
    main()
