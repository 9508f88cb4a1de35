"""PyTorch-ROCm eager restatement of the attention arithmetic, for measurement only: the same module
as style_transfer_amd.attention.CrossAttention with its two kernel calls replaced by stock torch ops
in the reference's formulation (materialised N x 3N logits, duplicate raw-similarity product,
softmax, value product: ldm/modules/attention.py:201-329, 538-546).  Used by tools/e2e.py to time
"the rest of the network + torch attention" against "the rest of the network + HIP kernels"."""
import torch

from style_transfer_amd.attention import CrossAttention


class TorchCrossAttention(CrossAttention):
    def _heads(self, t, layout):
        if layout == "bhnd":
            return t
        return self._split(t)

    def _merge(self, o):
        bh, n, d = o.shape
        h = self.heads
        return o.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, h * d)

    def _injected(self, xq, feats, m, gamma, T, compute):
        q_cnt, k1, v1, k2, v2, k_cnt, v_cnt = feats
        q = q_cnt * gamma + self._split(xq) * (1. - 0.5)
        cc = torch.einsum("hid,hjd->hij", q_cnt, k_cnt) * self.scale
        min_cc = cc.min(dim=-1, keepdim=True)[0]
        mm = m.reshape(1, -1, 1)
        sims = []
        for k, ch in ((k1, -1.0), (k2, 1.0)):
            sim = torch.einsum("hid,hjd->hij", q, k)
            max_sim = torch.einsum("hid,hjd->hij", q, k).max(dim=-1, keepdim=True)[0]     # the reference's second product
            sim = sim * T * self.scale
            sims.append(sim + (min_cc - max_sim) * (mm * ch))
        attn = torch.cat(sims + [cc], dim=-1).softmax(dim=-1)
        out = torch.einsum("hij,hjd->hid", attn, torch.cat([v1, v2, v_cnt], dim=1))
        return self._merge(out)

    def _plain(self, q, k, v, logit_scale, q_layout, kv_layout, compute):
        q, k, v = self._heads(q, q_layout), self._heads(k, kv_layout), self._heads(v, kv_layout)
        attn = (torch.einsum("hid,hjd->hij", q, k) * logit_scale).softmax(dim=-1)
        return self._merge(torch.einsum("hij,hjd->hid", attn, v))
