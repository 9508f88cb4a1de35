"""Seeded random shapes through both attention entry points against the float64 oracle: token counts
that are not multiples of the tile or stage sizes, every head_dim class (including the ones served by a
larger kernel instance and the ones whose rows are not 16-byte aligned), several heads and batch sizes,
soft mask values, random gamma / T, all arithmetic modes, packed and unpacked key/value pairs, the content
block computed in the call and taken from a partial."""
import numpy as np
import pytest
import torch

import sid_oracle as orc
from helpers import rel_err
from style_transfer_amd import ops, synth
from style_transfer_amd.attention import resolve_compute

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {"bf16": 2e-2, "f16": 1e-2, "fp32": 1e-3}


def _case(i):
    h = synth.hash_u64(77000 + i, 8).astype(np.int64) & 0x7FFFFFFF
    N = int([33 + h[0] % 200, 257 + h[0] % 1800, 64 * (1 + h[0] % 20), 300 + h[0] % 31][i % 4])
    d = int([1 + h[1] % 160, 40, 80, 8 * (1 + h[1] % 20), 4 * (1 + h[1] % 40)][i % 5])
    H = int([1, 2, 3, 8][h[2] % 4])
    B = int(1 + h[3] % 2)
    gamma = float([0.5, 0.75, 0.3, 1.0][h[4] % 4])
    T = float([1.5, 1.0, 0.7, 2.5][h[5] % 4])
    sigma = float([1.0, 1.0, 2.0][h[6] % 3])
    return dict(i=i, N=N, d=d, H=H, B=B, gamma=gamma, T=T, sigma=sigma)


@pytest.mark.parametrize("c", [_case(i) for i in range(36)], ids=lambda c: "%(i)d-N%(N)d-d%(d)d-H%(H)d-B%(B)d" % c)
def test_random_injected_shapes(c):
    N, d, H, B = c["N"], c["d"], c["H"], c["B"]
    t = synth.attention_inputs(88000 + c["i"], H, N, d, c["sigma"], 1.0, batch=B)
    g = {k: torch.from_numpy(v).to(DEV) for k, v in t.items()}
    mvals = np.array([-1.0, -1.0, -0.5, 0.0, 0.5, 1.0, 1.0], np.float32)
    m = mvals[(synth.hash_u64(99000 + c["i"], B * N) % np.uint64(7)).astype(np.int64)].reshape(B, N)
    mg = torch.from_numpy(m).to(DEV)
    scale = d ** -0.5
    rows = np.unique((synth.hash_u64(66000 + c["i"], 24) % np.uint64(N)).astype(np.int64))
    refs = []
    for b in range(B):
        sl = slice(b * H, (b + 1) * H)
        refs.append(orc.injected_attention_rows(rows, t["xq"][b], t["q_cnt"][sl], t["k1"][sl], t["v1"][sl], t["k2"][sl],
                                                t["v2"][sl], t["k_cnt"][sl], t["v_cnt"][sl], m[b], c["gamma"], c["T"], scale))
    for mode in ("bf16", "f16", "fp32"):
        compute = resolve_compute(mode, N, d, min(c["T"], 1.0) * scale)
        if mode == "f16" and compute != "f16":
            continue
        for flags in (0, ops.FLAG_THROUGHPUT):
            out = ops.injected_attention(g["xq"], g["q_cnt"], g["k1"], g["v1"], g["k2"], g["v2"], g["k_cnt"], g["v_cnt"], mg,
                                         c["gamma"], c["T"], scale, H, compute=compute, flags=flags)
            o = out.cpu().numpy()
            assert np.isfinite(o).all(), (mode, flags)
            for b in range(B):
                assert rel_err(o[b][rows], refs[b]) < TOL[mode], (mode, flags, b)
        kv = [ops.pack_kv(g[k], g[v], H, compute) for k, v in (("k1", "v1"), ("k2", "v2"), ("k_cnt", "v_cnt"))]
        packed = ops.injected_attention_packed(g["xq"], g["q_cnt"], kv[0], kv[1], kv[2], mg, c["gamma"], c["T"], scale, H,
                                               flags=ops.FLAG_THROUGHPUT)
        assert torch.equal(packed, out), mode
        # the content block from a partial made once: the same rows against the oracle, both decompositions
        partial = ops.content_partial(g["q_cnt"], kv[2], scale, H)
        for flags in (0, ops.FLAG_THROUGHPUT):
            cached = ops.injected_attention_cached(g["xq"], g["q_cnt"], kv[0], kv[1], partial, mg, c["gamma"], c["T"], scale, H,
                                                   flags=flags).cpu().numpy()
            assert np.isfinite(cached).all(), (mode, flags)
            for b in range(B):
                assert rel_err(cached[b][rows], refs[b]) < TOL[mode], (mode, flags, b, "cached")


@pytest.mark.parametrize("c", [_case(100 + i) for i in range(24)], ids=lambda c: "%(i)d-N%(N)d-d%(d)d-H%(H)d-B%(B)d" % c)
def test_random_plain_shapes(c):
    Nq, d, H, B = c["N"], c["d"], c["H"], c["B"]
    Nk = int(1 + (synth.hash_u64(55000 + c["i"], 1)[0] % np.uint64(2 * Nq)))
    q = synth.normal(44000 + c["i"], (B * H, Nq, d), c["sigma"])
    k = synth.normal(45000 + c["i"], (B * H, Nk, d), c["sigma"])
    v = synth.normal(46000 + c["i"], (B * H, Nk, d))
    scale = c["T"] * d ** -0.5
    rows = np.unique((synth.hash_u64(47000 + c["i"], 24) % np.uint64(Nq)).astype(np.int64))
    for mode in ("bf16", "f16", "fp32"):
        compute = resolve_compute(mode, Nk, d, scale)
        if mode == "f16" and compute != "f16":
            continue
        out = ops.plain_attention(torch.from_numpy(q).to(DEV), torch.from_numpy(k).to(DEV), torch.from_numpy(v).to(DEV),
                                  scale, H, "bhnd", "bhnd", compute).cpu().numpy()
        assert np.isfinite(out).all()
        for b in range(B):
            sl = slice(b * H, (b + 1) * H)
            ref = orc.plain_attention_rows(rows, q[sl], k[sl], v[sl], scale)
            assert rel_err(out[b][rows], ref) < TOL[mode], (mode, b)
