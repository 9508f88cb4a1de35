"""Pre-packed K/V (sid_pack_kv + sid_injected_attn_fwd_packed) against the per-call packing entry point,
and head dims that are served by a rounded-up kernel instance."""
import ctypes

import numpy as np
import pytest
import torch

import sid_oracle as orc
from helpers import rel_err
from style_transfer_amd import lib, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(seed, B, H, N, d, dtype=torch.float32):
    t = synth.attention_inputs(seed, H, N, d, 1.0, 1.0, batch=B)
    g = {k: torch.from_numpy(v).to(DEV).to(dtype) for k, v in t.items()}
    m = torch.sign(torch.from_numpy(synth.normal(seed + 1, (B, N))).to(DEV)).contiguous()
    return t, g, m


@pytest.mark.parametrize("B,N,d,compute,dtype", [(1, 4096, 40, "bf16", torch.float32), (1, 1024, 80, "bf16", torch.float32),
                                                 (2, 256, 160, "bf16", torch.float32), (1, 1024, 80, "fp32", torch.float32),
                                                 (1, 576, 40, "bf16", torch.bfloat16), (1, 100, 64, "fp32", torch.float32),
                                                 (1, 4096, 40, "f16", torch.float32), (1, 1024, 80, "f16", torch.bfloat16)])
@pytest.mark.parametrize("flags", [0, ops.FLAG_THROUGHPUT])
def test_packed_equals_unpacked_bit_for_bit(B, N, d, compute, dtype, flags):
    H = 8
    _, g, m = _inputs(3100 + N, B, H, N, d, dtype)
    want = ops.injected_attention(g["xq"], g["q_cnt"], g["k1"], g["v1"], g["k2"], g["v2"], g["k_cnt"], g["v_cnt"], m,
                                  0.75, 1.5, d ** -0.5, H, compute=compute, flags=flags)
    kv = [ops.pack_kv(g[k], g[v], H, compute) for k, v in (("k1", "v1"), ("k2", "v2"), ("k_cnt", "v_cnt"))]
    got = ops.injected_attention_packed(g["xq"], g["q_cnt"], kv[0], kv[1], kv[2], m, 0.75, 1.5, d ** -0.5, H, flags=flags)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want)
    # an image is reusable: another query, same images
    g2 = torch.roll(g["xq"], 7, dims=1).contiguous()
    want2 = ops.injected_attention(g2, g["q_cnt"], g["k1"], g["v1"], g["k2"], g["v2"], g["k_cnt"], g["v_cnt"], m,
                                   0.5, 1.0, d ** -0.5, H, compute=compute, flags=flags)
    got2 = ops.injected_attention_packed(g2, g["q_cnt"], kv[0], kv[1], kv[2], m, 0.5, 1.0, d ** -0.5, H, flags=flags)
    assert torch.equal(got2, want2)


def test_packed_rejects_mismatched_image():
    H, N, d = 8, 256, 40
    _, g, m = _inputs(3300, 1, H, N, d)
    kv = ops.pack_kv(g["k1"], g["v1"], H, "bf16")
    other = ops.pack_kv(g["k1"][:, :128].contiguous(), g["v1"][:, :128].contiguous(), H, "bf16")
    with pytest.raises(ValueError):
        ops.injected_attention_packed(g["xq"], g["q_cnt"], kv, other, kv, m, 0.5, 1.5, d ** -0.5, H)
    L = lib.load()
    need = L.sid_kv_image_bytes(1, H, N, d, 0)
    small = torch.empty(need - 256, dtype=torch.uint8, device=DEV)
    rc = L.sid_pack_kv(g["k1"].data_ptr(), g["v1"].data_ptr(), small.data_ptr(), small.numel(), 1, H, N, d, 0, 0, None)
    assert rc == -12 and b"smaller" in L.sid_last_error()


@pytest.mark.parametrize("d", [8, 12, 16, 88, 100, 120, 136, 150])
@pytest.mark.parametrize("compute,tol", [("bf16", 2e-2), ("fp32", 1e-3)])
def test_head_dims_served_by_a_larger_instance(d, compute, tol):
    """Head dims between the instantiated (k-steps, dv blocks) pairs run on the next larger pair."""
    H, N = 2, 320
    t, g, m = _inputs(3400 + d, 1, H, N, d)
    out = ops.injected_attention(g["xq"], g["q_cnt"], g["k1"], g["v1"], g["k2"], g["v2"], g["k_cnt"], g["v_cnt"], m,
                                 0.6, 1.5, d ** -0.5, H, compute=compute)[0].cpu().numpy()
    ref = orc.injected_attention(t["xq"][0], t["q_cnt"], t["k1"], t["v1"], t["k2"], t["v2"], t["k_cnt"], t["v_cnt"],
                                 m[0].cpu().numpy(), 0.6, 1.5, d ** -0.5, dtype=np.float64)
    assert np.isfinite(out).all()
    assert rel_err(out, ref) < tol
    q, k, v = (torch.from_numpy(synth.normal(3500 + d + i, (H, N, d))).to(DEV) for i in range(3))
    pl = ops.plain_attention(q, k, v, d ** -0.5, H, "bhnd", "bhnd", compute)[0].cpu().numpy()
    pref = orc.plain_attention(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), d ** -0.5, dtype=np.float64)
    assert rel_err(pl, pref) < tol


def test_graph_replay_matches_eager_call_sequence():
    """The launch sequence of a call (query pack -> flash -> merge, then a dependent second call that reads
    the first one's output) captured as a HIP graph and replayed on fresh inputs gives the eager results:
    the kernels of a call keep their order inside a graph, which the benchmark's image graphs rely on."""
    H, N, d = 8, 1024, 80
    _, g, m = _inputs(3600, 1, H, N, d)
    kv = [ops.pack_kv(g[k], g[v], H, "bf16") for k, v in (("k1", "v1"), ("k2", "v2"), ("k_cnt", "v_cnt"))]
    xq = g["xq"].clone()
    out1, out2 = torch.empty_like(xq), torch.empty_like(xq)

    def seq():
        ops.injected_attention_packed(xq, g["q_cnt"], kv[0], kv[1], kv[2], m, 0.5, 1.5, d ** -0.5, H, out=out1)
        ops.injected_attention_packed(out1, g["q_cnt"], kv[1], kv[0], kv[2], m, 0.75, 1.0, d ** -0.5, H, out=out2)
        ops.plain_attention(out2, xq, out1, d ** -0.5, H, out=out1)

    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        seq()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        seq()
    for trial in range(3):
        xq.copy_(torch.roll(g["xq"], 11 * (trial + 1), dims=1))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        got1, got2 = out1.clone(), out2.clone()
        with torch.cuda.stream(st):
            seq()
        torch.cuda.synchronize()
        assert torch.equal(got1, out1) and torch.equal(got2, out2), trial
