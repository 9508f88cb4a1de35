"""The entry point that allocates nothing, sid_injected_attn_fwd_static, against sid_injected_attn_fwd:
same kernels, same decomposition, scratch in the library's static device array instead of a caller
workspace.  Two calls in a row share that array, the second with a smaller input."""
import numpy as np
import pytest
import torch

from style_transfer_amd import lib, ops, synth

pytestmark = pytest.mark.gpu


def _inputs(dev, seed, H, N, d):
    t = synth.attention_inputs(seed, H, N, d, 1.0, 1.0)
    g = {k: torch.from_numpy(v).to(dev) for k, v in t.items()}
    side = int(round(N ** 0.5))
    m = ops.mask_downsample(torch.from_numpy(synth.disc_mask()).to(dev), side, side, 1.0)
    return g, m


def _both(g, m, H, d, gamma, T, compute="f16"):
    args = (g["xq"], g["q_cnt"], g["k1"], g["v1"], g["k2"], g["v2"], g["k_cnt"], g["v_cnt"], m, gamma, T, d ** -0.5, H)
    want = ops.injected_attention(*args, compute=compute)
    out = torch.full_like(want, float("nan"))
    got = ops.injected_attention(*args, compute=compute, out=out, static_scratch=True)
    torch.cuda.synchronize()
    return got, want


def test_static_scratch_two_calls_second_smaller():
    dev = torch.device("cuda:0")
    H = 8
    big, mb = _inputs(dev, 901, H, 4096, 40)
    small, ms = _inputs(dev, 902, H, 1024, 80)
    tiny, mt = _inputs(dev, 903, H, 256, 40)
    got, want = _both(big, mb, H, 40, 0.5, 1.5)
    assert torch.isfinite(got).all()
    assert torch.equal(got, want)                       # same kernels and decomposition: bit-identical
    got, want = _both(small, ms, H, 80, 0.75, 1.0)      # smaller input over the array the first call left
    assert torch.isfinite(got).all()
    assert torch.equal(got, want)
    got, want = _both(tiny, mt, H, 40, 0.5, 1.5, compute="fp32")
    assert torch.equal(got, want)
    got, want = _both(big, mb, H, 40, 0.5, 1.5)         # and the large one again
    assert torch.equal(got, want)
    # the oracle's bar as well (2e-2 relative), on the small call
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import sid_oracle as orc
    t = synth.attention_inputs(902, H, 1024, 80, 1.0, 1.0)
    mask = orc.mask_downsample(synth.disc_mask(), 32, 32, 1.0)
    ref = orc.injected_attention(t["xq"][0], t["q_cnt"], t["k1"], t["v1"], t["k2"], t["v2"], t["k_cnt"], t["v_cnt"],
                                 mask, 0.75, 1.0, 80 ** -0.5)
    got, _ = _both(small, ms, H, 80, 0.75, 1.0)
    err = np.abs(got[0].cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err <= 2e-2, err


def test_static_scratch_rejects_oversize():
    dev = torch.device("cuda:0")
    L = lib.load()
    H, N, d = 8, 16384, 40
    assert L.sid_injected_attn_workspace_bytes(1, H, N, d, 0) > lib.STATIC_SCRATCH_BYTES
    x = torch.zeros(1, N, H * d, device=dev)
    f = torch.zeros(H, N, d, device=dev)
    m = torch.zeros(N, device=dev)
    with pytest.raises(lib.SidError) as e:
        ops.injected_attention(x, f, f, f, f, f, f, f, m, 0.5, 1.5, d ** -0.5, H, out=torch.empty_like(x), static_scratch=True)
    assert e.value.code == -12 and "static" in str(e.value)


def test_static_scratch_under_graph_capture_allocates_nothing():
    """What the entry point is for: after one eager call on the device (it looks the array's address up),
    a call is captured into a HIP graph with a caller-provided output, so nothing may allocate: the
    caching allocator's counters do not move during capture, and the replay equals the eager result."""
    dev = torch.device("cuda:0")
    H, N, d = 8, 1024, 80
    g, m = _inputs(dev, 904, H, N, d)
    args = (g["xq"], g["q_cnt"], g["k1"], g["v1"], g["k2"], g["v2"], g["k_cnt"], g["v_cnt"], m, 0.75, 1.5, d ** -0.5, H)
    out = torch.empty_like(g["xq"])
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eager = ops.injected_attention(*args, out=out, static_scratch=True).clone()      # first call: eager
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    count = lambda: torch.cuda.memory_stats(dev)["allocation.all.allocated"]
    with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
        before = count()             # (entering the capture allocates the graph's own bookkeeping tensors)
        ops.injected_attention(*args, out=out, static_scratch=True)
        after = count()
    assert after == before
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
