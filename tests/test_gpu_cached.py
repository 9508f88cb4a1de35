"""Cached content block (sid_content_partial + sid_injected_attn_fwd_cached): the injected attention with
its third key block taken from a partial computed once per content feature entry, against the reference's
golden outputs, the float64 oracle and the entry points that compute all three blocks."""
import numpy as np
import pytest
import torch

import sid_oracle as orc
from helpers import attention_cases, attention_golden, case_inputs, mask_for, rel_err
from style_transfer_amd import lib, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {"bf16": 2e-2, "fp32": 1e-3, "f16": 1e-2}
# two decompositions of one call round their probabilities against different references: the cached call
# differs from the three-block call by that much and no more
DECOMP_TOL = {"bf16": 4e-3, "f16": 4e-3, "fp32": 2e-5}


def g(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def cached_call(tt, m, gamma, T, scale, H, compute, flags=0, partial=None):
    kv = [ops.pack_kv(tt[k], tt[v], H, compute) for k, v in (("k1", "v1"), ("k2", "v2"), ("k_cnt", "v_cnt"))]
    if partial is None:
        partial = ops.content_partial(tt["q_cnt"], kv[2], scale, H)
    out = ops.injected_attention_cached(tt["xq"], tt["q_cnt"], kv[0], kv[1], partial, m, gamma, T, scale, H, flags=flags)
    return out, kv, partial


INJ = [c for c in attention_cases() if c["branch"] == "injected" and not c.get("module_weights") and c["sigma_qk"] <= 2.0]


@pytest.mark.parametrize("compute", ["bf16", "fp32", "f16"])
@pytest.mark.parametrize("case", INJ, ids=lambda c: f"{c['id']}-N{c['N']}-d{c['d']}")
def test_cached_call_vs_reference_golden(case, compute):
    H, N, d = case["heads"], case["N"], case["d"]
    side = int(np.sqrt(N))
    t = case_inputs(case)
    tt = {k: g(v) for k, v in t.items()}
    m = ops.mask_downsample(g(mask_for(case["mask"])), side, side, 1.0)
    out, _, _ = cached_call(tt, m, case["gamma"], case["T"], d ** -0.5, H, compute)
    torch.cuda.synchronize()
    out = out[0].cpu().numpy()
    gold = attention_golden()
    rows = np.array(case["rows"])
    assert np.isfinite(out).all()
    assert rel_err(out[rows], gold[f"out_rows/{case['id']}"]) < TOL[compute]
    # every row enters the column sums
    cs, css = gold[f"colsum/{case['id']}"], gold[f"colsqsum/{case['id']}"]
    o = out.astype(np.float64)
    assert (np.abs(o.sum(0) - cs) <= TOL[compute] * np.sqrt(N * css) + 1e-6).all()
    # one row of every 32-query tile against the float64 oracle
    mo = orc.mask_downsample(mask_for(case["mask"]), side, side, 1.0)
    tiles = np.arange(0, N, 32)
    probe = np.minimum(tiles + (synth.hash_u64(case["seed"] + 11, len(tiles)) % np.uint64(32)).astype(np.int64), N - 1)
    ref = orc.injected_attention_rows(probe, t["xq"][0], t["q_cnt"], t["k1"], t["v1"], t["k2"], t["v2"], t["k_cnt"],
                                      t["v_cnt"], mo, case["gamma"], case["T"], d ** -0.5)
    assert rel_err(out[probe], ref) < TOL[compute]


@pytest.mark.parametrize("B,N,d,compute,dtype", [(1, 4096, 40, "bf16", torch.float32), (1, 1024, 80, "bf16", torch.float32),
                                                 (2, 256, 160, "bf16", torch.float32), (1, 1024, 80, "fp32", torch.float32),
                                                 (1, 576, 40, "bf16", torch.bfloat16), (1, 100, 64, "fp32", torch.float32),
                                                 (3, 333, 24, "bf16", torch.float32), (1, 4096, 40, "f16", torch.float32),
                                                 (4, 4096, 40, "bf16", torch.float32), (1, 16384, 40, "bf16", torch.float32),
                                                 (1, 4096, 80, "f16", torch.float32)])
@pytest.mark.parametrize("flags", [0, ops.FLAG_THROUGHPUT])
def test_cached_equals_three_block_call(B, N, d, compute, dtype, flags):
    """Same inputs through the packed three-block entry point and through the cached one; then the partial is
    reused for another latent, style order, gamma and T (it depends on none of them)."""
    H = 8
    t = synth.attention_inputs(4100 + N + d, H, N, d, 1.0, 1.0, batch=B)
    tt = {k: g(v, dtype) for k, v in t.items()}
    m = torch.from_numpy(synth.normal(4200 + N, (B, N))).to(DEV)
    m = torch.where(m.abs() < 0.15, torch.zeros_like(m), torch.sign(m)).contiguous()     # some boundary rows
    scale = d ** -0.5
    got, kv, partial = cached_call(tt, m, 0.75, 1.5, scale, H, compute, flags)
    want = ops.injected_attention_packed(tt["xq"], tt["q_cnt"], kv[0], kv[1], kv[2], m, 0.75, 1.5, scale, H, flags=flags)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    tol = DECOMP_TOL[compute] + (8e-3 if dtype == torch.bfloat16 else 0.0)      # (bf16 storage rounds the output)
    assert rel_err(got.float().cpu().numpy(), want.float().cpu().numpy()) < tol
    xq2 = torch.roll(tt["xq"], 5, dims=1).contiguous()
    got2 = ops.injected_attention_cached(xq2, tt["q_cnt"], kv[1], kv[0], partial, m, 0.5, 1.0, scale, H, flags=flags)
    want2 = ops.injected_attention_packed(xq2, tt["q_cnt"], kv[1], kv[0], kv[2], m, 0.5, 1.0, scale, H, flags=flags)
    assert rel_err(got2.float().cpu().numpy(), want2.float().cpu().numpy()) < tol


@pytest.mark.parametrize("gamma,T,scale", [(0.0, 1.5, None), (0.5, 0.0, None), (0.5, -1.5, None), (0.75, 1.5, -0.2),
                                           (0.5, 1e-6, None)])
@pytest.mark.parametrize("compute", ["bf16", "fp32"])
def test_cached_corner_parameters(gamma, T, scale, compute):
    """gamma = 0, T = 0, negative T and a negative scale (the content block's LARGEST raw product then feeds
    the bias) against the float64 oracle."""
    H, N, d = 4, 320, 40
    scale = d ** -0.5 if scale is None else scale
    t = synth.attention_inputs(4300, H, N, d, 1.0, 1.0)
    tt = {k: g(v) for k, v in t.items()}
    mo = np.sign(synth.normal(4301, (N,))).astype(np.float32)
    mo[::17] = 0.5
    out, _, _ = cached_call(tt, g(mo), gamma, T, scale, H, compute)
    ref = orc.injected_attention(t["xq"][0], t["q_cnt"], t["k1"], t["v1"], t["k2"], t["v2"], t["k_cnt"], t["v_cnt"],
                                 mo, gamma, T, scale, dtype=np.float64)
    out = out[0].cpu().numpy()
    assert np.isfinite(out).all()
    assert rel_err(out, ref) < TOL[compute]


def test_cached_full_size_properties():
    """(B = 2, N = 4096): V = 1 -> every output is 1 (the cached part's weight enters the row sum like a
    computed one); swapping the styles while negating the mask changes nothing."""
    B, H, N, d = 2, 8, 4096, 40
    t = synth.attention_inputs(4400, H, N, d, 1.0, 1.0, batch=B)
    tt = {k: g(v) for k, v in t.items()}
    for k in ("v1", "v2", "v_cnt"):
        tt[k] = torch.ones_like(tt[k])
    m = torch.sign(torch.from_numpy(synth.normal(4401, (B, N))).to(DEV)).contiguous()
    out, kv, partial = cached_call(tt, m, 0.5, 1.5, d ** -0.5, H, "bf16")
    assert (out - 1).abs().max().item() < 1e-2
    tt2 = {k: g(v) for k, v in t.items()}
    a, kv2, p2 = cached_call(tt2, m, 0.5, 1.5, d ** -0.5, H, "bf16")
    b = ops.injected_attention_cached(tt2["xq"], tt2["q_cnt"], kv2[1], kv2[0], p2, (-m).contiguous(), 0.5, 1.5, d ** -0.5, H)
    assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < DECOMP_TOL["bf16"]


def test_cached_rejects_what_does_not_match():
    H, N, d = 8, 256, 40
    t = synth.attention_inputs(4500, H, N, d, 1.0, 1.0)
    tt = {k: g(v) for k, v in t.items()}
    m = torch.ones(1, N, device=DEV)
    kv = [ops.pack_kv(tt[k], tt[v], H, "bf16") for k, v in (("k1", "v1"), ("k2", "v2"), ("k_cnt", "v_cnt"))]
    partial = ops.content_partial(tt["q_cnt"], kv[2], d ** -0.5, H)
    with pytest.raises(ValueError):          # another scale
        ops.injected_attention_cached(tt["xq"], tt["q_cnt"], kv[0], kv[1], partial, m, 0.5, 1.5, 0.3, H)
    with pytest.raises(ValueError):          # another q_cnt tensor
        ops.injected_attention_cached(tt["xq"], tt["q_cnt"].clone(), kv[0], kv[1], partial, m, 0.5, 1.5, d ** -0.5, H)
    L = lib.load()
    need = L.sid_content_partial_bytes(1, H, N, d, 0)
    small = torch.empty(need - 256, dtype=torch.uint8, device=DEV)
    ws = torch.empty(L.sid_content_partial_workspace_bytes(1, H, N, d, 0), dtype=torch.uint8, device=DEV)
    rc = L.sid_content_partial(tt["q_cnt"].data_ptr(), kv[2].image.data_ptr(), small.data_ptr(), small.numel(), 1, H, N, d,
                               d ** -0.5, 0, 0, ws.data_ptr(), ws.numel(), None)
    assert rc == -12 and b"smaller" in L.sid_last_error()


def test_cached_graph_replay_matches_eager():
    H, N, d = 8, 1024, 80
    t = synth.attention_inputs(4600, H, N, d, 1.0, 1.0)
    tt = {k: g(v) for k, v in t.items()}
    m = torch.sign(torch.from_numpy(synth.normal(4601, (1, N))).to(DEV)).contiguous()
    xq = tt["xq"].clone()
    tt["xq"] = xq
    out = torch.empty_like(xq)
    _, kv, partial = cached_call(tt, m, 0.5, 1.5, d ** -0.5, H, "bf16")

    def call():
        ops.injected_attention_cached(xq, tt["q_cnt"], kv[0], kv[1], partial, m, 0.5, 1.5, d ** -0.5, H, out=out)

    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        call()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        call()
    xq.copy_(torch.roll(xq, 9, dims=1))
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    got = out.clone()
    with torch.cuda.stream(st):
        call()
    torch.cuda.synchronize()
    assert torch.equal(got, out)


def test_partials_made_on_side_streams_equal_those_made_in_line():
    H, N, d = 8, 1024, 80
    jobs = []
    for i in range(7):
        t = synth.attention_inputs(4700 + i, H, N, d, 1.0, 1.0)
        q, k, v = g(t["q_cnt"]), g(t["k_cnt"]), g(t["v_cnt"])
        jobs.append((q, ops.pack_kv(k, v, H, "bf16"), d ** -0.5))
    many = ops.content_partials(jobs, H)
    torch.cuda.synchronize()
    for (q, kv, scale), p in zip(jobs, many):
        one = ops.content_partial(q, kv, scale, H)
        assert torch.equal(one.data, p.data) and p.q_ptr == q.data_ptr()


@pytest.mark.parametrize("shape,dtype", [((1, 320, 64, 64), torch.float32), ((2, 8, 32, 32), torch.float32),
                                         ((1, 5, 8, 16), torch.float32), ((1, 640, 32, 32), torch.bfloat16),
                                         ((1, 4, 128, 128), torch.bfloat16)])
@pytest.mark.parametrize("ratio", [0.5, 0.25])
def test_cached_hf_sum_equals_the_fused_filter_bit_for_bit(shape, dtype, ratio):
    """HF(h_cnt) filtered once (bare sid_hf_filter) and added by sid_residual_sum against the fused
    skip + h + HF(h_cnt): the same additions in the same order, so the same bits; and against the oracle."""
    h_cnt, skip, h = (torch.from_numpy(synth.normal(4800 + i, shape)).to(DEV).to(dtype) for i in range(3))
    fused = ops.hf_filter(h_cnt, ratio, skip=skip, h=h)
    hf = ops.hf_filter(h_cnt, ratio)
    got = ops.residual_sum(skip, h, hf)
    torch.cuda.synchronize()
    assert torch.equal(got, fused)
    if dtype == torch.float32:
        want = orc.resblock_hf_merge(skip.cpu().numpy(), h.cpu().numpy(), h_cnt.cpu().numpy(), ratio)
        assert np.abs(got.cpu().numpy() - want).max() < 1e-4 * np.abs(want).max()
    from style_transfer_amd.high_frequency import resblock_cached_hf_sum
    cache = {}
    a = resblock_cached_hf_sum(skip, h, h_cnt, ratio, cache, (1, 6))
    b = resblock_cached_hf_sum(skip, h, torch.zeros_like(h_cnt), ratio, cache, (1, 6))     # the second use reads the cache
    assert torch.equal(a, fused) and torch.equal(b, fused) and len(cache) == 1


def test_residual_sum_tail_and_argument_checks():
    L = lib.load()
    a, b, c = (torch.from_numpy(synth.normal(4900 + i, (1003,))).to(DEV) for i in range(3))      # 1003 = 4 * 250 + 3
    got = ops.residual_sum(a, b, c)
    assert torch.equal(got, (a + b) + c)
    rc = L.sid_residual_sum(a.data_ptr(), b.data_ptr(), None, got.data_ptr(), 1003, 0, None)
    assert rc == -22 and b"NULL" in L.sid_last_error()
    rc = L.sid_residual_sum(a.data_ptr() + 4, b.data_ptr(), c.data_ptr(), got.data_ptr(), 8, 0, None)
    assert rc == -22 and b"aligned" in L.sid_last_error()
