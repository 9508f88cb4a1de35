"""GPU parity: the HIP path (through the C ABI) against the reference's golden outputs and the
CPU oracle.  Tolerances are the bars of the path: mask bit-exact; attention max|err|/max|ref|
<= 2e-2 in the half-precision modes (the default, "f16", is held to 1e-2, at feature-sized logits too)
and <= 1e-3 in the f32-grade mode; FFT filter 1e-5."""
import os

import numpy as np
import pytest
import torch

import sid_oracle as orc
from helpers import (GOLDEN, attention_cases, attention_golden, case_inputs, hf_cases, load_masks, mask_for,
                     rel_err)
from style_transfer_amd import ops, synth

pytestmark = pytest.mark.gpu

TOL = {"bf16": 2e-2, "fp32": 1e-3, "f16": 1e-2}      # "f16": half-precision q, k operands, bf16 P, V
# bf16 operands: the error of a logit grows with |q||k|, and a peaked softmax turns it into an output
# error of the same relative size.  Measured (tools/err_dist.py) and reproduced by the operand-rounding
# model (oracle/bf16_model.py): 5e-3 at unit-scale q, k; 1.2e-2 at twice, 2.5-2.9e-2 at three times that
# scale.  The 2e-2 bar therefore holds up to sigma 2; at sigma 3 the bound below applies and the kernel
# must still sit on the model, i.e. add nothing to what bf16 operands cost any implementation.
BF16_BOUND_SIGMA3 = 3.5e-2
MODEL_TOL = 4e-3


def bar(case, compute):
    if compute == "bf16" and case["sigma_qk"] > 2.0:
        return BF16_BOUND_SIGMA3
    return TOL[compute]


def row_rel_err(a, b):
    """Per-row relative error max_c|a-b| / max_c|b|: stricter than the tensor-wide norm of `rel_err`
    on rows whose entries are small."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-30)


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def g(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def test_mask_downsample_bit_exact():
    down = np.load(os.path.join(GOLDEN, "mask_downsample.npz"))
    masks = load_masks()
    for key in down.files:
        name, r = key.split("/")
        r = int(r)
        m = g(masks[name])
        got = ops.mask_downsample(m, r, r, 1.0).cpu().numpy()
        assert np.array_equal(got, down[key]), key
        neg = ops.mask_downsample(m, r, r, -1.0).cpu().numpy()
        assert np.array_equal(neg, -down[key]), key


def test_mask_downsample_general():
    gen = np.load(os.path.join(GOLDEN, "mask_general.npz"))
    for key in gen.files:
        a, b, s, c = key.split("_")
        ih, iw = map(int, a.split("x"))
        oh, ow = map(int, b.split("x"))
        grey = synth.uniform(int(s[1:]), (ih, iw))
        grey[::7, ::5] = 0.5
        got = ops.mask_downsample(g(grey), oh, ow, float(c[2:])).cpu().numpy()
        # bit-exact against the oracle's evaluation order; 2 ulp of 1.0 against torch's CPU kernel
        assert np.array_equal(got, orc.mask_downsample(grey, oh, ow, float(c[2:]))), key
        assert np.abs(got - gen[key]).max() <= 2.4e-7, key


def check_sums(out, gold, case, tol):
    """Checksums over ALL rows.  With errors of relative size tol on every element the column sums
    move by at most tol * sum|ref| <= tol * sqrt(N * sum ref^2) (Cauchy-Schwarz)."""
    cs = gold[f"colsum/{case['id']}"]
    css = gold[f"colsqsum/{case['id']}"]
    o = out.astype(np.float64)
    assert (np.abs(o.sum(0) - cs) <= tol * np.sqrt(case["N"] * css) + 1e-6).all()
    assert abs(np.sqrt((o ** 2).sum() / css.sum()) - 1) < tol


def run_injected(case, t, compute, dtype=torch.float32):
    H, N, d = case["heads"], case["N"], case["d"]
    side = int(np.sqrt(N))
    m = ops.mask_downsample(g(mask_for(case["mask"])), side, side, 1.0)
    tt = {k: g(v).to(dtype) for k, v in t.items()}
    out = ops.injected_attention(tt["xq"], tt["q_cnt"], tt["k1"], tt["v1"], tt["k2"], tt["v2"], tt["k_cnt"],
                                 tt["v_cnt"], m, case["gamma"], case["T"], d ** -0.5, H, compute=compute)
    torch.cuda.synchronize()
    return out[0].float().cpu().numpy()


INJ = [c for c in attention_cases() if c["branch"] == "injected" and not c.get("module_weights")]


@pytest.mark.parametrize("compute", ["f16", "bf16", "fp32"])
@pytest.mark.parametrize("case", INJ, ids=lambda c: f"{c['id']}-N{c['N']}-d{c['d']}")
def test_injected_attention_vs_reference_golden(case, compute):
    gold = attention_golden()
    out = run_injected(case, case_inputs(case), compute)
    rows = np.array(case["rows"])
    want = gold[f"out_rows/{case['id']}"]
    assert np.isfinite(out).all()
    tol = bar(case, compute)
    assert rel_err(out[rows], want) < tol
    # per-row relative error, the stricter reading of "relative": twice the tensor-wide bar at most
    rr = row_rel_err(out[rows], want)
    print("case %d %s: max-normalised %.2e, per-row p50 %.2e max %.2e" % (case["id"], compute, rel_err(out[rows], want),
                                                                          np.median(rr), rr.max()))
    assert rr.max() < 2 * tol
    # every row enters the column sums: an error outside the sampled rows shows here
    check_sums(out, gold, case, tol)
    # one row of every 32-query tile (all heads) against the float64 oracle: no tile goes unchecked
    N = case["N"]
    t = case_inputs(case)
    side = int(np.sqrt(N))
    m = orc.mask_downsample(mask_for(case["mask"]), side, side, 1.0)
    tiles = np.arange(0, N, 32)
    probe = np.minimum(tiles + (synth.hash_u64(case["seed"] + 7, len(tiles)) % np.uint64(32)).astype(np.int64), N - 1)
    ref = orc.injected_attention_rows(probe, t["xq"][0], t["q_cnt"], t["k1"], t["v1"], t["k2"], t["v2"], t["k_cnt"],
                                      t["v_cnt"], m, case["gamma"], case["T"], case["d"] ** -0.5)
    assert rel_err(out[probe], ref) < tol
    if compute in ("bf16", "f16"):
        # the whole difference from the reference is operand rounding: the kernel sits on the model
        import bf16_model
        model = bf16_model.injected_attention_rows(rows, t["xq"][0], t["q_cnt"], t["k1"], t["v1"], t["k2"], t["v2"],
                                                   t["k_cnt"], t["v_cnt"], m, case["gamma"], case["T"], case["d"] ** -0.5,
                                                   qk=compute)
        assert rel_err(out[rows], model) < MODEL_TOL


PLAIN = [c for c in attention_cases() if c["branch"] in ("tplain", "plain", "styleid_nomask", "genpkl")
         and not c.get("module_weights")]


@pytest.mark.parametrize("compute", ["f16", "bf16", "fp32"])
@pytest.mark.parametrize("case", PLAIN, ids=lambda c: f"{c['id']}-{c['branch']}-N{c['N']}-d{c['d']}")
def test_plain_attention_vs_reference_golden(case, compute):
    gold = attention_golden()
    t = case_inputs(case)
    H, d = case["heads"], case["d"]
    scale = d ** -0.5
    if case["branch"] == "styleid_nomask":
        q = g(orc.blend_query(t["xq"][0], t["q_cnt"], case["gamma"]))
        out = ops.plain_attention(q, g(t["k1"]), g(t["v1"]), case["T"] * scale, H, "bhnd", "bhnd", compute)
    else:
        x = g(t["xq"])
        T = case["T"] if case["branch"] == "tplain" else 1.0
        out = ops.plain_attention(x, x, x, T * scale, H, "bnhd", "bnhd", compute)
    out = out[0].cpu().numpy()
    rows = np.array(case["rows"])
    assert rel_err(out[rows], gold[f"out_rows/{case['id']}"]) < TOL[compute]
    check_sums(out, gold, case, TOL[compute])


BIG_PLAIN = [c for c in PLAIN if c["N"] == 4096]


@pytest.mark.parametrize("flags", [0, ops.FLAG_THROUGHPUT], ids=["default", "throughput"])
@pytest.mark.parametrize("compute", ["f16", "bf16", "fp32"])
@pytest.mark.parametrize("case", BIG_PLAIN, ids=lambda c: f"{c['id']}-{c['branch']}-T{c['T']}")
def test_plain_attention_4096_every_query_tile(case, compute, flags):
    """Single-block attention at Nq = Nk = 4096, head_dim 40 (input blocks 1, 2 of the UNet; 98 calls per
    image): both decompositions (the default one that fills an idle GPU and the least-work one of
    concurrent callers) against the reference's golden rows, the checksums over all rows, and one
    float64 oracle row in every 32-query tile."""
    gold = attention_golden()
    t = case_inputs(case)
    H, d, N = case["heads"], case["d"], case["N"]
    scale = d ** -0.5
    if case["branch"] == "styleid_nomask":
        qh = orc.blend_query(t["xq"][0], t["q_cnt"], case["gamma"])
        kh, vh = t["k1"], t["v1"]
        T = case["T"]
        out = ops.plain_attention(g(qh), g(kh), g(vh), T * scale, H, "bhnd", "bhnd", compute, flags=flags)
    else:
        qh = kh = vh = orc.split_heads(t["xq"][0], H)
        T = case["T"] if case["branch"] == "tplain" else 1.0
        x = g(t["xq"])
        out = ops.plain_attention(x, x, x, T * scale, H, "bnhd", "bnhd", compute, flags=flags)
    out = out[0].cpu().numpy()
    assert np.isfinite(out).all()
    rows = np.array(case["rows"])
    assert rel_err(out[rows], gold[f"out_rows/{case['id']}"]) < TOL[compute]
    check_sums(out, gold, case, TOL[compute])
    tiles = np.arange(0, N, 32)
    probe = np.minimum(tiles + (synth.hash_u64(case["seed"] + 11, len(tiles)) % np.uint64(32)).astype(np.int64), N - 1)
    ref = orc.plain_attention_rows(probe, qh, kh, vh, scale, T)
    assert rel_err(out[probe], ref) < TOL[compute]
    assert row_rel_err(out[probe], ref).max() < 2 * TOL[compute]


def test_hf_filter_vs_reference_golden():
    gold = np.load(os.path.join(GOLDEN, "hf_filter.npz"))
    for c in hf_cases()["hf_cases"]:
        h = synth.normal(c["seed"], tuple(c["shape"]))
        got = ops.hf_filter(g(h), c["ratio"]).cpu().numpy()
        assert rel_err(got, gold[f"hf/{c['id']}"]) < 1e-5, c


def test_hf_filter_fused_residual_and_odd_channels():
    for (shape, ratio) in [((1, 5, 32, 32), 0.5), ((2, 3, 64, 64), 0.3), ((1, 1, 16, 16), 0.5), ((1, 7, 8, 8), 0.5)]:
        h, s, r = (synth.normal(900 + i, shape) for i in range(3))
        got = ops.hf_filter(g(h), ratio, skip=g(s), h=g(r)).cpu().numpy()
        want = orc.resblock_hf_merge(s, r, h, ratio)
        assert rel_err(got, want) < 1e-5, shape


MODULE_CASES = [c for c in attention_cases() if c.get("module_weights")]


@pytest.mark.parametrize("compute", [None, "bf16", "fp32"], ids=["default", "bf16", "fp32"])
@pytest.mark.parametrize("case", MODULE_CASES, ids=lambda c: f"{c['id']}-{c['branch']}")
def test_cross_attention_module_drop_in(case, compute, tmp_path):
    """The drop-in class end to end (to_q -> kernels -> to_out) against the reference module run
    with the same seeded weights, mask file and keywords."""
    from helpers import module_weights
    from style_transfer_amd.attention import CrossAttention
    gold = attention_golden()
    H, d = case["heads"], case["d"]
    w = module_weights(case)
    mod = CrossAttention(query_dim=H * d, heads=H, dim_head=d).to(dev())
    with torch.no_grad():
        mod.to_q.weight.copy_(g(w["to_q"]))
        mod.to_k.weight.copy_(g(w["to_k"]))
        mod.to_v.weight.copy_(g(w["to_v"]))
        mod.to_out[0].weight.copy_(g(w["to_out_w"]))
        mod.to_out[0].bias.copy_(g(w["to_out_b"]))
    mod.compute = compute            # None: the package default (half-precision logits operands)
    compute = compute or "f16"
    np.save(str(tmp_path / "img_mask.npy"), mask_for(case["mask"]))
    mod.cnt_name = str(tmp_path / "img.jpg")
    t = {k: g(v) for k, v in case_inputs(case).items()}
    cfg = {"T": case["T"], "gamma": case["gamma"]}
    with torch.no_grad():
        if case["branch"] == "injected":
            out = mod(t["xq"], q_injected=t["q_cnt"], k_injected=t["k1"], v_injected=t["v1"],
                      cnt_k_injected=t["k_cnt"], sty_q_injected=None, cnt_v_injected=t["v_cnt"],
                      sty2_q_injected=None, sty2_k_injected=t["k2"], sty2_v_injected=t["v2"], injection_config=cfg)
            # attributes the reference leaves on the module (attention.py:404-406)
            assert mod.k is t["k1"] or torch.equal(mod.k, t["k1"])
            q_ref = orc.blend_query((t["xq"][0] @ mod.to_q.weight.T).cpu().numpy(), t["q_cnt"].cpu().numpy(), case["gamma"])
            assert rel_err(mod.q.cpu().numpy(), q_ref) < 1e-5
        else:
            out = mod(t["xq"], injection_config=cfg)
    out = out[0].cpu().numpy()
    rows = np.array(case["rows"])
    assert rel_err(out[rows], gold[f"out_rows/{case['id']}"]) < TOL[compute]
    check_sums(out, gold, case, TOL[compute])


def test_gen_pkl_mode_exposes_features():
    """Feature-extraction mode keeps q/k/v/qk_sim on the module (generate_pkl_only.py:128-130 reads them)."""
    from style_transfer_amd.attention import CrossAttention
    case = [c for c in attention_cases() if c["branch"] == "genpkl"][0]
    gold = attention_golden()
    H, d, N = case["heads"], case["d"], case["N"]
    mod = CrossAttention(query_dim=H * d, heads=H, dim_head=d).to(dev())
    eye = torch.eye(H * d, device=dev())
    with torch.no_grad():
        for lin in (mod.to_q, mod.to_k, mod.to_v, mod.to_out[0]):
            lin.weight.copy_(eye)
        mod.to_out[0].bias.zero_()
    mod.gen_pkl = True
    x = g(case_inputs(case)["xq"])
    with torch.no_grad():
        out = mod(x)[0].cpu().numpy()
    assert rel_err(out[np.array(case["rows"])], gold[f"out_rows/{case['id']}"]) < 1e-4
    assert tuple(mod.q.shape) == (H, N, d) and tuple(mod.qk_sim.shape) == (H, N, N) and mod.attn is not None


def test_masked_adain_init_vs_reference_golden():
    """Initial latent m*adain(cnt,char) + (1-m)*adain(cnt,back) (run_ori.py:77-82, 228-232) against
    the reference's outputs, plus the single-style form and bf16 storage."""
    import json
    from style_transfer_amd.features import adain, masked_adain_init
    gold = np.load(os.path.join(GOLDEN, "adain.npz"))
    cases = json.load(open(os.path.join(GOLDEN, "adain_cases.json")))
    for c in cases:
        cnt, char, back = (synth.normal(sd, tuple(c["shape"]), sigma=c["sigma"]) + np.float32(0.3 * j)
                           for j, sd in enumerate(c["seeds"]))
        mask = load_masks()[c["mask"]]
        got = masked_adain_init(g(cnt), g(char), g(back), g(mask)).cpu().numpy()
        assert rel_err(got, gold[f"blend/{c['id']}"]) < 1e-5, c
        assert rel_err(got, orc.masked_adain_init(cnt, char, back, mask)) < 1e-5, c
        assert rel_err(adain(g(cnt), g(char)).cpu().numpy(), gold[f"adain/{c['id']}"]) < 1e-5, c
        b16 = masked_adain_init(g(cnt).bfloat16(), g(char).bfloat16(), g(back).bfloat16(), g(mask)).float().cpu().numpy()
        r = lambda a: torch.from_numpy(a).bfloat16().float().numpy()
        assert rel_err(b16, orc.masked_adain_init(r(cnt), r(char), r(back), mask)) < 2 ** -7, c


def test_f16_mode_limits():
    """SID_COMPUTE_F16QK serves every shape the bf16 mode serves, and refuses what it cannot do instead of
    degrading silently: logit scales that would push the scaled queries of the long-sequence kernels out
    of half's range."""
    from style_transfer_amd import lib
    H, N = 2, 512
    def qkv(d, n=N):
        return [g(synth.normal(9000 + i, (H, n, d))) for i in range(3)]
    for d, n, scale in ((64, N, 0.125), (40, 128, 0.158), (24, N, 0.2), (160, 64, 0.08)):
        q, k, v = qkv(d, n)
        out = ops.plain_attention(q, k, v, scale, H, "bhnd", "bhnd", "f16")[0].cpu().numpy()
        c = lambda x: x.cpu().numpy()
        ref = orc.plain_attention(c(q), c(k), c(v), scale)
        assert rel_err(out, ref) < TOL["f16"], (d, n)
    for scale in (1e-6, 1e5):
        q, k, v = qkv(40)
        with pytest.raises(lib.SidError) as e:
            ops.plain_attention(q, k, v, scale, H, "bhnd", "bhnd", "f16")
        assert e.value.code == -22, (scale, str(e.value))
        # the short-sequence kernels scale after the product: any scale goes
        q, k, v = qkv(40, 128)
        assert torch.isfinite(ops.plain_attention(q, k, v, scale, H, "bhnd", "bhnd", "f16")).all()
    # large magnitudes saturate instead of producing infinities
    q, k, v = qkv(40)
    out = ops.plain_attention(q * 1e6, k, v, 40 ** -0.5, H, "bhnd", "bhnd", "f16")
    assert torch.isfinite(out).all()
    q, k, v = qkv(40, 128)
    assert torch.isfinite(ops.plain_attention(q * 1e6, k, v, 40 ** -0.5, H, "bhnd", "bhnd", "f16")).all()
    # packed images carry the mode
    kv = ops.pack_kv(k, v, H, "f16")
    assert kv.compute == lib.COMPUTE_F16QK
    assert ops.pack_kv(*qkv(64)[1:], H, "f16").compute == lib.COMPUTE_F16QK
