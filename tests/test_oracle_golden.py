"""The CPU oracle against fixtures produced by the reference's own modules."""
import os

import numpy as np
import pytest

import sid_oracle as orc
from helpers import (GOLDEN, attention_cases, attention_golden, case_inputs, hf_cases, load_masks,
                     mask_for, module_weights, rel_err)
from style_transfer_amd import synth


def test_mask_downsample_bit_exact_all_masks():
    down = np.load(os.path.join(GOLDEN, "mask_downsample.npz"))
    masks = load_masks()
    assert len(masks) == 9
    for key in down.files:
        name, r = key.split("/")
        r = int(r)
        want = down[key]
        got = orc.mask_downsample(masks[name], r, r, 1.0)
        assert got.dtype == np.float32 and np.array_equal(got, want), key
        assert np.array_equal(orc.mask_downsample_pow2(masks[name], r, 1.0), want), key
        assert np.array_equal(orc.mask_downsample(masks[name], r, r, -1.0), -want), key
        assert set(np.unique(want)) <= {-1.0, -0.5, 0.0, 0.5, 1.0}


def test_mask_downsample_general_bilinear():
    """Grey, non-power-of-two masks.  torch's own result here depends on which of its kernels
    runs and on that build's FMA contraction (its CPU kernel switches form at out_h+out_w=128,
    its GPU kernel uses the nested form), so off the reference's real sizes the pin is 2 ulp of
    1.0, and bit-exact where every product is exact."""
    gen = np.load(os.path.join(GOLDEN, "mask_general.npz"))
    for key in gen.files:
        a, b, s, c = key.split("_")
        ih, iw = map(int, a.split("x"))
        oh, ow = map(int, b.split("x"))
        seed = int(s[1:])
        ch = float(c[2:])
        g = synth.uniform(seed, (ih, iw))
        g[::7, ::5] = 0.5
        got = orc.mask_downsample(g, oh, ow, ch)
        assert np.abs(got - gen[key]).max() <= 2.4e-7, key
        if (ih, iw) == (oh, ow):
            assert np.array_equal(got, gen[key]), key


def _oracle_case(case):
    t = case_inputs(case)
    H, N, d = case["heads"], case["N"], case["d"]
    scale = d ** -0.5
    side = int(np.sqrt(N))
    br = case["branch"]
    xq = t["xq"][0]
    w = module_weights(case) if case.get("module_weights") else None
    if w is not None:
        xq = (xq @ w["to_q"].T).astype(np.float32)
    if br == "injected":
        m = orc.mask_downsample(mask_for(case["mask"]), side, side, 1.0)
        out = orc.injected_attention(xq, t["q_cnt"], t["k1"], t["v1"], t["k2"], t["v2"], t["k_cnt"],
                                     t["v_cnt"], m, case["gamma"], case["T"], scale)
    elif br == "styleid_nomask":
        q = orc.blend_query(xq, t["q_cnt"], case["gamma"])
        out = orc.plain_attention(q, t["k1"], t["v1"], scale, case["T"])
    else:
        x = t["xq"][0]
        if w is not None:
            q, k, v = (orc.split_heads((x @ w[n].T).astype(np.float32), H) for n in ("to_q", "to_k", "to_v"))
        else:
            q = k = v = orc.split_heads(x, H)
        T = case["T"] if br == "tplain" else 1.0
        out = orc.plain_attention(q, k, v, scale, T)
    if w is not None:
        out = (out @ w["to_out_w"].T + w["to_out_b"]).astype(np.float32)
    return out


@pytest.mark.parametrize("case", attention_cases(), ids=lambda c: f"{c['id']}-{c['branch']}-N{c['N']}-d{c['d']}")
def test_attention_oracle_matches_reference(case):
    if case["N"] >= 4096:
        pytest.skip("covered by the sampled-row test (full fp32 replay is slow)")
    gold = attention_golden()
    out = _oracle_case(case)
    rows = np.array(case["rows"])
    want = gold[f"out_rows/{case['id']}"]
    assert rel_err(out[rows], want) < 2e-5
    cs = gold[f"colsum/{case['id']}"]
    assert np.abs(out.astype(np.float64).sum(0) - cs).max() < 1e-3 * max(1.0, np.abs(cs).max())


@pytest.mark.parametrize("case", [c for c in attention_cases() if c["branch"] == "injected"],
                         ids=lambda c: f"{c['id']}-N{c['N']}-d{c['d']}")
def test_injected_rows_oracle_f64(case):
    """Row-sampled float64 restatement (what the GPU tests use at large N) vs the reference."""
    if case.get("module_weights"):
        pytest.skip("row form takes to_q(x) directly")
    gold = attention_golden()
    t = case_inputs(case)
    side = int(np.sqrt(case["N"]))
    m = orc.mask_downsample(mask_for(case["mask"]), side, side, 1.0)
    rows = np.array(case["rows"])
    got = orc.injected_attention_rows(rows, t["xq"][0], t["q_cnt"], t["k1"], t["v1"], t["k2"], t["v2"],
                                      t["k_cnt"], t["v_cnt"], m, case["gamma"], case["T"], case["d"] ** -0.5)
    assert rel_err(got, gold[f"out_rows/{case['id']}"]) < 2e-5


def test_hf_filter_matches_reference():
    gold = np.load(os.path.join(GOLDEN, "hf_filter.npz"))
    misc = hf_cases()
    for c in misc["hf_cases"]:
        h = synth.normal(c["seed"], tuple(c["shape"]))
        want = gold[f"hf/{c['id']}"]
        assert rel_err(orc.high_freq_filter(h, c["ratio"]), want) < 2e-6, c
        assert rel_err(orc.high_freq_filter_unshifted(h, c["ratio"]), want) < 2e-6, c
    for r, n in misc["zeroed_bins_ratio_0p5"].items():
        r = int(r)
        assert int(r * r - orc.hf_keep_mask(r, r, 0.5).sum()) == n


def test_schedules_match_reference():
    misc = hf_cases()
    ts = orc.make_ddim_timesteps_uniform(50)
    assert [int(t) for t in ts] == misc["ddim_timesteps_50"]
    assert [int(t) for t in orc.make_ddim_timesteps_uniform(20)] == misc["ddim_timesteps_20"]
    assert [int(t) for t in orc.make_content_injection_schedule(ts)] == misc["content_injection_schedule_50"]
    assert [int(t) for t in orc.make_content_injection_schedule(ts, 0.25)] == misc["content_injection_schedule_50_a25"]


def test_masked_adain_oracle_matches_reference_goldens():
    """run_ori.py:77-82 (adain) and :228-232 (mask blend) against outputs of the reference's own
    `adain` and the same torch calls."""
    import json
    gold = np.load(os.path.join(GOLDEN, "adain.npz"))
    cases = json.load(open(os.path.join(GOLDEN, "adain_cases.json")))
    for c in cases:
        cnt, char, back = (synth.normal(sd, tuple(c["shape"]), sigma=c["sigma"]) + np.float32(0.3 * j)
                           for j, sd in enumerate(c["seeds"]))
        mask = load_masks()[c["mask"]]
        assert rel_err(orc.adain(cnt, char), gold[f"adain/{c['id']}"]) < 2e-6, c
        assert rel_err(orc.masked_adain_init(cnt, char, back, mask), gold[f"blend/{c['id']}"]) < 2e-6, c
