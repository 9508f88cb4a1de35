"""The benchmark configurations at full size: batch 16 at the four UNet levels, and the 1024x1024
levels (16384 and 4096 tokens, 128x128 feature maps).  Checked by sampled float64 oracle rows and by
properties that need no full-size oracle."""
import numpy as np
import pytest
import torch

import sid_oracle as orc
from helpers import rel_err
from style_transfer_amd import ops, synth

pytestmark = pytest.mark.gpu
H = 8


def dev_inputs(seed, B, N, d, sigma=1.0):
    g = torch.Generator(device="cuda:0"); g.manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device="cuda:0", generator=g) * sigma
    t = {"xq": rn(B, N, H * d)}
    for k in ("q_cnt", "k1", "v1", "k2", "v2", "k_cnt", "v_cnt"):
        t[k] = rn(B * H, N, d)
    t["mask"] = torch.sign(rn(B, N)).contiguous()
    return t


def run(t, d, compute):
    return ops.injected_attention(t["xq"], t["q_cnt"], t["k1"], t["v1"], t["k2"], t["v2"], t["k_cnt"], t["v_cnt"], t["mask"],
                                  0.5, 1.5, d ** -0.5, H, compute=compute)


def check_rows(t, out, b, rows, d, tol):
    sl = slice(b * H, (b + 1) * H)
    c = lambda x: x.cpu().numpy()
    ref = orc.injected_attention_rows(rows, c(t["xq"][b]), c(t["q_cnt"][sl]), c(t["k1"][sl]), c(t["v1"][sl]), c(t["k2"][sl]),
                                      c(t["v2"][sl]), c(t["k_cnt"][sl]), c(t["v_cnt"][sl]), c(t["mask"][b]), 0.5, 1.5, d ** -0.5)
    got = c(out[b])[rows]
    assert rel_err(got, ref) < tol, b
    return got, ref, (c(t["xq"][b]), c(t["q_cnt"][sl]), c(t["k1"][sl]), c(t["v1"][sl]), c(t["k2"][sl]), c(t["v2"][sl]),
                      c(t["k_cnt"][sl]), c(t["v_cnt"][sl]), c(t["mask"][b]), 0.5, 1.5, d ** -0.5)


@pytest.mark.parametrize("N,d,sigma", [(4096, 40, 1.0), (1024, 80, 1.0), (256, 160, 1.0), (64, 160, 1.0), (4096, 40, 2.0)])
def test_batch16_levels(N, d, sigma):
    """Kernel microbenchmark configuration: 16 independent problems, each with its own mask; sigma 2 gives
    feature-like logits (biases beyond +-100)."""
    B = 16
    t = dev_inputs(6000 + N, B, N, d, sigma)
    out = run(t, d, "bf16")
    assert torch.isfinite(out).all()
    rows = (synth.hash_u64(6001, 12) % np.uint64(N)).astype(np.int64)
    import bf16_model
    for b in (0, 7, 15):
        # 2e-2 holds at both scales; at sigma 2 the error (1.2e-2 at most, against 5e-3 at sigma 1) is the
        # rounding of the bf16 operands, which the operand-rounding model reproduces to 4e-3
        got, ref, args = check_rows(t, out, b, rows, d, 2e-2)
        model = bf16_model.injected_attention_rows(rows, *args)
        rr = np.abs(got - ref).max(1) / np.abs(ref).max(1)
        print("N=%d sigma=%.0f image %d: max-normalised %.2e, per-row max %.2e, vs operand model %.2e"
              % (N, sigma, b, rel_err(got, ref), rr.max(), rel_err(got, model)))
        assert rel_err(got, model) < 4e-3
    ones = dict(t, v1=torch.ones_like(t["v1"]), v2=torch.ones_like(t["v2"]), v_cnt=torch.ones_like(t["v_cnt"]))
    assert (run(ones, d, "bf16") - 1).abs().max().item() < 1e-5


@pytest.mark.parametrize("N,d", [(16384, 40), (4096, 80)])
@pytest.mark.parametrize("compute", ["f16", "bf16", "fp32"])
def test_1024px_levels(N, d, compute):
    """The 1024x1024 configuration: 16384-token top level and 4096-token second level."""
    t = dev_inputs(6100 + N, 1, N, d)
    out = run(t, d, compute)
    assert torch.isfinite(out).all()
    rows = (synth.hash_u64(6101, 16) % np.uint64(N)).astype(np.int64)
    check_rows(t, out, 0, rows, d, 1e-3 if compute == "fp32" else 2e-2)


def test_hf_filter_128px_maps():
    """128x128 maps (the 1024x1024 configuration) at 320 channels: a sample of channels against the
    oracle, idempotence over all of them."""
    g = torch.Generator(device="cuda:0"); g.manual_seed(6200)
    x = torch.randn(1, 320, 128, 128, device="cuda:0", generator=g)
    s, h = torch.randn_like(x), torch.randn_like(x)
    f = ops.hf_filter(x, 0.5)
    ch = [0, 1, 77, 318, 319]
    assert rel_err(f[:, ch].cpu().numpy(), orc.high_freq_filter(x[:, ch].cpu().numpy(), 0.5)) < 1e-5
    assert rel_err(ops.hf_filter(f, 0.5).cpu().numpy(), f.cpu().numpy()) < 1e-5
    fused = ops.hf_filter(x, 0.5, skip=s, h=h)
    assert rel_err(fused.cpu().numpy(), (s + h + f).cpu().numpy()) < 1e-6


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` as the driver starts it (no WORLD_SIZE): two ranks, here both on the one
    GPU of the box over gloo, one gather, one JSON line that says n_gpus 2."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SID_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--in-flight", "1", "--no-cpu-baseline", "--no-e2e"], env=env, capture_output=True,
                       text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["value"] > 0
    # one rate per rank, and every collective of the timed region (the gather, the closing barrier) had run before t0
    pr = line["per_rank"]
    assert len(pr["imgs_per_s"]) == 2 and pr["min"] == min(pr["imgs_per_s"]) and pr["min"] > 0
    assert pr["first_use_collectives_in_timed_region"] == [] and "gather" in pr["collectives_warmed_before_t0"]
    assert line["value"] <= sum(pr["imgs_per_s"]) * 1.001
    assert 0 < line["roofline"]["frac"] < 1 and 0 < line["roofline"]["call_frac"] <= line["roofline"]["frac"]
    assert 0 < line["hf_filter"]["frac"] < 1
