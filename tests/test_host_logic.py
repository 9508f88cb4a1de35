"""Host-side logic and the C ABI surface (no GPU, no compute calls)."""
import ctypes
import json
import os
import re
import types

import numpy as np
import pytest

from helpers import GOLDEN, hf_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_feat_merge_2sty_matches_reference():
    from style_transfer_amd.features import feat_merge_2sty
    with open(os.path.join(GOLDEN, "feat_merge.json")) as f:
        gold = json.load(f)
    keys = ["z_enc", "config"] + [f"output_block_{b}_self_attn_{t}" for b in range(6, 12) for t in "qkv"]
    mk = lambda tag: [{k: f"{tag}[{i}].{k}" for k in keys} for i in range(50)]
    opt = types.SimpleNamespace(gamma=0.75, T=1.5)
    for start_step, want in gold.items():
        got = feat_merge_2sty(opt, mk("cnt"), mk("char"), mk("back"), start_step=int(start_step))
        assert got == want, start_step
    # step 0 is skipped at the default start_step=49 and the UNet-facing keys exist from step 1
    got = feat_merge_2sty(opt, mk("cnt"), mk("char"), mk("back"), start_step=49)
    assert list(got[0].keys()) == ["config"]
    assert got[1]["output_block_9_self_attn_k_cnt"] == "cnt[1].output_block_9_self_attn_k"
    assert got[1]["output_block_9_self_attn_k"] == "char[1].output_block_9_self_attn_k"
    assert got[1]["output_block_9_self_attn_v_sty2"] == "back[1].output_block_9_self_attn_v"


def test_schedule_matches_reference():
    from style_transfer_amd.high_frequency import make_content_injection_schedule
    misc = hf_cases()
    ts = np.array(misc["ddim_timesteps_50"])
    assert [int(t) for t in make_content_injection_schedule(ts)] == misc["content_injection_schedule_50"]
    assert [int(t) for t in make_content_injection_schedule(ts, 0.25)] == misc["content_injection_schedule_50_a25"]


def test_library_exports_every_declared_symbol():
    from style_transfer_amd import lib
    header = open(os.path.join(ROOT, "include", "sid.h")).read()
    declared = set(re.findall(r"\b(sid_[a-z0-9_]+)\s*\(", header))
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    L = lib.load()
    for name in declared:
        assert hasattr(L, name), name
    assert L.sid_version() == 100


def test_argument_checks_reject_without_a_gpu():
    """Validation happens before any HIP call, so it can be exercised on a CPU-only host."""
    from style_transfer_amd import lib
    L = lib.load()
    assert L.sid_injected_attn_workspace_bytes(1, 8, 4096, 40, 0) > 0
    assert L.sid_injected_attn_workspace_bytes(1, 8, 4096, 40, 1) > L.sid_injected_attn_workspace_bytes(1, 8, 4096, 40, 0)
    assert L.sid_injected_attn_workspace_bytes(1, 8, 4096, 400, 0) == 0
    assert b"head_dim" in L.sid_last_error()
    rc = L.sid_injected_attn_fwd(*([None] * 10), 1, 8, 64, 40, 0.5, 0.5, 1.5, 0.1, 0, 0, 0, None, 0, None)
    assert rc == -22 and b"NULL" in L.sid_last_error()
    rc = L.sid_plain_attn_fwd(None, None, None, None, 1, 8, 0, 64, 40, 0.1, 0, 0, 0, 0, 0, None, 0, None)
    assert rc == -22
    rc = L.sid_masked_adain(None, None, None, None, 512, 512, None, 1, 4, 64, 64, 0, None)
    assert rc == -22 and b"NULL" in L.sid_last_error()
    one = 0x1000                                          # non-NULL placeholders: rejected before any dereference
    rc = L.sid_masked_adain(one, one, one, None, 512, 512, one, 1, 4, 64, 64, 0, None)
    assert rc == -22 and b"mask and sty_back" in L.sid_last_error()
    rc = L.sid_masked_adain(one, one, None, None, 0, 0, one, 1, 4, 1, 1, 0, None)
    assert rc == -22 and b"at least two elements" in L.sid_last_error()
    assert L.sid_plain_attn_workspace_bytes(1, 8, 4096, 77, 40, 0) == 256        # single-launch path: no scratch
    assert L.sid_plain_attn_workspace_bytes(1, 8, 4096, 4096, 40, 0) > 256
    rc = L.sid_hf_filter(None, None, None, None, 1, 4, 64, 64, 0.5, 0, None)
    assert rc == -22
    buf = ctypes.create_string_buffer(16)
    rc = L.sid_hf_filter(buf, None, None, buf, 1, 4, 48, 64, 0.5, 0, None)     # 48 is not a power of two
    assert rc == -38 and b"powers of two" in L.sid_last_error()
    rc = L.sid_mask_downsample(buf, 512, 512, 64, 64, 2.0, buf, None)
    assert rc == -22 and b"ch must be" in L.sid_last_error()


def test_cached_content_block_sizes_and_argument_checks(monkeypatch):
    """The cached content block: a partial is B*H*N rows of d halves (f32 in the f32-grade mode) plus four
    f32 statistics per row; its call packs one query image instead of two and keeps two blocks of partials,
    so its workspace is smaller than the packed call's; bad arguments are refused before any HIP call."""
    from style_transfer_amd import lib
    L = lib.load()
    rows = 8 * 4096
    assert L.sid_content_partial_bytes(1, 8, 4096, 40, 0) == rows * 40 * 2 + rows * 16
    assert L.sid_content_partial_bytes(1, 8, 4096, 40, 1) == rows * 40 * 4 + rows * 16
    assert L.sid_content_partial_bytes(1, 8, 4096, 400, 0) == 0
    assert 0 < L.sid_injected_attn_cached_workspace_bytes(1, 8, 4096, 40, 0) < L.sid_injected_attn_packed_workspace_bytes(1, 8, 4096, 40, 0)
    assert L.sid_content_partial_workspace_bytes(1, 8, 4096, 40, 0) > 0
    one, al = 0x1000, 0x10000
    rc = L.sid_injected_attn_fwd_cached(one, one, al, al, None, one, al, 1, 8, 64, 40, 0.5, 0.5, 1.5, 0.1, 0, 0, 0, al, 1 << 30, None)
    assert rc == -22 and b"NULL" in L.sid_last_error()
    rc = L.sid_injected_attn_fwd_cached(one, one, al, al, al + 16, one, al, 1, 8, 64, 40, 0.5, 0.5, 1.5, 0.1, 0, 0, 0, al, 1 << 30, None)
    assert rc == -22 and b"256-byte aligned" in L.sid_last_error()
    rc = L.sid_injected_attn_fwd_cached(one, one, al, al, al, one, al, 1, 8, 64, 40, 0.5, 0.5, 1.5, 0.1, 0, 0, 0, al, 256, None)
    assert rc == -12 and b"workspace" in L.sid_last_error()
    rc = L.sid_content_partial(one, al, al, 256, 1, 8, 64, 40, 0.1, 0, 0, al, 1 << 30, None)
    assert rc == -12 and b"partial" in L.sid_last_error()
    rc = L.sid_content_partial(one, al, al, 1 << 30, 1, 8, 64, 40, float("nan"), 0, 0, al, 1 << 30, None)
    assert rc == -22 and b"finite" in L.sid_last_error()
    # a forced split of the keys would need more than the one partial slot the buffer holds: refused
    monkeypatch.setenv("SID_TUNE_W", "2")
    assert L.sid_content_partial_workspace_bytes(1, 8, 4096, 40, 0) == 0
    assert b"one run per head" in L.sid_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from style_transfer_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libsid.so")
    with pytest.raises(ImportError, match="no fallback"):
        lib.load()


def test_cross_attention_shim_signature_matches_reference():
    import inspect
    from style_transfer_amd.attention import CrossAttention
    want = ["self", "x", "context", "mask", "q_injected", "k_injected", "v_injected", "cnt_k_injected",
            "sty_q_injected", "cnt_v_injected", "sty2_q_injected", "sty2_k_injected", "sty2_v_injected",
            "injection_config"]
    assert list(inspect.signature(CrossAttention.forward).parameters) == want
    m = CrossAttention(query_dim=320, heads=8, dim_head=40)
    assert m.scale == 40 ** -0.5 and m.heads == 8 and m.gen_pkl is False and m.cnt_name is None
    assert set(dict(m.named_parameters())) == {"to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight",
                                               "to_out.0.bias"}
    import torch
    with pytest.raises(TypeError):                       # cnt_name=None fails like the reference does
        m(torch.zeros(1, 64, 320))


def _plan(blocks, B, H, Nq, Nk, d, compute=0, flags=0):
    import ctypes
    from style_transfer_amd import lib
    out = (ctypes.c_int * 40)()
    lib.check(lib.load().sid_attn_plan(blocks, B, H, Nq, Nk, d, compute, flags, out))
    p = dict(zip(("W", "G", "SPB", "KS", "NW", "QT", "nQG", "stage_keys"), out))
    p["bound"] = list(out[8:9 + p["W"]])
    return p


def test_attention_plan_covers_every_stage_once():
    """The decomposition the flash kernel runs: the blocks' key stages form one sequence cut into W runs
    at plan[8..].  The runs must tile the sequence (every stage in exactly one, none empty), G must be
    the longest, and KS the largest number of runs any block overlaps (the partial slots the workspace
    holds)."""
    shapes = [(3, 1, 8, 4096, 4096, 40), (3, 2, 8, 4096, 4096, 40), (3, 4, 8, 4096, 4096, 40), (3, 16, 8, 4096, 4096, 40),
              (3, 1, 8, 1024, 1024, 80), (3, 1, 8, 1000, 1000, 40), (3, 1, 8, 16384, 16384, 40), (3, 1, 8, 300, 300, 160),
              (1, 1, 8, 4096, 4096, 40), (1, 2, 8, 1024, 1024, 80), (1, 1, 8, 4096, 700, 64), (3, 1, 1, 64, 64, 40),
              (2, 1, 8, 4096, 4096, 40), (2, 16, 8, 4096, 4096, 40), (2, 1, 8, 1024, 1024, 80), (2, 1, 8, 1000, 1000, 40)]
    for blocks, B, H, Nq, Nk, d in shapes:
        for compute in (0, 1):
            for flags in (0, 2):
                p = _plan(blocks, B, H, Nq, Nk, d, compute, flags)
                T = blocks * p["SPB"]
                bd = p["bound"]
                assert p["SPB"] == -(-Nk // p["stage_keys"])
                assert 1 <= p["W"] <= 24 and len(bd) == p["W"] + 1 and bd[0] == 0 and bd[-1] == T, p
                assert all(b1 > b0 for b0, b1 in zip(bd, bd[1:])), p
                assert p["G"] == max(b1 - b0 for b0, b1 in zip(bd, bd[1:]))
                segs = [sum(1 for b0, b1 in zip(bd, bd[1:]) if b0 < (b + 1) * p["SPB"] and b1 > b * p["SPB"])
                        for b in range(blocks)]
                assert p["KS"] == max(segs), (p, segs)
                assert p["nQG"] * p["NW"] * p["QT"] * 32 >= Nq
                if flags & 2:                                        # concurrent callers: one run per block
                    assert p["W"] == blocks and p["KS"] == 1 and bd == [b * p["SPB"] for b in range(blocks + 1)]


def test_attention_plan_of_the_headline_shapes():
    """N = 4096, d = 40 (the 64x64 levels of the UNet): 256 eight-wave workgroups of 48 stages at B = 1 (one per CU),
    one workgroup per (head, query group) walking all three blocks from B = 4 on."""
    p = _plan(3, 1, 8, 4096, 4096, 40)
    assert (p["W"], p["KS"], p["NW"], p["QT"], p["nQG"]) == (4, 2, 8, 2, 8)
    # a run that changes block is shorter by what the change costs: 192 stages + 2 changes in four shares
    assert p["bound"] == [0, 51, 96, 142, 192]
    p = _plan(3, 4, 8, 4096, 4096, 40)
    assert (p["W"], p["G"], p["KS"]) == (1, 192, 1)
    p = _plan(3, 1, 8, 1024, 1024, 80)
    assert (p["W"], p["G"], p["KS"], p["QT"]) == (3, 16, 1, 1)
    assert _plan(1, 1, 8, 256, 77, 160)["W"] == 0                  # the short-sequence kernel has no plan
    from style_transfer_amd import lib
    import ctypes
    out = (ctypes.c_int * 40)()
    # the cached-content call runs the two style blocks: four runs of 32 stages, one workgroup per CU
    p = _plan(2, 1, 8, 4096, 4096, 40)
    assert (p["W"], p["KS"], p["bound"]) == (4, 2, [0, 32, 64, 96, 128])
    assert lib.load().sid_attn_plan(4, 1, 8, 64, 64, 40, 0, 0, out) != 0
    assert lib.load().sid_attn_plan(2, 1, 8, 64, 32, 40, 0, 0, out) != 0
    assert lib.load().sid_attn_plan(3, 1, 8, 64, 32, 40, 0, 0, out) != 0


def test_every_head_dim_up_to_160_has_a_kernel_and_huge_key_counts_are_refused():
    from style_transfer_amd import lib
    L = lib.load()
    plan = (ctypes.c_int * 40)()
    for d in range(1, 161):
        for blocks in (1, 3):
            assert L.sid_attn_plan(blocks, 1, 8, 1024, 1024, d, 0, 0, plan) == 0, (d, blocks)
            assert plan[0] >= 1
    assert L.sid_attn_plan(3, 1, 8, 1024, 1024, 161, 0, 0, plan) == -22
    # 3 blocks x ceil(N / 64) stages must fit the 16-bit run boundaries
    assert L.sid_attn_plan(3, 1, 1, 1500000, 1500000, 40, 0, 0, plan) == -22
    assert b"stages" in L.sid_last_error()
    assert L.sid_kv_image_bytes(1, 8, 4096, 40, 0) == 8 * 4096 * (48 + 64) * 2
    assert L.sid_injected_attn_packed_workspace_bytes(1, 8, 4096, 40, 0) < L.sid_injected_attn_workspace_bytes(1, 8, 4096, 40, 0)


def test_block_skip_planning_queries():
    """SID_FLAG_BLOCK_SKIP: the plan accounts for the launch's padding groups, the counters have a place in
    the workspace exactly where the flag applies, and the workspace covers either decomposition."""
    import ctypes
    from style_transfer_amd import lib
    L = lib.load()
    plan = (ctypes.c_int * 40)()
    assert L.sid_attn_plan(3, 1, 8, 4096, 4096, 40, lib.COMPUTE_BF16, 0, plan) == 0
    w_off = plan[0]
    assert L.sid_attn_plan(3, 1, 8, 4096, 4096, 40, lib.COMPUTE_BF16, 1, plan) == 0
    w_on = plan[0]
    # 8 heads x (8 + 2) query groups x W workgroups of one CU each must fit the 256 CUs in one round
    assert w_off == 4 and w_on == 3 and 8 * 10 * w_on <= 256
    assert list(plan[8:8 + w_on + 1]) == [0, 64, 128, 192]
    ws = L.sid_injected_attn_workspace_bytes(1, 8, 4096, 40, lib.COMPUTE_BF16)
    for packed in (0, 1):
        off = L.sid_injected_attn_skip_counts_offset(1, 8, 4096, 40, lib.COMPUTE_BF16, 1, packed)
        assert 0 < off and off % 256 == 0 and off + 16 <= ws
    assert L.sid_injected_attn_skip_counts_offset(1, 8, 4096, 40, lib.COMPUTE_BF16, 0, 0) == -1        # flag off
    assert L.sid_injected_attn_skip_counts_offset(1, 8, 4096, 40, lib.COMPUTE_F32X3, 1, 0) == -1      # f32-grade mode
    assert L.sid_injected_attn_skip_counts_offset(1, 8, 1024, 80, lib.COMPUTE_BF16, 1, 0) == -1        # short key blocks
    assert ws <= lib.STATIC_SCRATCH_BYTES


def test_single_style_feat_merge_matches_reference():
    """feat_merge of hf_run.py:40-68 (fixture generated by running that function itself)."""
    from style_transfer_amd.features import feat_merge
    with open(os.path.join(GOLDEN, "feat_merge_single.json")) as f:
        gold = json.load(f)
    keys = ["z_enc", "config"] + [f"output_block_{b}_self_attn_{t}" for b in range(6, 12) for t in "qkv"]
    mk = lambda tag: [{k: f"{tag}[{i}].{k}" for k in keys} for i in range(50)]
    opt = types.SimpleNamespace(gamma=0.75, T=1.5)
    for start_step, want in gold.items():
        assert feat_merge(opt, mk("cnt"), mk("sty"), start_step=int(start_step)) == want, start_step


def test_two_style_merge_tolerates_a_style_list_without_queries():
    """A packed side-car of a style file keeps no queries: the merge then has no '_q_sty1' / '_q_sty2' entries
    (nothing reads them) and every other key is as before."""
    from style_transfer_amd.features import feat_merge_2sty
    keys = ["z_enc", "config"] + [f"output_block_{b}_self_attn_{t}" for b in range(6, 12) for t in "qkv"]
    mk = lambda tag, drop=False: [{k: f"{tag}[{i}].{k}" for k in keys if not (drop and k.endswith("_q"))} for i in range(50)]
    opt = types.SimpleNamespace(gamma=0.75, T=1.5)
    full = feat_merge_2sty(opt, mk("cnt"), mk("char"), mk("back"), start_step=49)
    lean = feat_merge_2sty(opt, mk("cnt"), mk("char", True), mk("back", True), start_step=49)
    for a, b in zip(full, lean):
        assert {k: v for k, v in a.items() if not (k.endswith("_q_sty1") or k.endswith("_q_sty2"))} == b
