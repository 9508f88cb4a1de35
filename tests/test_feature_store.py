"""Feature-store files: the `precomputed_feats_k` pickle layout (generate_pkl_only.py:101-142,
run_ori.py:168-222) read without executing anything but tensor reconstruction."""
import os
import pickle
import types

import pytest
import torch

from style_transfer_amd import feature_store as fs


def make_feats(seed, steps=5, heads=2):
    g = torch.Generator().manual_seed(seed)
    feats = []
    for i in range(steps):
        e = {"config": {"T": 1.5}, "z_enc": torch.randn(1, 4, 8, 8, generator=g)}
        for b in range(6, 12):
            n, d = (16, 8) if b < 9 else (64, 4)
            for t in "qkv":
                e["output_block_%d_self_attn_%s" % (b, t)] = torch.randn(heads, n, d, generator=g)
        feats.append(e)
    return feats


def test_reads_what_plain_pickle_dump_wrote(tmp_path):
    feats = make_feats(1)
    p = tmp_path / "content_cnt.pkl"
    with open(p, "wb") as f:
        pickle.dump(feats, f)               # exactly what generate_pkl_only.py:209-210 does
    got = fs.load_feature_file(str(p), device="cpu")
    assert len(got) == len(feats)
    for a, b in zip(got, feats):
        assert a["config"] == b["config"] and set(a) == set(b)
        for k in b:
            if k != "config":
                assert torch.equal(a[k], b[k]) and a[k].is_contiguous()


def test_views_and_non_contiguous_tensors_survive(tmp_path):
    feats = make_feats(2, steps=2)
    base = torch.randn(4, 16, 16)
    feats[0]["output_block_6_self_attn_q"] = base[1:3, :, ::2]      # storage offset + strides
    feats[0]["output_block_6_self_attn_k"] = base[1:3, :, ::2].clone()
    feats[0]["output_block_6_self_attn_v"] = base[1:3, :, ::2].clone()
    p = tmp_path / "x_sty.pkl"
    with open(p, "wb") as f:
        pickle.dump(feats, f)
    got = fs.load_feature_file(str(p), device="cpu")
    assert torch.equal(got[0]["output_block_6_self_attn_q"], base[1:3, :, ::2])


def test_cast_keeps_the_latent_in_full_precision(tmp_path):
    p = tmp_path / "c_cnt.pkl"
    fs.save_feature_file(str(p), make_feats(3, steps=2))
    got = fs.load_feature_file(str(p), device="cpu", dtype=torch.bfloat16)
    assert got[0]["z_enc"].dtype == torch.float32
    assert got[0]["output_block_9_self_attn_k"].dtype == torch.bfloat16


class _Evil:
    def __reduce__(self):
        return (os.system, ("echo pwned > /dev/null",))


def test_refuses_files_that_would_run_code(tmp_path):
    p = tmp_path / "evil_cnt.pkl"
    with open(p, "wb") as f:
        pickle.dump([{"config": {"T": 1.5}, "z_enc": _Evil()}], f)
    with pytest.raises(pickle.UnpicklingError, match="refusing"):
        fs.load_feature_file(str(p), device="cpu")


@pytest.mark.parametrize("bad,msg", [
    ({"not": "a list"}, "non-empty list"),
    ([], "non-empty list"),
    ([["x"]], "not a dict"),
    ([{"config": 3}], "'config' must be a dict"),
    ([{"z_enc": torch.zeros(4, 8, 8)}], "z_enc must be"),
    ([{"output_block_6_self_attn_q": torch.zeros(2, 4)}], "must be \\(heads, N, d\\)"),
    ([{"input_block_1_self_attn_q": torch.zeros(2, 4, 4)}], "unexpected key"),
    ([{"output_block_6_self_attn_q": torch.zeros(2, 4, 4), "output_block_6_self_attn_k": torch.zeros(2, 5, 4)}], "disagree"),
])
def test_layout_errors_name_the_problem(bad, msg):
    with pytest.raises(ValueError, match=msg):
        fs.validate_layout(bad)


def test_store_merges_like_run_ori(tmp_path):
    for name, seed in (("c_cnt.pkl", 10), ("a_sty.pkl", 11), ("b_sty.pkl", 12)):
        fs.save_feature_file(str(tmp_path / name), make_feats(seed, steps=50, heads=1))
    st = fs.FeatureStore.load(str(tmp_path), "c", "a", "b", device="cpu")
    zc, za, zb = st.z_enc()
    assert torch.equal(zc, st.cnt[0]["z_enc"]) and zc.data_ptr() != st.cnt[0]["z_enc"].data_ptr()
    opt = types.SimpleNamespace(gamma=0.75, T=1.5)
    merged = st.merged(opt, start_step=49)
    assert len(merged) == 50
    e = merged[1]
    assert e["output_block_9_self_attn_q"] is st.cnt[1]["output_block_9_self_attn_q"]          # content query
    assert e["output_block_9_self_attn_k"] is st.char[1]["output_block_9_self_attn_k"]         # character style
    assert e["output_block_9_self_attn_k_sty2"] is st.back[1]["output_block_9_self_attn_k"]    # background style
    assert e["output_block_9_self_attn_k_cnt"] is st.cnt[1]["output_block_9_self_attn_k"]
    assert e["config"]["gamma"] == 0.75 and e["config"]["T"] == 1.5
    assert st.nbytes() > 0
