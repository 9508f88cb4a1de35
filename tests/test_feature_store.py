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


# ---- loader: prefetch, LRU, packed side-car (host side; the GPU routes are in test_gpu_loader.py) ----
def _write_files(tmp_path, names, steps=50):
    for i, n in enumerate(names):
        fs.save_feature_file(str(tmp_path / n), make_feats(40 + i, steps=steps))


def _same_lists(a, b):
    for x, y in zip(a, b):
        assert set(x) == set(y)
        for k in x:
            if k != "config":
                assert torch.equal(x[k], y[k]), k
            else:
                assert x[k] == y[k]


def test_loader_prefetch_hands_out_what_the_plain_reader_reads(tmp_path):
    from style_transfer_amd.loader import FeatureLoader
    _write_files(tmp_path, ["c_cnt.pkl", "a_sty.pkl", "b_sty.pkl"])
    ref = fs.FeatureStore.load(str(tmp_path), "c", "a", "b", "cpu")
    for prefetch in (True, False):
        ld = FeatureLoader(str(tmp_path), "cpu", heads=2, packed=False, prefetch=prefetch)
        ld.request(("c", "a", "b"))
        st = ld.get(("c", "a", "b"))
        for got, want in zip(st.lists(), ref.lists()):
            _same_lists(got, want)
        assert ld.stats["files"] == 3 and ld.stats["hits"] == 0 and ld.stats["routes"]["pkl"] == 3
        single = ld.get(("c", "a", None))          # single-style pair: two lists, both already resident
        assert single.back is None and ld.stats["hits"] == 2 and ld.stats["files"] == 3
        same = ld.get(("c", "a", "a"))             # one style file in both roles is read once
        assert same.char is same.back
        ld.close()


def test_loader_keeps_lists_resident_and_evicts_least_recently_used(tmp_path):
    from style_transfer_amd.loader import FeatureLoader
    _write_files(tmp_path, ["c1_cnt.pkl", "c2_cnt.pkl", "a_sty.pkl", "b_sty.pkl", "b2_sty.pkl"], steps=3)
    ld = FeatureLoader(str(tmp_path), "cpu", heads=2, packed=False, prefetch=True)
    sweep = [("c1", "a", "b"), ("c2", "a", "b"), ("c1", "a", "b2"), ("c2", "a", "b2")]
    ld.request(sweep[0])
    for i, key in enumerate(sweep):
        ld.get(key)
        if i + 1 < len(sweep):
            ld.request(sweep[i + 1])
    assert ld.stats["files"] == 5 and ld.stats["hits"] == 7          # 12 list uses, 5 reads
    one = ld.resident_bytes() // 5
    ld.cache_bytes = 3 * one + one // 2                               # room for three lists
    ld.get(("c1", "a", "b"))
    assert ld.stats["evicted"] == 2 and len(ld._cache) == 3
    assert set(ld._cache) == {("cnt", "c1"), ("sty", "a"), ("sty", "b")}     # the triple in use is never evicted
    ld.close()
    with pytest.raises(FileNotFoundError):
        FeatureLoader(str(tmp_path), "cpu", packed=False, prefetch=False).get(("nope", "a", "b"))


def _fake_packed(feats, role):
    """Stand-ins for PackedKV images (no GPU here): what write_sidecar stores is bytes and shape numbers."""
    from style_transfer_amd import ops
    out = []
    for e in feats:
        n = {"config": e["config"], "z_enc": e["z_enc"]}
        for b in range(6, 12):
            pre = "output_block_%d_self_attn_" % b
            n[pre + "q"] = e[pre + "q"]
            k = e[pre + "k"]
            img = torch.cat([k.reshape(-1), e[pre + "v"].reshape(-1)]).view(torch.uint8).clone()
            kv = ops.PackedKV(img, 1, k.shape[0], k.shape[1], k.shape[2], 0)
            n[pre + "k"] = n[pre + "v"] = kv
        out.append(n)
    return out


@pytest.mark.parametrize("role", ["cnt", "sty"])
def test_sidecar_round_trip_is_byte_exact(tmp_path, role):
    from style_transfer_amd import ops
    pkl = tmp_path / ("x_%s.pkl" % role)
    feats = make_feats(77, steps=4)
    fs.save_feature_file(str(pkl), feats)
    packed = _fake_packed(feats, role)
    side = fs.sidecar_path(str(pkl))
    nbytes = fs.write_sidecar(side, packed, role, heads=2, compute="bf16", pkl_path=str(pkl))
    assert side.endswith(".sidpack") and os.path.getsize(side) == nbytes
    rd = fs.SidecarReader(side)
    assert rd.matches(str(pkl), role, 2, "bf16") and not rd.matches(str(pkl), role, 8, "bf16")
    assert not rd.matches(str(pkl), "cnt" if role == "sty" else "sty", 2, "bf16")
    with open(side, "rb") as f:
        for i, want in enumerate(packed):
            fo, n = rd.entry_range(i)
            f.seek(fo)
            flat = torch.frombuffer(bytearray(f.read(n)), dtype=torch.uint8)
            got = rd.build_entry(i, flat)
            assert got["config"] == want["config"] and torch.equal(got["z_enc"], want["z_enc"])
            for b in range(6, 12):
                pre = "output_block_%d_self_attn_" % b
                assert got[pre + "k"] is got[pre + "v"] and isinstance(got[pre + "k"], ops.PackedKV)
                assert torch.equal(got[pre + "k"].image, want[pre + "k"].image)
                assert (got[pre + "k"].H, got[pre + "k"].N, got[pre + "k"].d) == (want[pre + "k"].H, want[pre + "k"].N, want[pre + "k"].d)
                if role == "cnt":
                    assert torch.equal(got[pre + "q"], want[pre + "q"])
                else:
                    assert pre + "q" not in got          # a style list keeps only what the attention reads
    # a side-car made from another pickle (here: the same name rewritten) is stale
    os.utime(str(pkl), ns=(1, 1))
    assert not fs.SidecarReader(side).matches(str(pkl), role, 2, "bf16")
    # a file that is not a side-car, and a truncated one, are refused
    bad = tmp_path / "bad.sidpack"
    bad.write_bytes(b"not a sidecar at all")
    with pytest.raises(ValueError):
        fs.SidecarReader(str(bad))
    with open(side, "rb") as f:
        head = f.read(nbytes - 100)
    bad.write_bytes(head)
    with pytest.raises(ValueError, match="truncated"):
        fs.SidecarReader(str(bad))
