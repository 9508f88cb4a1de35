"""Command line: the reference's run_ori.py flags with the reference's defaults (run_ori.py:97-117),
the dataset.txt listing and its errors (dataset.py:17-37)."""
import os

import pytest

from style_transfer_amd import run as cli

REFERENCE_DEFAULTS = {   # run_ori.py:97-117
    "ddim_inv_steps": 50, "save_feat_steps": 50, "start_step": 49, "ddim_eta": 0.0, "H": 512, "W": 512, "C": 4, "f": 8,
    "T": 1.5, "gamma": 0.75, "attn_layer": "6,7,8,9,10,11",
    "model_config": "models/ldm/stable-diffusion-v1/v1-inference.yaml", "precomputed": "./precomputed_feats_k",
    "ckpt": "models/ldm/stable-diffusion-v1/model.ckpt", "precision": "autocast", "output_path": "output_dk",
    "without_init_adain": False, "without_attn_injection": False, "ratio": 0.5, "seed": 22, "data_root": "./data_vis",
}


def test_flags_and_defaults_are_the_reference_ones():
    opt = cli.build_parser().parse_args([])
    for k, v in REFERENCE_DEFAULTS.items():
        assert getattr(opt, k) == v, k
    opt = cli.build_parser().parse_args("--gamma 0.5 --T 1.2 --ratio 0.3 --data_root /d --without_init_adain".split())
    assert (opt.gamma, opt.T, opt.ratio, opt.data_root, opt.without_init_adain) == (0.5, 1.2, 0.3, "/d", True)


def make_root(tmp_path, lines, missing=()):
    for sub in ("cnt", "sty/char", "sty/back"):
        os.makedirs(tmp_path / sub, exist_ok=True)
    for ln in lines:
        c, a, b = ln.split()
        for sub, name in (("cnt", c), ("sty/char", a), ("sty/back", b)):
            if name not in missing:
                (tmp_path / sub / name).write_bytes(b"x")
    (tmp_path / "dataset.txt").write_text("\n".join(lines) + "\n\n")
    return str(tmp_path)


def test_list_triples(tmp_path):
    root = make_root(tmp_path, ["c1.jpg s1.png b1.png", "c2.jpg s1.png b2.png"])
    t = cli.list_triples(root)
    assert len(t) == 2
    assert t[1] == (os.path.join(root, "cnt", "c2.jpg"), os.path.join(root, "sty/char", "s1.png"),
                    os.path.join(root, "sty/back", "b2.png"))
    assert cli.base(t[0][0]) == "c1"


def test_missing_image_raises_like_the_reference(tmp_path):
    root = make_root(tmp_path, ["c1.jpg s1.png b1.png"], missing=("s1.png",))
    with pytest.raises(FileNotFoundError, match="Char image not found"):
        cli.list_triples(root)


def test_two_column_lines_are_single_style_pairs(tmp_path):
    """`cnt sty` lines: the reference's single-style driver (hf_run.py:383-455); the style image is looked
    up under sty/char, then sty."""
    for sub in ("cnt", "sty/char", "sty/back", "sty"):
        os.makedirs(tmp_path / sub, exist_ok=True)
    for sub, name in (("cnt", "c1.jpg"), ("cnt", "c2.jpg"), ("sty/char", "s1.png"), ("sty/back", "b1.png"), ("sty", "s9.png")):
        (tmp_path / sub / name).write_bytes(b"x")
    (tmp_path / "dataset.txt").write_text("c1.jpg s1.png b1.png\nc2.jpg s1.png\nc2.jpg s9.png\n")
    t = cli.list_triples(str(tmp_path))
    assert t[0][2] is not None and t[1][2] is None and t[2][2] is None
    assert t[1][1].endswith(os.path.join("sty/char", "s1.png")) and t[2][1].endswith(os.path.join("sty", "s9.png"))
    (tmp_path / "dataset.txt").write_text("c1.jpg nothere.png\n")
    with pytest.raises(FileNotFoundError, match="Style image not found"):
        cli.list_triples(str(tmp_path))


def test_mask_file_decides_the_path_of_a_line(tmp_path):
    """Three names and a mask file: two styles.  Two names and no mask: single style.  Three names without a
    mask fall back to the character style alone; two names WITH a mask are refused (the reference's mask
    branch needs both styles, ldm/modules/attention.py:449-478)."""
    import numpy as np
    import types
    os.makedirs(tmp_path / "cnt")
    np.save(str(tmp_path / "cnt" / "c1_mask.npy"), np.zeros((4, 4), np.float32))
    opt = types.SimpleNamespace(data_root=str(tmp_path))
    p = lambda *names: tuple(os.path.join(str(tmp_path), n) if n else None for n in names)
    assert cli.bases_of(opt, p("cnt/c1.jpg", "sty/char/a.png", "sty/back/b.png")) == ("c1", "a", "b")
    assert cli.bases_of(opt, p("cnt/c2.jpg", "sty/char/a.png", None)) == ("c2", "a", None)
    assert cli.bases_of(opt, p("cnt/c2.jpg", "sty/char/a.png", "sty/back/b.png")) == ("c2", "a", None)
    with pytest.raises(ValueError, match="has a mask file"):
        cli.bases_of(opt, p("cnt/c1.jpg", "sty/char/a.png", None))


def test_default_arithmetic_and_loader_flags():
    opt = cli.build_parser().parse_args([])
    assert opt.compute == "f16" and opt.sidecar == "use" and not opt.no_prefetch and opt.store_cache_gb > 0
    from style_transfer_amd import attention
    assert attention._default_compute == "f16"


def test_loader_packs_for_the_models_head_count():
    """A UNet configured with other than 8 heads: the loader packs for that count (a store packed for 8 heads
    would be refused by the attention calls of such a model)."""
    import types
    from style_transfer_amd import unet as U
    model = U.UNetModel(model_channels=32, num_heads=4, channel_mult=(1, 2), attention_resolutions=(1, 2), context_dim=16)
    assert cli.model_heads(model) == 4
    opt = types.SimpleNamespace(precomputed="/nowhere", feature_dtype="float32", unpacked_store=False, without_attn_injection=False,
                                compute="f16", T=1.5)
    loader = cli.make_loader(opt, model, "cpu", prefetch=False)
    assert loader.heads == 4 and loader.packed and loader.compute == "f16" and loader.T == 1.5
    loader.close()


def test_producer_flags_are_the_reference_ones():
    """generate_pkl_only.py:58-75."""
    from style_transfer_amd import produce
    opt = produce.build_parser().parse_args([])
    want = {"ddim_inv_steps": 50, "save_feat_steps": 50, "start_step": 49, "ddim_eta": 0.0, "H": 512, "W": 512, "C": 4, "f": 8,
            "attn_layer": "6,7,8,9,10,11", "model_config": "models/ldm/stable-diffusion-v1/v1-inference.yaml",
            "precomputed": "./precomputed_feats_k", "ckpt": "models/ldm/stable-diffusion-v1/model.ckpt",
            "precision": "autocast", "seed": 22, "data_root": "./data_vis", "cnt": None, "sty": None}
    for k, v in want.items():
        assert getattr(opt, k) == v, k
