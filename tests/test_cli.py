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
