"""Producer of the feature store, flag-compatible with the reference's generate_pkl_only.py:57-76:

    python -m style_transfer_amd.produce --data_root ./data_vis --precomputed ./precomputed_feats_k \
        --latents ./latents --ckpt models/ldm/stable-diffusion-v1/model.ckpt

For every `cnt char back` (or `cnt sty`) line of `<data_root>/dataset.txt` it does what
generate_pkl_only.py:181-250 does after the VAE: DDIM inversion of the image's latent with every
attention module in feature-extraction mode, and one file per image in the `precomputed_feats_k` layout,
`{cnt}_cnt.pkl`, `{char}_sty.pkl`, `{back}_sty.pkl`, each skipped when it exists (the reference's only
form of resume).  For content images it also writes `{cnt}_residuals_all.pkl`, the per-timestep ResBlock
residuals that the high-frequency injection reads (generate_pkl_only.py:144-178, :310-316; commented out
in the reference's loop, which is why its run_ori.py never filters).  The VAE is outside the path: the
initial latents are read from `<latents>/{base}.pt`, a (1, C, H/f, W/f) tensor per image.
"""
import argparse
import os
import sys

import torch


def build_parser():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    # the reference's flags, with its defaults (generate_pkl_only.py:58-75)
    p.add_argument('--cnt', default=None)
    p.add_argument('--sty', default=None)
    p.add_argument('--ddim_inv_steps', type=int, default=50)
    p.add_argument('--save_feat_steps', type=int, default=50)
    p.add_argument('--start_step', type=int, default=49)
    p.add_argument('--ddim_eta', type=float, default=0.0)
    p.add_argument('--H', type=int, default=512)
    p.add_argument('--W', type=int, default=512)
    p.add_argument('--C', type=int, default=4)
    p.add_argument('--f', type=int, default=8)
    p.add_argument("--attn_layer", type=str, default='6,7,8,9,10,11')
    p.add_argument('--model_config', type=str, default='models/ldm/stable-diffusion-v1/v1-inference.yaml')
    p.add_argument('--precomputed', type=str, default='./precomputed_feats_k')
    p.add_argument('--ckpt', type=str, default='models/ldm/stable-diffusion-v1/model.ckpt')
    p.add_argument('--precision', type=str, default='autocast')
    p.add_argument("--seed", default=22, type=int)
    p.add_argument('--data_root', type=str, default='./data_vis')
    # additions of this implementation
    p.add_argument('--latents', type=str, default='./latents', help="directory of {base}.pt initial latents (the VAE encoder's output)")
    p.add_argument('--uc', type=str, default='', help="file with the (1,77,768) unconditional text embedding; zeros if empty")
    p.add_argument('--unet_config', type=str, default='', help="JSON with UNetModel keyword arguments (default: SD-v1)")
    p.add_argument('--random_unet', action='store_true', help="seeded random weights instead of --ckpt (measurement)")
    p.add_argument('--no_residuals', action='store_true', help="do not write {cnt}_residuals_all.pkl")
    p.add_argument('--T', type=float, default=1.5, help="the temperature recorded in every entry's config (generate_pkl_only.py:101-104)")
    return p


def produce_one(opt, model, uc, device, image_path, role):
    """One image -> its feature file (and residual file for a content image).  Returns the paths written."""
    from . import feature_store as fs
    from .run import base
    from .sampler import extract_features
    name = base(image_path)
    out = os.path.join(opt.precomputed, name + ("_cnt.pkl" if role == "cnt" else "_sty.pkl"))
    if os.path.isfile(out):                                    # generate_pkl_only.py:196, :216, :236
        print("Precomputed %s feature exists: %s" % ("content" if role == "cnt" else "style", out))
        return []
    lat = os.path.join(opt.latents, name + ".pt")
    if not os.path.isfile(lat):
        raise FileNotFoundError("initial latent not found: %s" % lat)
    x0 = torch.load(lat, map_location=device, weights_only=True).float()
    want = (1, opt.C, opt.H // opt.f, opt.W // opt.f)
    if tuple(x0.shape) != want:
        raise ValueError("%s holds a %s tensor, expected %s" % (lat, tuple(x0.shape), want))
    blocks = [int(b) for b in opt.attn_layer.split(",")]
    residuals = {} if role == "cnt" and not opt.no_residuals else None
    model.set_content_name(image_path)
    with torch.no_grad():
        feats = extract_features(model, x0, uc, num_steps=opt.ddim_inv_steps, blocks=blocks, T=opt.T, residuals=residuals)
    fs.save_feature_file(out, feats)
    written = [out]
    if residuals is not None:
        res = os.path.join(opt.precomputed, name + "_residuals_all.pkl")
        fs.save_residuals_file(res, residuals)
        written.append(res)
    print("Saved %s" % ", ".join(written))
    return written


def main(argv=None):
    opt = build_parser().parse_args(argv)
    from . import lib
    from .run import list_triples, load_unet
    lib.load()
    if not torch.cuda.is_available():
        raise SystemExit("style_transfer_amd.produce needs a GPU: the path has no CPU fallback")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    torch.manual_seed(opt.seed)
    os.makedirs(opt.precomputed, exist_ok=True)
    model = load_unet(opt, device)
    uc = (torch.load(opt.uc, map_location=device, weights_only=True) if opt.uc
          else torch.zeros(1, 77, 768, device=device))
    written = []
    for cnt_path, char_path, back_path in list_triples(opt.data_root):
        written += produce_one(opt, model, uc, device, cnt_path, "cnt")
        written += produce_one(opt, model, uc, device, char_path, "sty")
        if back_path is not None:
            written += produce_one(opt, model, uc, device, back_path, "sty")
    return written


if __name__ == "__main__":
    main(sys.argv[1:])
