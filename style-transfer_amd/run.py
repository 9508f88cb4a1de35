"""Command line of the stylisation path, flag-compatible with the reference's run_ori.py:96-118:

    python -m style_transfer_amd.run --data_root ./data_vis --precomputed ./precomputed_feats_k \
        --gamma 0.75 --T 1.5 --ratio 0.5 --ckpt models/ldm/stable-diffusion-v1/model.ckpt

For every `cnt char back` line of `<data_root>/dataset.txt` it does what run_ori.py:155-246 does
between loading the features and decoding the image: reads the three feature files of the triple,
builds the masked AdaIN initial latent, merges the features, and runs the 49 injected UNet
evaluations under DDIM.  What it leaves to the host application is everything outside the path:
the VAE (the result is saved as the final latent `<cnt>_<char>_<back>.pt`, not a PNG), the CLIP text
encoder (the unconditional embedding is read from `--uc`, a (1,77,768) tensor file, or is zeros),
and DDIM inversion from images (feature files must exist; `style_transfer_amd.sampler.
extract_features` produces them from latents).

Launched under torch.distributed.run it shards the triples round-robin over the ranks (one process
per GPU) and gathers the final latents on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    # the reference's flags, with its defaults (run_ori.py:97-117)
    p.add_argument('--ddim_inv_steps', type=int, default=50)
    p.add_argument('--save_feat_steps', type=int, default=50)
    p.add_argument('--start_step', type=int, default=49)
    p.add_argument('--ddim_eta', type=float, default=0.0)
    p.add_argument('--H', type=int, default=512)
    p.add_argument('--W', type=int, default=512)
    p.add_argument('--C', type=int, default=4)
    p.add_argument('--f', type=int, default=8)
    p.add_argument('--T', type=float, default=1.5)
    p.add_argument('--gamma', type=float, default=0.75)
    p.add_argument("--attn_layer", type=str, default='6,7,8,9,10,11')
    p.add_argument('--model_config', type=str, default='models/ldm/stable-diffusion-v1/v1-inference.yaml')
    p.add_argument('--precomputed', type=str, default='./precomputed_feats_k')
    p.add_argument('--ckpt', type=str, default='models/ldm/stable-diffusion-v1/model.ckpt')
    p.add_argument('--precision', type=str, default='autocast')
    p.add_argument('--output_path', type=str, default='output_dk')
    p.add_argument("--without_init_adain", action='store_true')
    p.add_argument("--without_attn_injection", action='store_true')
    p.add_argument("--ratio", default=0.5, type=float)
    p.add_argument("--seed", default=22, type=int)
    p.add_argument('--data_root', type=str, default='./data_vis')
    # additions of this implementation
    p.add_argument('--uc', type=str, default='', help="file with the (1,77,768) unconditional text embedding; zeros if empty")
    p.add_argument('--unet_config', type=str, default='', help="JSON with UNetModel keyword arguments (default: SD-v1)")
    p.add_argument('--random_unet', action='store_true', help="seeded random weights instead of --ckpt (measurement)")
    p.add_argument('--compute', default='bf16', choices=['bf16', 'f16', 'fp32'],
                   help="attention kernel arithmetic: bf16 operands; f16 = half-precision query/key operands in the injected "
                        "layers (same speed, finer logits); fp32 = split-bf16, f32-grade")
    p.add_argument('--feature_dtype', default='float32', choices=['float32', 'bfloat16'], help="storage type of the feature store in HBM")
    p.add_argument('--graph_unet', action='store_true', help="replay every UNet evaluation as one HIP graph (not with a residual file)")
    p.add_argument('--block_skip', action='store_true',
                   help="injected attention leaves out key blocks whose softmax mass is provably nil (SID_FLAG_BLOCK_SKIP); "
                        "every layer keeps the flag for an image only if its first call dropped something")
    p.add_argument('--cache_content', action='store_true',
                   help="compute the content block of the injected attention once per content image (when its features are "
                        "loaded) and reuse it, its feature list and its filtered residuals for every style pair with the "
                        "same content image: the dataset.txt lines of one content are processed next to each other, on one rank")
    p.add_argument('--unpacked_store', action='store_true', help="keep key/value features as tensors instead of the kernels' operand layout")
    return p


def list_triples(data_root, dataset_txt="dataset.txt"):
    """(cnt_path, char_path, back_path) per non-empty line; FileNotFoundError for a missing image, with
    the reference's messages (dataset.py:17-37)."""
    triples = []
    with open(os.path.join(data_root, dataset_txt)) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            cnt, char, back = line.split()
            paths = (os.path.join(data_root, "cnt", cnt), os.path.join(data_root, "sty/char", char),
                     os.path.join(data_root, "sty/back", back))
            for what, pth in zip(("Content", "Char", "Back"), paths):
                if not os.path.isfile(pth):
                    raise FileNotFoundError("%s image not found: %s" % (what, pth))
            triples.append(paths)
    return triples


def base(path):
    return os.path.splitext(os.path.basename(path))[0]


def load_unet(opt, device):
    from . import unet as U
    cfg = json.load(open(opt.unet_config)) if opt.unet_config else {}
    model = U.UNetModel(**cfg)
    if opt.random_unet:
        torch.manual_seed(opt.seed)
        for p in model.parameters():
            if p.dim() > 1:
                torch.nn.init.normal_(p, std=p[0].numel() ** -0.5)
    else:
        sd = torch.load(opt.ckpt, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        pre = "model.diffusion_model."
        model.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
    return model.eval().to(device)


def stylise(opt, model, sampler, triple, uc, device):
    """One triple -> final latent (1, C, H/f, W/f)."""
    from . import feature_store as fs
    from .features import masked_adain_init
    from .high_frequency import make_content_injection_schedule, patch_decoder_resblocks_h_and_cnt_hf
    cnt_path, char_path, back_path = triple
    model.set_content_name(cnt_path)                                   # run_ori.py:159-161
    if getattr(opt, "block_skip", False):
        from . import ops
        from .attention import CrossAttention
        for m in model.modules():
            if isinstance(m, CrossAttention):
                m.kernel_flags |= ops.FLAG_BLOCK_SKIP      # kept per layer only where blocks actually die
                m.skip_policy = "auto"
    dtype = torch.bfloat16 if opt.feature_dtype == "bfloat16" else None
    mask_file = os.path.join(opt.data_root, "cnt", base(cnt_path) + "_mask.npy")
    packed = not opt.unpacked_store and not opt.without_attn_injection and os.path.isfile(mask_file)
    cache_content = packed and getattr(opt, "cache_content", False)
    kept = getattr(sampler, "_content_kept", None)          # (content name, its feature list) of the previous triple
    cnt_list = kept[1] if cache_content and kept is not None and kept[0] == base(cnt_path) else None
    store = fs.FeatureStore.load(opt.precomputed, base(cnt_path), base(char_path), base(back_path), device, dtype, cnt=cnt_list)
    if packed:
        store.pack(compute=opt.compute, cache_content=cache_content)   # pre-packed K/V serve the mask-based injected attention
        # (every injected layer of SD-v1 has head_dim 40 or 80, which the f16 mode serves)
    sampler._content_kept = (base(cnt_path), store.cnt) if cache_content else None
    cnt_z, char_z, back_z = store.z_enc()
    if opt.without_init_adain:
        x_T = cnt_z
    else:
        mask = np.load(os.path.join(opt.data_root, "cnt", base(cnt_path) + "_mask.npy")).astype(np.float32)
        x_T = masked_adain_init(cnt_z, char_z, back_z, torch.from_numpy(mask).to(device))
    feats = None if opt.without_attn_injection else store.merged(opt, start_step=opt.start_step)
    res_path = os.path.join(opt.precomputed, base(cnt_path) + "_residuals_all.pkl")
    callback = None
    if os.path.isfile(res_path):                                       # high-frequency injection, --ratio
        from .sampler import last_resblocks
        residuals = fs.load_residuals_file(res_path, device)
        schedule = set(int(t) for t in make_content_injection_schedule(sampler.ddim_timesteps, alpha=0.4))
        hf_cache = None
        if getattr(opt, "cache_content", False):            # HF(h_cnt) too depends on the content image (and the ratio) alone
            kept = getattr(sampler, "_hf_kept", None)
            hf_cache = kept[1] if kept is not None and kept[0] == (base(cnt_path), opt.ratio) else {}
            sampler._hf_kept = ((base(cnt_path), opt.ratio), hf_cache)
        patch_decoder_resblocks_h_and_cnt_hf(model, schedule, residuals, ratio=opt.ratio, hf_cache=hf_cache)
        for _, m in last_resblocks(model):
            m.ri_timestep = None                                       # no stale timestep from the previous image

        def callback(i):                                               # hf_run.py:256-270
            t = int(sampler.ddim_timesteps[i])
            for _, m in last_resblocks(model):
                m.ri_timestep = t
    else:
        from .high_frequency import unpatch_decoder_resblocks
        unpatch_decoder_resblocks(model)                               # a patch left by the previous image
    if opt.graph_unet and callback is None:
        from .graphed import GraphedUNet
        graphed = getattr(sampler, "_graphed", None)
        if graphed is None:
            graphed = sampler._graphed = GraphedUNet(model)
        eager, sampler.unet = sampler.unet, graphed
        try:
            return sampler.sample(x_T, uc, injected_features=feats, start_step=opt.start_step)
        finally:
            sampler.unet = eager
    return sampler.sample(x_T, uc, injected_features=feats, start_step=opt.start_step, callback=callback)


def main(argv=None):
    opt = build_parser().parse_args(argv)
    from . import lib, sharding
    from .attention import set_default_compute
    from .sampler import DDIMSampler
    lib.load()                                                         # fail early without the kernels
    if not torch.cuda.is_available():
        raise SystemExit("style_transfer_amd.run needs a GPU: the path has no CPU fallback")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal on a box with one GPU: SID_DIST_BACKEND=gloo gathers through the host, SID_ONE_DEVICE=1 puts
    # every rank on GPU 0.  Production is one rank per GPU with nccl (= RCCL over xGMI).
    backend = os.environ.get("SID_DIST_BACKEND", "nccl")
    if os.environ.get("SID_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    torch.manual_seed(opt.seed)
    set_default_compute(opt.compute)
    os.makedirs(opt.output_path, exist_ok=True)
    model = load_unet(opt, device)
    sampler = DDIMSampler(model)
    sampler.make_schedule(opt.save_feat_steps, opt.ddim_eta)
    uc = (torch.load(opt.uc, map_location=device, weights_only=True) if opt.uc
          else torch.zeros(1, 77, 768, device=device))
    triples = list_triples(opt.data_root)
    # with --cache_content all triples of a content image go to one rank (what is kept per content image is
    # kept per rank); otherwise round-robin
    lists = (sharding.shard_by_content([base(t[0]) for t in triples], world) if opt.cache_content
             else [sharding.shard_indices(len(triples), r, world) for r in range(world)])
    mine = lists[rank]
    begin = time.time()
    latents = []
    with torch.no_grad():
        for idx in mine:
            z = stylise(opt, model, sampler, triples[idx], uc, device)
            latents.append(z[0].float())
            name = "%s_%s_%s.pt" % tuple(base(p) for p in triples[idx])
            torch.save(z.cpu(), os.path.join(opt.output_path, name))  # run_ori.py:245 names the PNG the same way
    shape = (opt.C, opt.H // opt.f, opt.W // opt.f)
    local_stack = torch.stack(latents) if latents else torch.zeros((0,) + shape, device=device)
    local_stack, _ = sharding.pad_to_common(local_stack, len(triples), world, per=max(len(l) for l in lists))
    if dist is not None and backend != "nccl":
        local_stack = local_stack.cpu()
    gathered = sharding.gather_latents(local_stack, dist, dst=0)
    if rank == 0:
        allz = sharding.reassemble_lists(gathered, lists)
        torch.save(allz.cpu(), os.path.join(opt.output_path, "latents_all.pt"))
        print("Total end: %.3f s for %d triples on %d GPU(s)" % (time.time() - begin, len(triples), world))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
