"""Command line of the stylisation path, flag-compatible with the reference's run_ori.py:96-118:

    python -m style_transfer_amd.run --data_root ./data_vis --precomputed ./precomputed_feats_k \
        --gamma 0.75 --T 1.5 --ratio 0.5 --ckpt models/ldm/stable-diffusion-v1/model.ckpt

For every `cnt char back` line of `<data_root>/dataset.txt` it does what run_ori.py:155-246 does
between loading the features and decoding the image: reads the three feature files of the triple,
builds the masked AdaIN initial latent, merges the features, and runs the 49 injected UNet
evaluations under DDIM.  A two-column line `cnt sty` is the single-style path of the reference's
hf_run.py:383-455 (initial latent adain(cnt, sty), `feat_merge`, the no-mask attention branch; the
content image must have no mask file, as there).  The feature files of the next line are read, moved
to HBM and packed while the current one samples, and every list stays resident under an LRU bound, so
a sweep reads each file once (style_transfer_amd.loader).  What it leaves to the host application is everything outside the path:
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
    p.add_argument('--precision', type=str, default='autocast',
                   help="accepted as in the reference, whose two-style loop parses it and never reads it (run_ori.py:111): the "
                        "network runs in the precision of its weights; see --network_autocast")
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
    p.add_argument('--compute', default='f16', choices=['bf16', 'f16', 'fp32'],
                   help="attention kernel arithmetic: f16 (default) = half-precision query/key operands, bf16 probabilities and "
                        "values (2e-2 of the fp32 reference also at feature-sized logits); bf16 = bf16 operands throughout "
                        "(3 %% faster on the longest layers, up to 3.2e-2 at feature-sized logits); fp32 = split-bf16, f32-grade")
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
    p.add_argument('--lanes', type=int, default=1,
                   help="images in flight on one GPU: each lane replays its own HIP graph of the UNet evaluation on its own stream "
                        "(implies --graph_unet; lines whose content has a residual file are stylised one at a time)")
    p.add_argument('--network_autocast', action='store_true',
                   help="run the host network under bf16 autocast (convolutions and linears in bf16, as hf_run.py:285,418 does "
                        "with fp16 on its hardware); the attention kernels then read bf16 activations: 1.6x end to end")
    p.add_argument('--progress', action='store_true', help="one line on stderr per finished batch of lines")
    p.add_argument('--no_prefetch', action='store_true', help="read the feature files of a line only when its turn comes (the reference's order)")
    p.add_argument('--sidecar', default='use', choices=['off', 'use', 'write'],
                   help="packed side-car files beside the pickles ({base}_cnt.sidpack / {base}_sty.sidpack): 'use' reads one when it "
                        "is present and was made from the pickle beside it, 'write' also writes one after packing a pickle")
    p.add_argument('--store_cache_gb', type=float, default=96.0, help="HBM the loaded feature lists may occupy before the least "
                                                                      "recently used one is dropped")
    return p


def list_triples(data_root, dataset_txt="dataset.txt"):
    """(cnt_path, char_path, back_path) per non-empty line; FileNotFoundError for a missing image, with
    the reference's messages (dataset.py:17-37).  A line of two names, `cnt sty`, is a single-style pair:
    (cnt_path, sty_path, None), the style image looked up under sty/char, then sty."""
    triples = []
    with open(os.path.join(data_root, dataset_txt)) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            names = line.split()
            if len(names) == 2:
                cnt, sty = names
                sty_path = os.path.join(data_root, "sty/char", sty)
                if not os.path.isfile(sty_path):
                    sty_path = os.path.join(data_root, "sty", sty)
                found = ((("Content", os.path.join(data_root, "cnt", cnt)), ("Style", sty_path)))
            else:
                cnt, char, back = names
                found = (("Content", os.path.join(data_root, "cnt", cnt)), ("Char", os.path.join(data_root, "sty/char", char)),
                         ("Back", os.path.join(data_root, "sty/back", back)))
            for what, pth in found:
                if not os.path.isfile(pth):
                    raise FileNotFoundError("%s image not found: %s" % (what, pth))
            paths = tuple(p for _, p in found)
            triples.append(paths if len(paths) == 3 else paths + (None,))
    return triples


def base(path):
    return os.path.splitext(os.path.basename(path))[0]


def load_unet(opt, device):
    from . import unet as U
    cfg = json.load(open(opt.unet_config)) if opt.unet_config else {}
    if opt.random_unet:
        torch.manual_seed(opt.seed)             # before construction: the constructor draws the biases
    model = U.UNetModel(**cfg)
    if opt.random_unet:
        for p in model.parameters():
            if p.dim() > 1:
                torch.nn.init.normal_(p, std=p[0].numel() ** -0.5)
    else:
        sd = torch.load(opt.ckpt, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        pre = "model.diffusion_model."
        model.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
    return model.eval().to(device)


def model_heads(model):
    mods = model.attention_modules()
    return mods[0].heads if mods else 8


def make_loader(opt, model, device, prefetch=None):
    """The feature-store loader of a run: packs for the model's head count and the run's arithmetic."""
    from .loader import FeatureLoader
    dtype = torch.bfloat16 if opt.feature_dtype == "bfloat16" else None
    packed = not opt.unpacked_store and not opt.without_attn_injection
    return FeatureLoader(opt.precomputed, device, heads=model_heads(model), compute=opt.compute, T=opt.T, dtype=dtype,
                         packed=packed, cache_content=packed and getattr(opt, "cache_content", False),
                         cache_bytes=int(getattr(opt, "store_cache_gb", 96.0) * (1 << 30)),
                         sidecar=getattr(opt, "sidecar", "use"),
                         prefetch=(not getattr(opt, "no_prefetch", False)) if prefetch is None else prefetch)


def bases_of(opt, triple):
    """File stems of a line's feature files.  A three-name line whose content image has no mask file cannot
    take the two-style path (the reference's run_ori.py:228 fails on the missing file): it is stylised with
    its character style alone, as a two-name line."""
    cnt_path, char_path, back_path = triple
    mask_file = os.path.join(opt.data_root, "cnt", base(cnt_path) + "_mask.npy")
    if back_path is not None and not os.path.isfile(mask_file):
        print("no mask file %s: %s is stylised with the single style %s" % (mask_file, base(cnt_path), base(char_path)))
        back_path = None
    if back_path is None and os.path.isfile(mask_file):
        raise ValueError("%s has a mask file: the mask-based attention needs a character and a background style "
                         "(three names per line); the single-style path is for content images without a mask "
                         "(hf_run.py:383-455)" % base(cnt_path))
    return (base(cnt_path), base(char_path), base(back_path) if back_path is not None else None)


def prepare(opt, model, triple, device, store):
    """Initial latent and merged features of one line (run_ori.py:224-235; hf_run.py:424-428 for a pair)."""
    from .features import adain, masked_adain_init
    cnt_path = triple[0]
    if store.back is None:                                             # single style: hf_run.py:424-428
        cnt_z, sty_z = store.z_enc()
        x_T = cnt_z if opt.without_init_adain else adain(cnt_z, sty_z)
    else:
        cnt_z, char_z, back_z = store.z_enc()
        if opt.without_init_adain:
            x_T = cnt_z
        else:
            mask = np.load(os.path.join(opt.data_root, "cnt", base(cnt_path) + "_mask.npy")).astype(np.float32)
            x_T = masked_adain_init(cnt_z, char_z, back_z, torch.from_numpy(mask).to(device))
    feats = None if opt.without_attn_injection else store.merged(opt, start_step=opt.start_step)
    return x_T, feats


def residual_path(opt, triple):
    return os.path.join(opt.precomputed, base(triple[0]) + "_residuals_all.pkl")


def stylise(opt, model, sampler, triple, uc, device, store=None):
    """One triple -> final latent (1, C, H/f, W/f).  `store`: its FeatureStore from a FeatureLoader (default:
    loaded here, synchronously)."""
    from . import feature_store as fs
    from .high_frequency import make_content_injection_schedule, patch_decoder_resblocks_h_and_cnt_hf
    cnt_path = triple[0]
    model.set_content_name(cnt_path)                                   # run_ori.py:159-161
    if getattr(opt, "block_skip", False):
        from . import ops
        from .attention import CrossAttention
        for m in model.modules():
            if isinstance(m, CrossAttention):
                m.kernel_flags |= ops.FLAG_BLOCK_SKIP      # kept per layer only where blocks actually die
                m.skip_policy = "auto"
    if store is None:
        loader = make_loader(opt, model, device, prefetch=False)
        store = loader.get(bases_of(opt, triple))
    x_T, feats = prepare(opt, model, triple, device, store)
    res_path = residual_path(opt, triple)
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


def stylise_lanes(opt, model, sampler, batch, uc, device, stores, lanes):
    """Several lines at once, one per lane (`lanes`: [(GraphedUNet, stream)]): final latents in line order."""
    from .high_frequency import unpatch_decoder_resblocks
    unpatch_decoder_resblocks(model)
    x_Ts, feats = [], []
    for (g, _), triple, store in zip(lanes, batch, stores):
        g.content_name = triple[0]                                     # every lane replays its own content's mask
        x_T, f = prepare(opt, model, triple, device, store)
        x_Ts.append(x_T)
        feats.append(f)
    n = len(batch)
    return sampler.sample_lanes([g for g, _ in lanes[:n]], x_Ts, [uc] * n, feats, start_step=opt.start_step,
                                streams=[st for _, st in lanes[:n]])


def run_lines(opt, model, sampler, triples, mine, uc, device):
    """The loop of run_ori.py:155-246 over this rank's lines `mine` (indices into `triples`): every line's
    feature files through the loader (read ahead, kept resident, side-cars), its latent saved under the
    reference's output name.  Returns (final latents in the order of `mine`, loader statistics)."""
    latents = []
    loader = make_loader(opt, model, device)
    keys = [bases_of(opt, triples[idx]) for idx in mine]
    n_lanes = max(1, getattr(opt, "lanes", 1))
    lanes = []
    if n_lanes > 1:
        from . import ops
        from .graphed import GraphedUNet
        lanes = [(GraphedUNet(model), torch.cuda.Stream(device=device)) for _ in range(n_lanes)]
        for a in model.attention_modules():
            a.kernel_flags |= ops.FLAG_THROUGHPUT          # the images share the GPU: least-work decomposition
    # batches of lines that go together: one per lane, a line with a residual file (high-frequency injection sets
    # per-step state on the shared network) by itself
    batches, j = [], 0
    while j < len(mine):
        take = 1
        if n_lanes > 1 and not os.path.isfile(residual_path(opt, triples[mine[j]])):
            while (take < n_lanes and j + take < len(mine)
                   and not os.path.isfile(residual_path(opt, triples[mine[j + take]]))):
                take += 1
        batches.append(list(range(j, j + take)))
        j += take

    def save(idx, z):
        latents.append(z[0].float())
        name = "_".join(base(p) for p in triples[idx] if p is not None) + ".pt"
        torch.save(z.cpu(), os.path.join(opt.output_path, name))      # run_ori.py:245 names the PNG the same way

    for j in (batches[0] if batches else []):
        loader.request(keys[j])
    import contextlib
    scope = (torch.autocast("cuda", dtype=torch.bfloat16) if getattr(opt, "network_autocast", False)
             else contextlib.nullcontext())
    with torch.no_grad(), scope:
        for bi, batch in enumerate(batches):
            stores = [loader.get(keys[j]) for j in batch]
            for j in (batches[bi + 1] if bi + 1 < len(batches) else []):
                loader.request(keys[j])                    # read, upload and pack while this batch samples
            if len(batch) == 1 and (n_lanes == 1 or os.path.isfile(residual_path(opt, triples[mine[batch[0]]]))):
                save(mine[batch[0]], stylise(opt, model, sampler, triples[mine[batch[0]]], uc, device, stores[0]))
            else:
                zs = stylise_lanes(opt, model, sampler, [triples[mine[j]] for j in batch], uc, device, stores, lanes)
                for j, z in zip(batch, zs):
                    save(mine[j], z)
            del stores
            if getattr(opt, "progress", False):
                sys.stderr.write("lines done: %d of %d\n" % (batch[-1] + 1, len(mine)))
                sys.stderr.flush()
    loader.close()
    return latents, dict(loader.stats)


def main(argv=None):
    opt = build_parser().parse_args(argv)
    if opt.lanes > 1:
        # images in flight want hardware queues of their own (the runtime's default of four is shared with the
        # loader's copy stream: measured 45 against 56 images/s for the attention path at four lanes); read by the
        # HIP runtime when it initialises, so it is set before anything touches the GPU.  An exported value wins.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
    latents, stats = run_lines(opt, model, sampler, triples, mine, uc, device)
    opt.loader_stats = stats
    shape = (opt.C, opt.H // opt.f, opt.W // opt.f)
    local_stack = torch.stack(latents) if latents else torch.zeros((0,) + shape, device=device)
    local_stack, _ = sharding.pad_to_common(local_stack, len(triples), world, per=max(len(l) for l in lists))
    if dist is not None and backend != "nccl":
        local_stack = local_stack.cpu()
    gathered = sharding.gather_latents(local_stack, dist, dst=0)
    if rank == 0:
        allz = sharding.reassemble_lists(gathered, lists)
        torch.save(allz.cpu(), os.path.join(opt.output_path, "latents_all.pt"))
        st = stats
        print("Total end: %.3f s for %d triples on %d GPU(s); rank 0 read %d files (%.2f GB, %d from side-cars), %d served from HBM"
              % (time.time() - begin, len(triples), world, st["files"], st["file_bytes"] / 1e9, st["routes"]["sidecar"], st["hits"]))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
