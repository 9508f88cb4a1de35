"""DDIM sampling loop around the UNet, as the reference drives it for stylisation
(ldm/models/diffusion/ddim.py:25-55 schedule, :155-240 loop, :286-322 update; run_ori.py:238-242):
eta = 0, no classifier-free guidance (the conditioning is the unconditional embedding only), one
UNet evaluation per step with that step's merged feature dict, and the steps whose index is
>= start_step are skipped (with start_step = 49 of 50 that is the first, noisiest step)."""
import numpy as np
import torch


def make_ddim_timesteps(num_ddim, num_ddpm=1000):
    """uniform discretisation, +1 (ldm/modules/diffusionmodules/util.py:46-60): 1, 21, ..., 981."""
    c = num_ddpm // num_ddim
    return (np.arange(0, num_ddim) * c).astype(int) + 1


class DDIMSampler:
    def __init__(self, unet, num_ddpm=1000, linear_start=0.00085, linear_end=0.012):
        self.unet = unet
        betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, num_ddpm, dtype=np.float64) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas)
        self.num_ddpm = num_ddpm

    def make_schedule(self, ddim_num_steps, ddim_eta=0.0):
        ts = make_ddim_timesteps(ddim_num_steps, self.num_ddpm)
        ac = self.alphas_cumprod
        a = ac[ts]
        a_prev = np.asarray([ac[0]] + ac[ts[:-1]].tolist())
        self.ddim_timesteps = ts
        self.ddim_alphas = a.astype(np.float32)
        self.ddim_alphas_prev = a_prev.astype(np.float32)
        self.ddim_sigmas = (ddim_eta * np.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))).astype(np.float32)
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1.0 - a).astype(np.float32)

    @torch.no_grad()
    def sample(self, x_T, unconditional_conditioning, injected_features=None, start_step=9999, callback=None,
               generator=None, img_callback=None):
        """x_T (B,4,h,w) -> final latent.  `injected_features`: list indexed by loop position i
        (i = 0 is the largest timestep), as feat_merge_2sty produces; `callback(i)` runs after step i
        (the high-frequency patch reads the timestep it sets), then `img_callback(pred_x0, img, step)`
        (ldm/models/diffusion/ddim.py:230-233)."""
        img = x_T
        total = len(self.ddim_timesteps)
        for i, step in enumerate(np.flip(self.ddim_timesteps)):
            index = total - i - 1
            if index >= start_step:
                continue
            feats = injected_features[i] if injected_features else None
            img, pred_x0 = self._step(self.unet, img, step, index, unconditional_conditioning, feats, generator)
            if callback is not None:
                callback(i)
            if img_callback is not None:
                img_callback(pred_x0, img, int(step))
        return img

    def _step(self, unet, img, step, index, cond, feats, generator=None):
        """One DDIM update x_t -> x_{t-1} (ldm/models/diffusion/ddim.py:286-322), scalars on the host."""
        ts = torch.full((img.shape[0],), int(step), device=img.device, dtype=torch.long)
        e_t = unet(img, ts, cond, injected_features=feats)
        a_t, a_prev = float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index])
        sigma_t = float(self.ddim_sigmas[index])
        pred_x0 = (img - float(self.ddim_sqrt_one_minus_alphas[index]) * e_t) / a_t ** 0.5
        dir_xt = (1.0 - a_prev - sigma_t ** 2) ** 0.5 * e_t
        out = a_prev ** 0.5 * pred_x0 + dir_xt
        if sigma_t > 0:
            out = out + sigma_t * torch.randn(out.shape, device=out.device, generator=generator)
        return out, pred_x0

    @torch.no_grad()
    def sample_lanes(self, unets, x_Ts, conds, features, start_step=9999, streams=None):
        """Several images in flight on one GPU: image l is sampled by `unets[l]` (its own GraphedUNet: own
        static slots, own workspace) on `streams[l]`, the steps of the images interleaved from this one host
        thread (nothing in a step waits for the GPU).  Images are independent (run_ori.py:155-246 stylises
        them one after the other); a batch-1 evaluation leaves most of the GPU idle, several at once fill it.
        Returns the final latents, ready on the calling stream."""
        n = len(x_Ts)
        dev = x_Ts[0].device
        streams = streams or [torch.cuda.Stream(device=dev) for _ in range(n)]
        cur = torch.cuda.current_stream(dev)
        imgs = list(x_Ts)
        for l in range(n):
            streams[l].wait_stream(cur)
            imgs[l].record_stream(streams[l])
        total = len(self.ddim_timesteps)
        for i, step in enumerate(np.flip(self.ddim_timesteps)):
            index = total - i - 1
            if index >= start_step:
                continue
            for l in range(n):
                with torch.cuda.stream(streams[l]):
                    feats = features[l][i] if features[l] else None
                    imgs[l], _ = self._step(unets[l], imgs[l], step, index, conds[l], feats)
        for l in range(n):
            cur.wait_stream(streams[l])
            imgs[l].record_stream(cur)
        return imgs

    @torch.no_grad()
    def encode_ddim(self, x0, unconditional_conditioning, num_steps=50, img_callback=None):
        """DDIM inversion x0 -> x_T (ldm/models/diffusion/ddim.py:324-384): steps t = 1, 21, ..., 981,
        each to the next one (the last to 999); `img_callback(pred_x0, x_next, t)` after every step."""
        T = self.num_ddpm
        c = T // num_steps
        steps = list(range(1, T + c, c))
        steps[-1] = T - 1
        ac = self.alphas_cumprod
        img, pred_x0 = x0, x0
        for i, t in enumerate(range(1, T, c)):
            ts = torch.full((x0.shape[0],), t, device=x0.device, dtype=torch.long)
            e_t = self.unet(img, ts, unconditional_conditioning)
            a_t, a_next = float(ac[t]), float(ac[steps[i + 1]])
            pred_x0 = (img - (1.0 - a_t) ** 0.5 * e_t) / a_t ** 0.5
            img = a_next ** 0.5 * pred_x0 + (1.0 - a_next) ** 0.5 * e_t
            if img_callback is not None:
                img_callback(pred_x0, img, t)
        return img, pred_x0


def last_resblocks(unet, blocks=range(6, 12)):
    """(block_id, module) of the last ResBlock of each output block in `blocks`: the ones the
    high-frequency injection reads and patches (high_frequency_final.py:53-58,
    generate_pkl_only.py:153-158)."""
    found = []
    for block_id in blocks:
        if block_id >= len(unet.output_blocks):
            break
        for module in reversed(unet.output_blocks[block_id]):
            if module.__class__.__name__.endswith("ResBlock"):
                found.append((block_id, module))
                break
    return found


def capture_residuals(unet, blocks=range(6, 12)):
    """The content residuals of the UNet evaluation that has just run, in the layout of one timestep of
    `{base}_residuals_all.pkl` (generate_pkl_only.py:153-178): `output_block_{i}_cnt_skip` = out_skip,
    `..._cnt_h` = out_h, `..._residual` = their sum, of the last ResBlock of output blocks 6-11, as CPU
    tensors."""
    entry = {}
    for block_id, module in last_resblocks(unet, blocks):
        if module.out_skip is None or module.out_h is None:
            continue
        entry["output_block_%d_cnt_skip" % block_id] = module.out_skip.detach().cpu()
        entry["output_block_%d_cnt_h" % block_id] = module.out_h.detach().cpu()
        entry["output_block_%d_residual" % block_id] = (module.out_skip + module.out_h).detach().cpu()
    return entry


def extract_features(unet, x0, unconditional_conditioning, num_steps=50, blocks=range(6, 12), T=1.5, residuals=None):
    """The producer of a `precomputed_feats_k` list (generate_pkl_only.py:90-142): DDIM inversion with the
    attention modules in feature-extraction mode; after the step at timestep t the q, k, v of the
    self-attention of output blocks 6-11 and the latent go to entry i = (981 - t) / 20, so entry 0
    holds the fully inverted latent.  With `residuals` a dict, every step also records the content
    residuals of its UNet evaluation under `residuals[int(t)]` (the content image's
    `residual_injection_callback`, generate_pkl_only.py:144-178): what `{base}_residuals_all.pkl` holds."""
    sampler = DDIMSampler(unet)
    sampler.make_schedule(num_steps, 0.0)
    index_of = {int(t): i for i, t in enumerate(np.flip(sampler.ddim_timesteps))}
    feats = [{"config": {"T": T}} for _ in range(num_steps)]
    attns = unet.attention_modules()
    for a in attns:
        a.gen_pkl = True

    def record(pred_x0, xt, t):
        e = feats[index_of[int(t)]]
        for b in blocks:
            a = unet.output_blocks[b][1].transformer_blocks[0].attn1
            for nm in "qkv":
                e["output_block_%d_self_attn_%s" % (b, nm)] = getattr(a, nm).detach().clone()
        e["z_enc"] = xt.detach().clone()
        if residuals is not None:
            residuals.setdefault(int(t), {}).update(capture_residuals(unet, blocks))

    try:
        sampler.encode_ddim(x0, unconditional_conditioning, num_steps, img_callback=record)
    finally:
        for a in attns:
            a.gen_pkl = False
    return feats
