"""Generate tests/golden/*.npz|json by running the REFERENCE's own modules.

Run in the build container only (it reads /root/reference, which does not exist on the GPU
box):   python tests/golden/make_goldens.py

The fixtures hold data only: seeds/parameters of the inputs (regenerated anywhere with
style_transfer_amd.synth), the reference's outputs, and the reference's mask files packed
to bits.  Harness adjustments, none of which edits a reference file:
  * torch.Tensor.cuda is made a no-op: ldm/modules/attention.py:270 hard-codes .cuda()
  * the mask is written to <tmp>/x_mask.npy and cnt_name='<tmp>/x.jpg', the path the
    reference derives at ldm/modules/attention.py:410-411
  * to_q / to_out are set to identity (or to seeded weights in the 'module' cases) so the
    fixture pins the attention arithmetic itself
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from style_transfer_amd import synth  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
torch.manual_seed(0)
torch.set_num_threads(8)

import ldm.modules.attention as ref_attn  # noqa: E402
import high_frequency_final as ref_hf  # noqa: E402
from ldm.modules.diffusionmodules.util import make_ddim_timesteps  # noqa: E402

MASK_DIR = os.path.join(REF, "dataset_ours", "mask")
MASK_NAMES = sorted(f[:-4] for f in os.listdir(MASK_DIR) if f.endswith(".npy"))


def ref_mask_downsample(mask512, r_h, r_w, ch):
    """Literal replay of ldm/modules/attention.py:270-283 for one mask."""
    mask = torch.tensor(mask512, dtype=torch.float32).cuda()
    mask[mask < 0.5] = -1.0
    mask[mask > 0.5] = 1.0
    mask = mask * ch
    mask = mask.unsqueeze(0).unsqueeze(0)
    mask = torch.nn.functional.interpolate(mask, size=(r_h, r_w), mode="bilinear", align_corners=False)
    return mask.reshape(-1).numpy().copy()


def gen_masks():
    packed, down = {}, {}
    for nm in MASK_NAMES:
        m = np.load(os.path.join(MASK_DIR, nm + ".npy"))
        assert m.shape == (512, 512) and m.dtype == np.float32
        assert set(np.unique(m)) <= {0.0, 1.0}
        packed[nm] = np.packbits(m.astype(np.uint8))
        for r in (128, 64, 32, 16, 8):
            pos = ref_mask_downsample(m, r, r, 1.0)
            neg = ref_mask_downsample(m, r, r, -1.0)
            assert np.array_equal(neg, -pos)
            down[f"{nm}/{r}"] = pos
    dv = np.load(os.path.join(REF, "data_vis", "cnt", "content_flower_mask.npy"))
    assert np.array_equal(dv, np.load(os.path.join(MASK_DIR, "content_flower_mask.npy")))
    np.savez_compressed(os.path.join(HERE, "masks_packed.npz"), **packed)
    np.savez_compressed(os.path.join(HERE, "mask_downsample.npz"), **down)
    # general bilinear weights: grey, non-power-of-two, non-square
    gen = {}
    specs = [(500, 500, 64, 64, 11), (512, 512, 48, 48, 12), (300, 420, 37, 53, 13), (512, 512, 96, 96, 14),
             (64, 64, 64, 64, 15), (40, 40, 64, 64, 16)]
    for (ih, iw, oh, ow, seed) in specs:
        g = synth.uniform(seed, (ih, iw))
        g[::7, ::5] = 0.5          # exact 0.5 stays 0.5 (neither comparison fires)
        for ch in (1.0, -1.0):
            gen[f"{ih}x{iw}_{oh}x{ow}_s{seed}_ch{int(ch)}"] = ref_mask_downsample(g, oh, ow, ch)
    np.savez_compressed(os.path.join(HERE, "mask_general.npz"), **gen)
    print("masks:", len(packed), "files;", len(down), "downsamples;", len(gen), "general")


def mask_for(kind):
    if kind == "disc":
        return synth.disc_mask()
    if kind == "half":
        return synth.half_plane_mask()
    return np.load(os.path.join(MASK_DIR, kind + ".npy"))


def make_module(heads, d, identity=True, wseed=0):
    mod = ref_attn.CrossAttention(query_dim=heads * d, heads=heads, dim_head=d)
    C = heads * d
    with torch.no_grad():
        if identity:
            mod.to_q.weight.copy_(torch.eye(C))
            mod.to_k.weight.copy_(torch.eye(C))
            mod.to_v.weight.copy_(torch.eye(C))
            mod.to_out[0].weight.copy_(torch.eye(C))
            mod.to_out[0].bias.zero_()
        else:
            s = 1.0 / np.sqrt(C)
            mod.to_q.weight.copy_(torch.from_numpy(synth.normal(wseed + 1, (C, C), s)))
            mod.to_k.weight.copy_(torch.from_numpy(synth.normal(wseed + 2, (C, C), s)))
            mod.to_v.weight.copy_(torch.from_numpy(synth.normal(wseed + 3, (C, C), s)))
            mod.to_out[0].weight.copy_(torch.from_numpy(synth.normal(wseed + 4, (C, C), s)))
            mod.to_out[0].bias.copy_(torch.from_numpy(synth.normal(wseed + 5, (C,), 0.1)))
    return mod.eval()


def run_case(tmp, case):
    H, N, d = case["heads"], case["N"], case["d"]
    t = synth.attention_inputs(case["seed"], H, N, d, case["sigma_qk"], case["sigma_v"])
    T = {k: torch.from_numpy(v) for k, v in t.items()}
    mod = make_module(H, d, identity=not case.get("module_weights", False), wseed=case["seed"] * 100)
    mask_kind = case["mask"]
    base = os.path.join(tmp, f"c{case['id']}")
    if mask_kind != "none":
        np.save(base + "_mask.npy", mask_for(mask_kind))
    mod.cnt_name = base + ".jpg"
    cfg = None if case["T"] is None else {"T": case["T"], "gamma": case["gamma"]}
    x = T["xq"]                       # (1, N, H*d); with identity to_q this IS to_q(x)
    branch = case["branch"]
    with torch.no_grad():
        if branch == "injected":
            out = mod(x, q_injected=T["q_cnt"], k_injected=T["k1"], v_injected=T["v1"],
                      cnt_k_injected=T["k_cnt"], sty_q_injected=None, cnt_v_injected=T["v_cnt"],
                      sty2_q_injected=None, sty2_k_injected=T["k2"], sty2_v_injected=T["v2"],
                      injection_config=cfg)
        elif branch == "styleid_nomask":      # attention.py:569-579, q,k,v injected, no mask file
            out = mod(x, q_injected=T["q_cnt"], k_injected=T["k1"], v_injected=T["v1"], injection_config=cfg)
        elif branch in ("tplain", "plain"):   # attention.py:556-566 (mask present) / 569-579 (none)
            out = mod(x, injection_config=cfg)
        elif branch == "genpkl":              # attention.py:584-604
            mod.gen_pkl = True
            out = mod(x)
        else:
            raise ValueError(branch)
    return out[0].numpy()


def attn_cases():
    cases = []
    cid = 0
    shapes = [(64, 160), (256, 160), (1024, 80), (4096, 40)]
    combos = [(0.5, 1.5, 1.0, "content_flower_mask"), (0.75, 1.5, 2.0, "content_flower_mask"),
              (0.5, 1.0, 1.0, "disc"), (0.75, 1.0, 2.0, "half"), (0.75, 1.5, 1.0, "content_car_mask")]
    for (N, d) in shapes:
        for (gamma, T, sig, mk) in combos:
            if N == 4096 and mk == "content_car_mask":
                continue
            cases.append(dict(id=cid, branch="injected", heads=8, N=N, d=d, gamma=gamma, T=T,
                              sigma_qk=sig, sigma_v=1.0, mask=mk, seed=100 + cid)); cid += 1
    # other branches
    for (N, d) in [(64, 160), (256, 160), (1024, 80)]:
        cases.append(dict(id=cid, branch="tplain", heads=8, N=N, d=d, gamma=0.75, T=1.5, sigma_qk=1.0,
                          sigma_v=1.0, mask="content_flower_mask", seed=100 + cid)); cid += 1
        cases.append(dict(id=cid, branch="plain", heads=8, N=N, d=d, gamma=None, T=None, sigma_qk=1.0,
                          sigma_v=1.0, mask="none", seed=100 + cid)); cid += 1
        cases.append(dict(id=cid, branch="styleid_nomask", heads=8, N=N, d=d, gamma=0.75, T=1.5, sigma_qk=1.5,
                          sigma_v=1.0, mask="none", seed=100 + cid)); cid += 1
    cases.append(dict(id=cid, branch="genpkl", heads=8, N=256, d=160, gamma=None, T=None, sigma_qk=1.0,
                      sigma_v=1.0, mask="content_flower_mask", seed=100 + cid)); cid += 1
    # whole-module cases (seeded to_q / to_out weights): pins the drop-in class end to end
    cases.append(dict(id=cid, branch="injected", heads=8, N=256, d=40, gamma=0.75, T=1.5, sigma_qk=1.0,
                      sigma_v=1.0, mask="content_flower_mask", seed=100 + cid, module_weights=True)); cid += 1
    cases.append(dict(id=cid, branch="tplain", heads=8, N=256, d=40, gamma=0.75, T=1.5, sigma_qk=1.0,
                      sigma_v=1.0, mask="content_flower_mask", seed=100 + cid, module_weights=True)); cid += 1
    # odd head counts / sizes the kernels must not silently mishandle
    cases.append(dict(id=cid, branch="injected", heads=4, N=144, d=24, gamma=0.6, T=2.0, sigma_qk=1.0,
                      sigma_v=1.0, mask="disc", seed=100 + cid)); cid += 1
    cases.append(dict(id=cid, branch="injected", heads=2, N=100, d=64, gamma=0.3, T=0.5, sigma_qk=1.0,
                      sigma_v=1.0, mask="half", seed=100 + cid)); cid += 1
    # single-block attention at the 64x64 level (input blocks 1, 2 of the UNet): 4096 keys, head_dim 40
    cases.append(dict(id=cid, branch="tplain", heads=8, N=4096, d=40, gamma=0.75, T=1.5, sigma_qk=1.0,
                      sigma_v=1.0, mask="content_flower_mask", seed=100 + cid)); cid += 1
    cases.append(dict(id=cid, branch="plain", heads=8, N=4096, d=40, gamma=None, T=None, sigma_qk=1.0,
                      sigma_v=1.0, mask="none", seed=100 + cid)); cid += 1
    cases.append(dict(id=cid, branch="tplain", heads=8, N=4096, d=40, gamma=0.5, T=1.0, sigma_qk=2.0,
                      sigma_v=1.0, mask="content_flower_mask", seed=100 + cid)); cid += 1
    cases.append(dict(id=cid, branch="styleid_nomask", heads=8, N=4096, d=40, gamma=0.75, T=1.5, sigma_qk=1.5,
                      sigma_v=1.0, mask="none", seed=100 + cid)); cid += 1
    # feature-like magnitudes (q, k three times unit scale: raw q.k reaches the hundreds, so does the
    # mask bias) with EVERY fractional (mask-edge) row of the real mask among the stored rows
    cases.append(dict(id=cid, branch="injected", heads=8, N=1024, d=80, gamma=0.75, T=1.5, sigma_qk=3.0,
                      sigma_v=1.0, mask="content_flower_mask", seed=100 + cid, all_frac=True)); cid += 1
    cases.append(dict(id=cid, branch="injected", heads=8, N=4096, d=40, gamma=0.5, T=1.5, sigma_qk=3.0,
                      sigma_v=1.0, mask="content_flower_mask", seed=100 + cid, all_frac=True)); cid += 1
    cases.append(dict(id=cid, branch="injected", heads=8, N=4096, d=40, gamma=0.75, T=1.5, sigma_qk=3.0,
                      sigma_v=1.0, mask="content_car_mask", seed=100 + cid, all_frac=True)); cid += 1
    return cases


def gen_attention():
    cases = attn_cases()
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        for c in cases:
            out = run_case(tmp, c)             # (N, H*d)
            N = c["N"]
            if N <= 144:
                rows = np.arange(N)
            else:
                # sampled rows: always include mask-boundary rows when there are any
                rows = np.unique(np.concatenate([
                    (synth.hash_u64(c["seed"], 24) % np.uint64(N)).astype(np.int64),
                    np.array([0, N - 1, N // 2])]))
                if c["mask"] != "none":
                    side = int(round(np.sqrt(N)))
                    m = ref_mask_downsample(mask_for(c["mask"]), side, side, 1.0)
                    frac = np.nonzero(np.abs(m) != 1.0)[0]
                    rows = np.unique(np.concatenate([rows, frac if c.get("all_frac") else frac[:8]]))
            c["rows"] = [int(r) for r in rows]
            arrays[f"out_rows/{c['id']}"] = out[rows].astype(np.float32)
            arrays[f"colsum/{c['id']}"] = out.astype(np.float64).sum(axis=0)
            arrays[f"colsqsum/{c['id']}"] = (out.astype(np.float64) ** 2).sum(axis=0)
            print("attn case", c["id"], c["branch"], c["N"], c["d"], "rows", len(rows),
                  "absmax %.3f" % np.abs(out).max())
    np.savez_compressed(os.path.join(HERE, "attention.npz"), **arrays)
    with open(os.path.join(HERE, "attention_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)


def gen_hf():
    arrays, cases = {}, []
    cid = 0
    for (B, C, Hh, Ww) in [(1, 4, 8, 8), (1, 4, 16, 16), (1, 8, 32, 32), (1, 8, 64, 64), (1, 4, 128, 128),
                           (2, 3, 32, 32), (1, 2, 16, 64), (1, 2, 64, 32)]:
        for ratio in (0.25, 0.5, 0.75):
            seed = 500 + cid
            h = synth.normal(seed, (B, C, Hh, Ww))
            out = ref_hf.high_freq_filter(torch.from_numpy(h), radius_ratio=ratio).numpy()
            arrays[f"hf/{cid}"] = out.astype(np.float32)
            cases.append(dict(id=cid, shape=[B, C, Hh, Ww], ratio=ratio, seed=seed))
            cid += 1
    # zeroed-bin counts at ratio 0.5 (derived from the reference's mask construction)
    counts = {}
    for r in (8, 16, 32, 64, 128):
        cy = cx = r // 2
        radius = int(r * 0.5)
        y = torch.arange(r).view(-1, 1); x = torch.arange(r).view(1, -1)
        counts[str(r)] = int(((y - cy) ** 2 + (x - cx) ** 2 < radius ** 2).sum())
    np.savez_compressed(os.path.join(HERE, "hf_filter.npz"), **arrays)
    sched50 = make_ddim_timesteps("uniform", 50, 1000, verbose=False)
    misc = dict(
        hf_cases=cases, zeroed_bins_ratio_0p5=counts,
        ddim_timesteps_50=[int(t) for t in sched50],
        ddim_timesteps_20=[int(t) for t in make_ddim_timesteps("uniform", 20, 1000, verbose=False)],
        content_injection_schedule_50=[int(t) for t in ref_hf.make_content_injection_schedule(sched50)],
        content_injection_schedule_50_a25=[int(t) for t in ref_hf.make_content_injection_schedule(sched50, 0.25)],
    )
    with open(os.path.join(HERE, "hf_cases.json"), "w") as f:
        json.dump(misc, f, indent=1)
    print("hf:", len(cases), "cases")


def gen_feat_merge():
    """run_ori.py cannot be imported here (omegaconf / pytorch_lightning are absent), so only its
    function `feat_merge_2sty` is compiled from the file's syntax tree and run on tagged
    placeholder 'tensors'; the fixture records which source each output key refers to."""
    import ast
    import types
    src = open(os.path.join(REF, "run_ori.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "feat_merge_2sty"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "run_ori.py", "exec"), ns)
    keys = ["z_enc", "config"] + [f"output_block_{b}_self_attn_{t}" for b in range(6, 12) for t in "qkv"]
    mk = lambda tag: [{k: f"{tag}[{i}].{k}" for k in keys} for i in range(50)]
    opt = types.SimpleNamespace(gamma=0.75, T=1.5)
    out = {}
    for start_step in (49, 30, 50, 0):
        merged = ns["feat_merge_2sty"](opt, mk("cnt"), mk("char"), mk("back"), start_step=start_step)
        out[str(start_step)] = merged
    with open(os.path.join(HERE, "feat_merge.json"), "w") as f:
        json.dump(out, f)
    print("feat_merge: steps", {k: sum(len(d) > 1 for d in v) for k, v in out.items()})
    # the single-style merge of hf_run.py:40-68, compiled the same way
    src = open(os.path.join(REF, "hf_run.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "feat_merge"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "hf_run.py", "exec"), ns)
    single = {str(ss): ns["feat_merge"](opt, mk("cnt"), mk("sty"), start_step=ss) for ss in (49, 30, 50, 0)}
    with open(os.path.join(HERE, "feat_merge_single.json"), "w") as f:
        json.dump(single, f)


def gen_adain():
    """`adain` is compiled from run_ori.py's syntax tree (the file itself cannot be imported here); the
    mask blend of run_ori.py:228-232 is restated with the same torch calls on real masks and seeded
    latents.  Fixture: the blended initial latents and the plain adain outputs."""
    import ast
    import torch
    import torch.nn.functional as F
    from style_transfer_amd import synth
    src = open(os.path.join(REF, "run_ori.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "adain"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "run_ori.py", "exec"), ns)
    adain = ns["adain"]
    masks = np.load(os.path.join(HERE, "masks_packed.npz"))
    out, cases = {}, []
    names = sorted(masks.files)[:3]
    for ci, (shape, mname, sig) in enumerate([((1, 4, 64, 64), names[0], 1.0), ((1, 4, 64, 64), names[1], 3.0),
                                              ((2, 4, 32, 32), names[2], 1.0), ((1, 8, 16, 48), names[0], 0.5)]):
        bits = np.unpackbits(masks[mname]).reshape(512, 512).astype(np.float32)
        seeds = [7000 + 10 * ci + j for j in range(3)]
        cnt, char, back = (torch.from_numpy(synth.normal(sd, shape, sigma=sig) + np.float32(0.3 * j))
                           for j, sd in enumerate(seeds))
        m = torch.tensor(bits, dtype=torch.float32).unsqueeze(0).unsqueeze(0)
        m = F.interpolate(m, size=(shape[2], shape[3]), mode="bilinear", align_corners=False)
        m = m.expand(-1, shape[1], -1, -1)
        blended = m * adain(cnt, char) + (1 - m) * adain(cnt, back)
        out[f"blend/{ci}"] = blended.numpy()
        out[f"adain/{ci}"] = adain(cnt, char).numpy()
        cases.append({"id": ci, "shape": list(shape), "mask": mname, "sigma": sig, "seeds": seeds})
    np.savez_compressed(os.path.join(HERE, "adain.npz"), **out)
    with open(os.path.join(HERE, "adain_cases.json"), "w") as f:
        json.dump(cases, f)
    print("adain:", len(cases), "cases")


UNET_CFG = dict(in_channels=4, out_channels=4, model_channels=160, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                channel_mult=[1, 2, 4, 4], num_heads=8, transformer_depth=1, context_dim=768)


def unet_inputs(tag):
    """Seeded latent, timestep and text context shared by the generator and the tests."""
    base = {"cnt": 8100, "char": 8200, "back": 8300, "cur": 8400}[tag]
    x = synth.normal(base, (1, 4, 16, 16))
    ctx = synth.normal(8500, (1, 77, 768), sigma=0.5)
    return x, ctx


def gen_unet():
    """Half-width SD-v1 topology (head dims 20/40/80, 12 output blocks) built from the REFERENCE's
    UNetModel with name-seeded weights.  Three feature-extraction passes (gen_pkl=True, as
    generate_pkl_only.py:90-134 does) give content / character / background q,k,v of output blocks
    6-11; feat_merge_2sty (compiled from run_ori.py) merges them; one injected forward with a real
    mask file gives the output this fixture pins, with checksums of the captured features."""
    import ast
    import types
    stub = types.ModuleType("omegaconf")
    stub.listconfig = types.ModuleType("omegaconf.listconfig")
    stub.listconfig.ListConfig = list
    sys.modules["omegaconf"], sys.modules["omegaconf.listconfig"] = stub, stub.listconfig
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    torch.manual_seed(0)
    model = UNetModel(image_size=32, use_spatial_transformer=True, use_checkpoint=False, legacy=False, **UNET_CFG).eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(synth.param_init(name, p.shape)))
    attns = [m for m in model.modules() if isinstance(m, ref_attn.CrossAttention)]
    t = torch.tensor([481])
    out = {}

    def capture(tag):
        for m in attns:
            m.gen_pkl = True
        x, ctx = unet_inputs(tag)
        with torch.no_grad():
            model(torch.from_numpy(x), t, torch.from_numpy(ctx))
        feats = {"config": {"T": 1.5}, "z_enc": torch.from_numpy(x)}
        for bi in range(6, 12):
            a = model.output_blocks[bi][1].transformer_blocks[0].attn1
            for nm in "qkv":
                feats[f"output_block_{bi}_self_attn_{nm}"] = getattr(a, nm).clone()
        for m in attns:
            m.gen_pkl = False
        return feats

    feats = {tag: capture(tag) for tag in ("cnt", "char", "back")}
    for tag, f in feats.items():
        for k, v in f.items():
            if torch.is_tensor(v) and k != "z_enc":
                out[f"feat/{tag}/{k}"] = np.array([v.double().sum().item(), (v.double() ** 2).sum().item()])
    src = open(os.path.join(REF, "run_ori.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "feat_merge_2sty"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "run_ori.py", "exec"), ns)
    opt = types.SimpleNamespace(gamma=0.75, T=1.5)
    merged = ns["feat_merge_2sty"](opt, [feats["cnt"]] * 50, [feats["char"]] * 50, [feats["back"]] * 50, start_step=50)[0]
    mask = np.load(os.path.join(MASK_DIR, MASK_NAMES[0] + ".npy"))
    x, ctx = unet_inputs("cur")
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "img_mask.npy"), mask)
        for m in attns:
            m.cnt_name = os.path.join(td, "img.jpg")
        with torch.no_grad():
            out["injected"] = model(torch.from_numpy(x), t, torch.from_numpy(ctx), injected_features=merged).numpy()
        for m in attns:
            m.cnt_name = os.path.join(td, "nomask.jpg")
        with torch.no_grad():
            out["plain"] = model(torch.from_numpy(x), t, torch.from_numpy(ctx)).numpy()
    np.savez_compressed(os.path.join(HERE, "unet.npz"), **out)
    with open(os.path.join(HERE, "unet_case.json"), "w") as f:
        json.dump({"cfg": UNET_CFG, "mask": MASK_NAMES[0], "t": 481, "gamma": 0.75, "T": 1.5,
                   "n_params": sum(p.numel() for p in model.parameters()),
                   "param_shapes": {n: list(p.shape) for n, p in model.named_parameters()}}, f)
    print("unet: params", sum(p.numel() for p in model.parameters()), "out std", out["injected"].std(), out["plain"].std(),
          "diff", np.abs(out["injected"] - out["plain"]).max())


def build_ref_unet():
    """The reference's UNetModel, half width, with name-seeded weights (see gen_unet)."""
    import types
    stub = types.ModuleType("omegaconf")
    stub.listconfig = types.ModuleType("omegaconf.listconfig")
    stub.listconfig.ListConfig = list
    sys.modules["omegaconf"], sys.modules["omegaconf.listconfig"] = stub, stub.listconfig
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    model = UNetModel(image_size=32, use_spatial_transformer=True, use_checkpoint=False, legacy=False, **UNET_CFG).eval()
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(synth.param_init(name, p.shape)))
    return model


def last_resblocks(model):
    """(block_id, module) of the last ResBlock of output blocks 6..11, found the way
    high_frequency_final.py:53-58 and generate_pkl_only.py:153-158 find it."""
    found = []
    for block_id in range(6, 12):
        for module in reversed(model.output_blocks[block_id]):
            if module.__class__.__name__.endswith("ResBlock"):
                found.append((block_id, module))
                break
    return found


HF_PATCH = dict(residual_steps=[21, 381, 401], unet_t=481, ratios=[0.5, 0.25])


def gen_hf_patch():
    """The high-frequency injection end to end on the reference's own code: content residuals captured
    per timestep exactly as generate_pkl_only.py:144-178 does (out_skip / out_h / out_merged of the last
    ResBlock of output blocks 6-11 after a forward on the content latent), the decoder patched by the
    reference's patch_decoder_resblocks_h_and_cnt_hf with the reference's schedule, `ri_timestep` set
    on those ResBlocks as hf_run.py:256-270 does, then an injected forward.  Pinned: checksums of the
    residuals, and the UNet output for ri_timestep = 21 (scheduled, stored: filter applied), 401
    (stored, not scheduled) and 1 (scheduled, not stored): the last two must equal the unpatched
    output."""
    import ast
    import types
    model = build_ref_unet()
    attns = [m for m in model.modules() if isinstance(m, ref_attn.CrossAttention)]
    out = {}
    ctx = torch.from_numpy(unet_inputs("cnt")[1])
    # content residuals at three "timesteps" (the content latent is re-seeded per step so they differ)
    residuals_all = {}
    with tempfile.TemporaryDirectory() as td:
        for m in attns:
            m.cnt_name = os.path.join(td, "nomask.jpg")
        for t_int in HF_PATCH["residual_steps"]:
            x = torch.from_numpy(synth.normal(8100 + t_int, (1, 4, 16, 16)))
            with torch.no_grad():
                model(x, torch.tensor([t_int]), ctx)
            residuals_all[t_int] = {}
            for block_id, module in last_resblocks(model):
                residuals_all[t_int][f"output_block_{block_id}_cnt_skip"] = module.out_skip.detach().cpu()
                residuals_all[t_int][f"output_block_{block_id}_cnt_h"] = module.out_h.detach().cpu()
                residuals_all[t_int][f"output_block_{block_id}_residual"] = module.out_merged.detach().cpu()
            for k, v in residuals_all[t_int].items():
                out[f"res/{t_int}/{k}"] = np.array([v.double().sum().item(), (v.double() ** 2).sum().item()])
        # merged injected features, as gen_unet builds them
        feats = {}
        for tag in ("cnt", "char", "back"):
            for m in attns:
                m.gen_pkl = True
            x, c2 = unet_inputs(tag)
            with torch.no_grad():
                model(torch.from_numpy(x), torch.tensor([HF_PATCH["unet_t"]]), torch.from_numpy(c2))
            f = {"config": {"T": 1.5}, "z_enc": torch.from_numpy(x)}
            for bi in range(6, 12):
                a = model.output_blocks[bi][1].transformer_blocks[0].attn1
                for nm in "qkv":
                    f[f"output_block_{bi}_self_attn_{nm}"] = getattr(a, nm).clone()
            for m in attns:
                m.gen_pkl = False
            feats[tag] = f
        src = open(os.path.join(REF, "run_ori.py")).read()
        fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "feat_merge_2sty"][0]
        ns = {}
        exec(compile(ast.Module(body=[fn], type_ignores=[]), "run_ori.py", "exec"), ns)
        opt = types.SimpleNamespace(gamma=0.75, T=1.5)
        merged = ns["feat_merge_2sty"](opt, [feats["cnt"]] * 50, [feats["char"]] * 50, [feats["back"]] * 50, start_step=50)[0]
        np.save(os.path.join(td, "img_mask.npy"), np.load(os.path.join(MASK_DIR, MASK_NAMES[0] + ".npy")))
        for m in attns:
            m.cnt_name = os.path.join(td, "img.jpg")
        x, c2 = unet_inputs("cur")
        xt, ct, tt = torch.from_numpy(x), torch.from_numpy(c2), torch.tensor([HF_PATCH["unet_t"]])
        with torch.no_grad():
            out["unpatched"] = model(xt, tt, ct, injected_features=merged).numpy()
        sched50 = make_ddim_timesteps("uniform", 50, 1000, verbose=False)
        schedule = ref_hf.make_content_injection_schedule(sched50, alpha=0.4)
        import contextlib
        import io
        for ratio in HF_PATCH["ratios"]:
            fresh = build_ref_unet()
            for m in fresh.modules():
                if isinstance(m, ref_attn.CrossAttention):
                    m.cnt_name = os.path.join(td, "img.jpg")
            ref_hf.patch_decoder_resblocks_h_and_cnt_hf(fresh, schedule, residuals_all, ratio=ratio)
            for ri in (21, 401, 1):
                for _, module in last_resblocks(fresh):          # hf_run.py:263-270
                    module.ri_timestep = int(ri)
                with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                    out[f"patched/r{ratio}/ri{ri}"] = fresh(xt, tt, ct, injected_features=merged).numpy()
            assert np.array_equal(out[f"patched/r{ratio}/ri401"], out["unpatched"])
            assert np.array_equal(out[f"patched/r{ratio}/ri1"], out["unpatched"])
            assert np.abs(out[f"patched/r{ratio}/ri21"] - out["unpatched"]).max() > 1e-2
    np.savez_compressed(os.path.join(HERE, "hf_patch.npz"), **out)
    with open(os.path.join(HERE, "hf_patch_case.json"), "w") as f:
        json.dump(dict(HF_PATCH, schedule=[int(t) for t in schedule]), f)
    print("hf_patch: diff patched vs unpatched", np.abs(out["patched/r0.5/ri21"] - out["unpatched"]).max(),
          "r0.25 vs r0.5", np.abs(out["patched/r0.5/ri21"] - out["patched/r0.25/ri21"]).max())


def gen_ddim():
    """The reference DDIMSampler (ldm/models/diffusion/ddim.py) driven by a stub model that carries the
    SD-v1 noise schedule (ddpm.py:117-146 with linear_start 0.00085, linear_end 0.012 of
    v1-inference.yaml) and a closed-form noise prediction (synth.stub_eps).  Pinned: the ddim_* tables
    for 50 and 5 steps at eta 0 and 0.5, the encode_ddim trajectory (every callback latent) and the
    sample() result with and without the skipped first step.  The only harness change: register_buffer
    keeps tensors where they are (the reference moves them to "cuda", ddim.py:19-23)."""
    import contextlib
    import io
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.util import make_beta_schedule

    class Stub:
        def __init__(self):
            betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.0120, cosine_s=8e-3)
            ac = np.cumprod(1.0 - betas, axis=0)
            f32 = lambda a: torch.tensor(a, dtype=torch.float32)
            self.betas, self.alphas_cumprod = f32(betas), f32(ac)
            self.alphas_cumprod_prev = f32(np.append(1.0, ac[:-1]))
            self.sqrt_alphas_cumprod = f32(np.sqrt(ac))
            self.sqrt_one_minus_alphas_cumprod = f32(np.sqrt(1.0 - ac))
            self.num_timesteps = 1000
            self.device = torch.device("cpu")
            self.calls = []

        def apply_model(self, x, t, c, injected_features=None):
            self.calls.append((int(t[0]), None if injected_features is None else injected_features["tag"]))
            return synth.stub_eps(x, t, c, injected_features)

    class Sampler(DDIMSampler):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    out, meta = {}, {}
    quiet = contextlib.redirect_stdout(io.StringIO())
    for steps in (50, 5):
        for eta in (0.0, 0.5):
            s = Sampler(Stub())
            with quiet:
                s.make_schedule(ddim_num_steps=steps, ddim_eta=eta, verbose=False)
            for nm in ("ddim_sigmas", "ddim_alphas", "ddim_alphas_prev", "ddim_sqrt_one_minus_alphas"):
                out[f"tab/{steps}/{eta}/{nm}"] = np.asarray(getattr(s, nm), dtype=np.float64)
            out[f"tab/{steps}/{eta}/ddim_timesteps"] = np.asarray(s.ddim_timesteps)
        x0 = torch.from_numpy(synth.normal(9100 + steps, (1, 4, 8, 8)))
        uc = torch.from_numpy(synth.normal(9200, (1, 77, 16), sigma=0.5))
        stub = Stub()
        s = Sampler(stub)
        with quiet:
            s.make_schedule(ddim_num_steps=steps, ddim_eta=0.0, verbose=False)
        traj = []
        with quiet, contextlib.redirect_stderr(io.StringIO()):
            z, pred = s.encode_ddim(x0.clone(), num_steps=steps, unconditional_conditioning=uc, end_step=981,
                                    callback_ddim_timesteps=steps,
                                    img_callback=lambda p, xt, t: traj.append((int(t), xt.clone(), p.clone())))
        meta[f"encode_t/{steps}"] = [t for t, _, _ in traj]
        meta[f"encode_calls/{steps}"] = [c[0] for c in stub.calls]
        out[f"encode/{steps}/xt"] = torch.stack([a for _, a, _ in traj]).numpy()
        out[f"encode/{steps}/pred_x0"] = torch.stack([b for _, _, b in traj]).numpy()
        out[f"encode/{steps}/final"] = z.numpy()
        feats = [{"tag": i, "shift": 0.01 * i} for i in range(steps)]
        for start_step in (steps - 1, 9999):
            stub = Stub()
            s = Sampler(stub)
            seen = []
            with quiet, contextlib.redirect_stderr(io.StringIO()):
                samples, _ = s.sample(S=steps, batch_size=1, shape=[4, 8, 8], verbose=False, unconditional_conditioning=uc,
                                      eta=0.0, x_T=z.clone(), injected_features=feats, start_step=start_step,
                                      callback=lambda i: seen.append(int(i)))
            out[f"sample/{steps}/{start_step}"] = samples.numpy()
            meta[f"sample_calls/{steps}/{start_step}"] = [list(c) for c in stub.calls]
            meta[f"sample_callback_i/{steps}/{start_step}"] = seen
    np.savez_compressed(os.path.join(HERE, "ddim.npz"), **out)
    with open(os.path.join(HERE, "ddim_case.json"), "w") as f:
        json.dump(meta, f)
    print("ddim: tables", sum(k.startswith("tab/") for k in out), "trajectories", sum(k.startswith("encode/") for k in out))


def gen_loop():
    """The whole stylisation loop of run_ori.py:224-242 run with the REFERENCE's own pieces: its UNetModel
    (half width, name-seeded weights) as `apply_model` under its DDIMSampler.sample for all 49 evaluations
    of a 50-step schedule (start_step 49), features merged by its feat_merge_2sty, the initial latent from
    its adain and the mask blend of run_ori.py:228-232, a real mask file on every attention module.
    Features: for every loop position i the q, k, v of output blocks 6-11 captured in gen_pkl mode
    (generate_pkl_only.py:113-142) from one forward of the seeded content / character / background latent at
    that position's timestep, so that the test can regenerate them with its own network one forward at a
    time.  Pinned: the final latent, the latents after 16 and 32 evaluations, checksums of the features."""
    import ast
    import contextlib
    import io
    import types
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.util import make_beta_schedule
    model = build_ref_unet()
    attns = [m for m in model.modules() if isinstance(m, ref_attn.CrossAttention)]

    class Wrapper:                       # LatentDiffusion as the sampler sees it (ddpm.py:117-146, :892-1006, :1430-1435)
        def __init__(self):
            betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.0120, cosine_s=8e-3)
            ac = np.cumprod(1.0 - betas, axis=0)
            f32 = lambda a: torch.tensor(a, dtype=torch.float32)
            self.betas, self.alphas_cumprod = f32(betas), f32(ac)
            self.alphas_cumprod_prev = f32(np.append(1.0, ac[:-1]))
            self.sqrt_alphas_cumprod = f32(np.sqrt(ac))
            self.sqrt_one_minus_alphas_cumprod = f32(np.sqrt(1.0 - ac))
            self.num_timesteps = 1000
            self.device = torch.device("cpu")
            self.parameterization = "eps"

        def apply_model(self, x, t, c, injected_features=None):
            return model(x, t, context=c, injected_features=injected_features)

    class Sampler(DDIMSampler):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    quiet = contextlib.redirect_stdout(io.StringIO())
    sampler = Sampler(Wrapper())
    with quiet:
        sampler.make_schedule(ddim_num_steps=50, ddim_eta=0.0, verbose=False)
    time_range = np.flip(sampler.ddim_timesteps)

    def capture(tag, t_int):
        for m in attns:
            m.gen_pkl = True
        x, ctx = unet_inputs(tag)
        with torch.no_grad():
            model(torch.from_numpy(x), torch.tensor([int(t_int)]), torch.from_numpy(ctx))
        feats = {"config": {"T": 1.5}, "z_enc": torch.from_numpy(x)}
        for bi in range(6, 12):
            a = model.output_blocks[bi][1].transformer_blocks[0].attn1
            for nm in "qkv":
                feats[f"output_block_{bi}_self_attn_{nm}"] = getattr(a, nm).clone()
        for m in attns:
            m.gen_pkl = False
        return feats

    lists = {tag: [capture(tag, t) for t in time_range] for tag in ("cnt", "char", "back")}
    out = {}
    for tag, lst in lists.items():
        for i in (1, 17, 49):
            for k, v in lst[i].items():
                if torch.is_tensor(v) and k != "z_enc":
                    out[f"feat/{tag}/{i}/{k}"] = np.array([v.double().sum().item(), (v.double() ** 2).sum().item()])
    src = open(os.path.join(REF, "run_ori.py")).read()
    ns = {"torch": torch}
    for name in ("feat_merge_2sty", "adain"):
        fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name][0]
        exec(compile(ast.Module(body=[fn], type_ignores=[]), "run_ori.py", "exec"), ns)
    case = {"gamma": 0.75, "T": 1.5, "start_step": 49, "mask": MASK_NAMES[0], "steps": 50, "inter": [16, 32]}
    opt = types.SimpleNamespace(gamma=case["gamma"], T=case["T"])
    feat_maps = ns["feat_merge_2sty"](opt, lists["cnt"], lists["char"], lists["back"], start_step=case["start_step"])
    mask_np = np.load(os.path.join(MASK_DIR, case["mask"] + ".npy"))
    cnt_z, char_z, back_z = (torch.clone(lists[t][0]["z_enc"]) for t in ("cnt", "char", "back"))
    mask = torch.tensor(mask_np, dtype=torch.float32).unsqueeze(0).unsqueeze(0)            # run_ori.py:228-232
    mask = torch.nn.functional.interpolate(mask, size=(cnt_z.shape[2], cnt_z.shape[3]), mode="bilinear", align_corners=False)
    mask = mask.expand(-1, cnt_z.shape[1], -1, -1)
    x_T = mask * ns["adain"](cnt_z, char_z) + (1 - mask) * ns["adain"](cnt_z, back_z)
    out["x_T"] = x_T.numpy()
    _, ctx = unet_inputs("cur")
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "img_mask.npy"), mask_np)
        for m in attns:
            m.cnt_name = os.path.join(td, "img.jpg")
        with quiet, contextlib.redirect_stderr(io.StringIO()):
            samples, inter = sampler.sample(S=50, batch_size=1, shape=[4, 16, 16], verbose=False,
                                            unconditional_conditioning=torch.from_numpy(ctx), eta=0.0, x_T=x_T.clone(),
                                            injected_features=feat_maps, start_step=case["start_step"], log_every_t=1)
        # the same loop without injection (features None), to show that the pin is sensitive to the injection
        for m in attns:
            m.cnt_name = os.path.join(td, "nomask.jpg")
        with quiet, contextlib.redirect_stderr(io.StringIO()):
            plain, _ = sampler.sample(S=50, batch_size=1, shape=[4, 16, 16], verbose=False,
                                      unconditional_conditioning=torch.from_numpy(ctx), eta=0.0, x_T=x_T.clone(),
                                      start_step=case["start_step"])
    xs = inter["x_inter"]                 # [x_T, after evaluation 1, 2, ...]
    assert len(xs) == 1 + 49, len(xs)
    out["final"] = samples.numpy()
    for n in case["inter"]:
        out[f"after/{n}"] = xs[n].numpy()
    out["plain_final"] = plain.numpy()
    np.savez_compressed(os.path.join(HERE, "loop.npz"), **out)
    with open(os.path.join(HERE, "loop_case.json"), "w") as f:
        json.dump(case, f)
    rng = float(out["final"].max() - out["final"].min())
    print("loop: final std %.3f range %.3f; injected vs plain max diff %.3f; finite %s"
          % (out["final"].std(), rng, np.abs(out["final"] - out["plain_final"]).max(), np.isfinite(out["final"]).all()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["masks", "hf", "attention", "feat_merge", "adain", "unet", "hf_patch", "ddim", "loop"]
    if "loop" in which:
        gen_loop()
    if "ddim" in which:
        gen_ddim()
    if "hf_patch" in which:
        gen_hf_patch()
    if "unet" in which:
        gen_unet()
    if "adain" in which:
        gen_adain()
    if "feat_merge" in which:
        gen_feat_merge()
    if "masks" in which:
        gen_masks()
    if "hf" in which:
        gen_hf()
    if "attention" in which:
        gen_attention()
