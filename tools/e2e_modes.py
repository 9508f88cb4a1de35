import sys, time, types, tempfile, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from style_transfer_amd import feature_store as fs, synth, unet as U
from style_transfer_amd.features import masked_adain_init
from style_transfer_amd.sampler import DDIMSampler
from style_transfer_amd.graphed import GraphedUNet
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
def feature_list(seed):
    g.manual_seed(seed); out = []
    for _ in range(50):
        e = {"config": {"T": 1.5}, "z_enc": torch.randn(1, 4, 64, 64, device=dev, generator=g)}
        for b in range(6, 12):
            n, d = (1024, 80) if b < 9 else (4096, 40)
            for t in "qkv":
                e["output_block_%d_self_attn_%s" % (b, t)] = torch.randn(8, n, d, device=dev, generator=g)
        out.append(e)
    return out
store = fs.FeatureStore(feature_list(1), feature_list(2), feature_list(3)); z = store.z_enc(); store.pack(8, "bf16")
merged = store.merged(types.SimpleNamespace(gamma=0.5, T=1.5), start_step=49)
mask = synth.disc_mask(); td = tempfile.mkdtemp(); np.save(os.path.join(td, "img_mask.npy"), mask)
torch.manual_seed(0)
model = U.UNetModel().eval().to(dev); model.set_content_name(os.path.join(td, "img.jpg"))
ctx = torch.randn(1, 77, 768, device=dev, generator=g); mask_dev = torch.from_numpy(mask).to(dev)
def timed(sampler, images=2):
    def one():
        x_T = masked_adain_init(z[0], z[1], z[2], mask_dev)
        return sampler.sample(x_T, ctx, injected_features=merged, start_step=49)
    one(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(images): out = one()
    torch.cuda.synchronize(); assert torch.isfinite(out.float()).all()
    return (time.perf_counter() - t0) / images
def mk(m):
    s = DDIMSampler(m); s.make_schedule(50, 0.0); return s
print("eager fp32      %.3f s" % timed(mk(model)))
print("graph fp32      %.3f s" % timed(mk(GraphedUNet(model))))
with torch.autocast("cuda", dtype=torch.bfloat16):
    print("graph bf16 autocast %.3f s" % timed(mk(GraphedUNet(model))))
m2 = model.to(memory_format=torch.channels_last)
with torch.autocast("cuda", dtype=torch.bfloat16):
    print("graph bf16 autocast channels_last %.3f s" % timed(mk(GraphedUNet(m2))))
torch.backends.cudnn.benchmark = True
print("graph fp32 benchmark=True %.3f s" % timed(mk(GraphedUNet(model))))
