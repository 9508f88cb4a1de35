"""Where the host UNet's time goes: a few evaluations of the SD-v1 topology (random weights, no injection),
meant to run under `rocprofv3 --kernel-trace --stats`.
    python tools/prof_unet.py [evals] [fp32|bf16] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import unet as U

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = U.UNetModel().eval().to(dev)
model.set_content_name("/tmp/no_such_image.jpg")      # no mask file: every attention takes the plain path
if os.environ.get("PROF_BENCHMARK"):
    torch.backends.cudnn.benchmark = True
x = torch.randn(B, 4, 64, 64, device=dev)
t = torch.full((B,), 500, device=dev, dtype=torch.long)
ctx = torch.randn(B, 77, 768, device=dev)
import contextlib
ac = torch.autocast("cuda", dtype=torch.bfloat16) if mode == "bf16" else contextlib.nullcontext()
with torch.no_grad(), ac:
    for _ in range(2):
        model(x, t, ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model(x, t, ctx)
    torch.cuda.synchronize()
print("%s B=%d: %.2f ms per evaluation (eager)" % (mode, B, (time.perf_counter() - t0) / n * 1e3))
