"""Kernel microbenchmark: injected attention at the UNet levels, batch B, HIP events around whole calls.
    python tools/microbench.py [B]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = 8
dev = torch.device("cuda:0")
rows = []
for (N, d, b) in [(4096, 40, B), (1024, 80, B), (256, 160, B), (64, 160, B), (16384, 40, 1), (16384, 40, 4), (4096, 80, 4)]:
    g = torch.Generator(device=dev); g.manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    xq = rn(b, N, H * d); f = [rn(b * H, N, d) for _ in range(7)]; m = torch.sign(rn(b, N)).contiguous()
    for compute in ("bf16", "fp32"):
        call = lambda flags=0: ops.injected_attention(xq, *f, m, 0.5, 1.5, d ** -0.5, H, compute=compute, flags=flags)
        res = {}
        for name, flags in (("call", 0), ("flash", 0x20)):
            for _ in range(3):
                call(flags)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20 if N * b <= 65536 else 5
            e0.record()
            for _ in range(reps):
                call(flags)
            e1.record(); e1.synchronize()
            res[name] = e0.elapsed_time(e1) / reps * 1e-3
        fl = 12.0 * b * H * N * N * d
        rows.append({"N": N, "d": d, "B": b, "compute": compute, "call_us": res["call"] * 1e6, "flash_us": res["flash"] * 1e6,
                     "call_TFs": fl / res["call"] / 1e12, "flash_TFs": fl / res["flash"] / 1e12})
        print(json.dumps(rows[-1]))
