"""Time the fused high-frequency filter at the decoder ResBlock shapes (HIP events, back-to-back launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops
dev = torch.device("cuda:0")
reps = 300
for (C, S) in ((320, 64), (640, 32), (1280, 16), (320, 128)):
    x, s, h = (torch.randn(1, C, S, S, device=dev) for _ in range(3))
    out = torch.empty_like(x)
    for fused in (True, False):
        fn = (lambda: ops.hf_filter(x, 0.5, skip=s, h=h, out=out)) if fused else (lambda: ops.hf_filter(x, 0.5, out=out))
        # a HIP graph of `reps` launches: the host's launch path (about 10 us per call from Python) is not timed
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(10):
                fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            for _ in range(reps):
                fn()
        graph.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        nbytes = (16 if fused else 8) * C * S * S
        print("(%d,%d,%d) %s: %.2f us, %.0f GB/s (%.1f%% of 8 TB/s)" % (C, S, S, "skip+h+HF" if fused else "HF only", t * 1e6,
                                                                         nbytes / t / 1e9, nbytes / t / 8e12 * 100))
