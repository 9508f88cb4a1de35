"""Image-level effect of SID_FLAG_BLOCK_SKIP: the bench's hot path of one image at feature-sized logits,
for 1 and 3 images in flight, with the latency and the throughput decompositions.
    python tools/skip_image.py [sigma]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from style_transfer_amd import ops
def say(*a):
    print(*a, flush=True)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
lanes = []
for li in range(3):
    gen.manual_seed(4300 + li)
    say("lane", li, "workload")
    w = bench.make_image_workload(dev, gen, torch.float32, "packed", "bf16", sigma=sigma)
    torch.cuda.synchronize()
    st = torch.cuda.Stream(device=dev)
    graphs = {}
    for fl in (2, 3, 0, 1):
        say("lane", li, "eager, flags", fl)
        with torch.cuda.stream(st):
            bench.run_image(w, "bf16", fl)
        torch.cuda.synchronize()
        say("lane", li, "capture, flags", fl)
        graphs[fl] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graphs[fl], stream=st, capture_error_mode="thread_local"):
            bench.run_image(w, "bf16", fl)
        torch.cuda.synchronize()
    lanes.append((st, graphs, w))
for n in (1, 3):
    for fl in (2, 3, 0, 1):
        say("replays:", n, "in flight, flags", fl)
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(12):
                st, graphs, _ = lanes[i % n]
                with torch.cuda.stream(st):
                    graphs[fl].replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 12
        say("sigma %.0f, %d in flight, flags %d (%s%s): %.2f ms per image" % (
            sigma, n, fl, "throughput" if fl & 2 else "latency", " + skip" if fl & 1 else "", dt * 1e3))
