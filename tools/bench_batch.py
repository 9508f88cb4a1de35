"""Experiment: B images per call x S streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from style_transfer_amd import ops
dev = torch.device("cuda:0")
for B, S in [(1, 3), (2, 1), (2, 2), (2, 3), (3, 1), (3, 2), (4, 1), (4, 2)]:
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    w = bench.make_image_workload(dev, gen, torch.float32, B)
    fl = ops.FLAG_THROUGHPUT
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    graphs = []
    for s in streams:
        with torch.cuda.stream(s):
            bench.run_image(w, "bf16", fl)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            bench.run_image(w, "bf16", fl)
        graphs.append(g)
    torch.cuda.synchronize()
    for g in graphs:
        g.replay()
    torch.cuda.synchronize()
    reps = 4
    t0 = time.perf_counter()
    for _ in range(reps):
        for s, g in zip(streams, graphs):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("batch %d x %d streams: %.2f imgs/s" % (B, S, reps * S * B / dt), flush=True)
    del graphs, w
    torch.cuda.empty_cache()
