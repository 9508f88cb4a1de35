"""Host cost of replaying one image's HIP graph against its GPU time.  python tools/prof_replay.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from style_transfer_amd import ops
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1)
w = bench.make_image_workload(dev, gen, torch.float32)
for thr in (0, ops.FLAG_THROUGHPUT):
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        bench.run_image(w, "bf16", thr)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
        bench.run_image(w, "bf16", thr)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        g.replay()
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
    print("throughput flag %d: replay() returns after %.2f ms on the host; image done after %.2f ms" % (thr != 0, 1e3 * min(host), 1e3 * min(total)))
    # eager launch cost for comparison
    t0 = time.perf_counter()
    with torch.cuda.stream(st):
        bench.run_image(w, "bf16", thr)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("                   eager: launched after %.2f ms on the host; done after %.2f ms" % (1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)))
