"""Experiment: images in flight on S streams per GPU, each image's hot path captured once as a HIP graph.
    python tools/bench_streams.py S images"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
IMAGES = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1234)
w = bench.make_image_workload(dev, gen, torch.float32)
bench.run_image(w, "bf16")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(IMAGES):
    bench.run_image(w, "bf16")
torch.cuda.synchronize()
print("eager, 1 stream: %.2f imgs/s" % (IMAGES / (time.perf_counter() - t0)))
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
graphs, outs = [], []
for s in streams:
    with torch.cuda.stream(s):
        bench.run_image(w, "bf16")          # warm-up on this stream: its workspace is allocated here
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        outs.append(bench.run_image(w, "bf16"))
    graphs.append(g)
torch.cuda.synchronize()
for n_active in range(1, S + 1):
    for g in graphs[:n_active]:
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per = IMAGES
    for _ in range(per):
        for s, g in zip(streams[:n_active], graphs[:n_active]):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("graphs, %d stream(s): %.2f imgs/s" % (n_active, per * n_active / dt))
ref = bench.run_image(w, "bf16")
torch.cuda.synchronize()
print("graph output equals eager output:", all(torch.equal(o, ref) for o in outs))
