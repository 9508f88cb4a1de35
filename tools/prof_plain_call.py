"""Plain (single-block) attention, graph-timed whole call and its stages.
    python tools/prof_plain_call.py [N d B reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops
N, d, B, reps = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 4096), (2, 40), (3, 1), (4, 200)))
H = 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
q, k, v = (torch.randn(B, N, H * d, device=dev, generator=g) for _ in range(3))
out = torch.empty_like(q)


def call(flags=0):
    return ops.plain_attention(q, k, v, d ** -0.5, H, out=out, flags=flags)


def graph_time(fn):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        call()
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        for _ in range(reps):
            fn()
    with torch.cuda.stream(st):
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graph.replay(); e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


fl = 4.0 * B * H * N * N * d
t = {name: graph_time(lambda f=f: call(f)) for name, f in (("call", 0), ("pack", 0x10), ("flash", 0x20), ("merge", 0x40))}
print("plain N=%d d=%d B=%d: call %.1f us (%.0f TF/s) | pack %.1f | flash %.1f (%.0f TF/s) | merge %.1f" % (
    N, d, B, t["call"], fl / t["call"] / 1e6, t["pack"], t["flash"], fl / t["flash"] / 1e6, t["merge"]))
