import os, sys, torch
sys.path.insert(0, ".")
from style_transfer_amd import ops
H = 8
dev = torch.device("cuda:0")
def t(Nq, Nk, d, B, qt):
    os.environ["SID_TUNE_SMALL_QT"] = str(qt)
    q = torch.randn(B, Nq, H * d, device=dev); k = torch.randn(B, Nk, H * d, device=dev); v = torch.randn(B, Nk, H * d, device=dev)
    out = torch.empty_like(q)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): ops.plain_attention(q, k, v, d ** -0.5, H, out=out)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for _ in range(200): ops.plain_attention(q, k, v, d ** -0.5, H, out=out)
    with torch.cuda.stream(st):
        gr.replay(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 200
for shape in ((4096, 77, 40, 1), (1024, 77, 80, 1), (256, 77, 160, 1), (256, 256, 160, 1), (256, 256, 160, 4), (4096, 77, 40, 4)):
    print("Nq=%d Nk=%d d=%d B=%d: one tile per workgroup %.1f us, two %.1f us" % (shape + (t(*shape, 1), t(*shape, 2))))
