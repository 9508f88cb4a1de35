import os, sys
sys.path.insert(0, "/root/repo")
import torch
from style_transfer_amd import ops
N, d, B, reps, H = 4096, 40, 1, 200, 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
xq = rn(B, N, H * d); q_cnt = rn(B * H, N, d)
kv = [ops.pack_kv(rn(B * H, N, d), rn(B * H, N, d), H, "bf16") for _ in range(3)]
m = torch.sign(rn(B, N)).contiguous(); out = torch.empty_like(xq)
big = torch.empty(64 << 20, device=dev, dtype=torch.float32); big2 = torch.empty_like(big)
def call(flags=0):
    return ops.injected_attention_packed(xq, q_cnt, kv[0], kv[1], kv[2], m, 0.5, 1.5, d ** -0.5, H, out=out, flags=flags)
def graph_time(fn):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        for _ in range(reps): fn()
    with torch.cuda.stream(st):
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graph.replay(); e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
call()
t = {}
t["flash"] = graph_time(lambda: call(0x20))
t["pack"] = graph_time(lambda: call(0x10))
t["merge"] = graph_time(lambda: call(0x40))
t["pack+flash"] = graph_time(lambda: call(0x30))
t["flash+merge"] = graph_time(lambda: call(0x60))
t["all"] = graph_time(lambda: call(0))
t["copy256MB"] = graph_time(lambda: big2.copy_(big))
def f2(): big2.copy_(big); call(0x20)
t["copy+flash"] = graph_time(f2)
small = torch.empty(1 << 20, device=dev); small2 = torch.empty_like(small)
t["copy4MB"] = graph_time(lambda: small2.copy_(small))
def f3(): small2.copy_(small); call(0x20)
t["copy4MB+flash"] = graph_time(f3)
for k, v in t.items(): print("%-16s %.1f us" % (k, v))

# which dependency costs time: the flash stage reading the query images the pack stage has just written?
from style_transfer_amd import lib as _lib
L = _lib.load()
nbytes = L.sid_injected_attn_packed_workspace_bytes(B, H, N, d, 0)
wsA = torch.empty(nbytes, dtype=torch.uint8, device=dev); wsB = torch.empty(nbytes, dtype=torch.uint8, device=dev)
def raw(ws, flags):
    _lib.check(L.sid_injected_attn_fwd_packed(xq.data_ptr(), q_cnt.data_ptr(), kv[0].image.data_ptr(), kv[1].image.data_ptr(),
               kv[2].image.data_ptr(), m.data_ptr(), out.data_ptr(), B, H, N, d, 0.5, 0.5, 1.5, d ** -0.5, 0, 0, flags,
               ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
raw(wsA, 0); raw(wsB, 0)
def s_same(): raw(wsA, 0x10); raw(wsA, 0x20)
def s_other(): raw(wsA, 0x10); raw(wsB, 0x20)
def s_gap(): raw(wsA, 0x10); small2.copy_(small); raw(wsA, 0x20)
print("pack(A)+flash(A)        %.1f us" % graph_time(s_same))
print("pack(A)+flash(B)        %.1f us" % graph_time(s_other))
print("pack(A)+copy4MB+flash(A) %.1f us" % graph_time(s_gap))

# a real streaming kernel (not a copy engine) in front of the flash stage: does evicting L2 cost the same?
mid = torch.empty(8 << 20, device=dev, dtype=torch.float32)      # 32 MB: every XCD's 4 MB L2 several times over
tiny = torch.empty(64 << 10, device=dev, dtype=torch.float32)    # 256 KB
t_mid = graph_time(lambda: mid.add_(1.0)); t_tiny = graph_time(lambda: tiny.add_(1.0))
def s_mid(): mid.add_(1.0); raw(wsA, 0x20)
def s_tiny(): tiny.add_(1.0); raw(wsA, 0x20)
print("add 32MB %.1f us | add 32MB + flash %.1f us | add 256KB %.1f | add 256KB + flash %.1f us" % (t_mid, graph_time(s_mid), t_tiny, graph_time(s_tiny)))

# alternate two different flash kernel objects of the same shape (bf16 and half-precision instances)
kv16 = [ops.pack_kv(rn(B * H, N, d), rn(B * H, N, d), H, "f16") for _ in range(3)]
def raw16(ws, flags):
    _lib.check(L.sid_injected_attn_fwd_packed(xq.data_ptr(), q_cnt.data_ptr(), kv16[0].image.data_ptr(), kv16[1].image.data_ptr(),
               kv16[2].image.data_ptr(), m.data_ptr(), out.data_ptr(), B, H, N, d, 0.5, 0.5, 1.5, d ** -0.5, 0, 2, flags,
               ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
raw16(wsB, 0)
def s_alt(): raw(wsA, 0x20); raw16(wsB, 0x20)
print("flash bf16 + flash f16 (two kernel objects): %.1f us per pair" % graph_time(s_alt))
one = torch.empty(64, device=dev)
def s_one(): one.add_(1.0); raw(wsA, 0x20)
print("add 64 elements + flash: %.1f us" % graph_time(s_one))
def s_two(): raw(wsA, 0x20); raw(wsA, 0x20); one.add_(1.0)
print("flash, flash, add 64 elements: %.1f us per triple" % graph_time(s_two))

# marginal cost of the k-th flash launch after a different kernel
reps = 60
prev = None
for K in (1, 2, 3, 4, 6, 8, 12, 16):
    def s_k(K=K):
        one.add_(1.0)
        for _ in range(K): raw(wsA, 0x20)
    tK = graph_time(s_k)
    print("tiny + %2d x flash: %.1f us%s" % (K, tK, "" if prev is None else "   (+%.1f us per extra launch)" % ((tK - prev[1]) / (K - prev[0]))))
    prev = (K, tK)
