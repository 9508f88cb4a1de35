"""Run the injected attention a few times at one shape (for rocprofv3 kernel-trace / PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops
N, d, B, reps = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 4096), (2, 40), (3, 1), (4, 10)))
compute = sys.argv[5] if len(sys.argv) > 5 else "f16"      # the package default
H = 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
xq = rn(B, N, H * d)
f = [rn(B * H, N, d) for _ in range(7)]
side = int(N ** 0.5)
m = torch.sign(rn(B, N)).contiguous()
for _ in range(reps):
    ops.injected_attention(xq, *f, m, 0.5, 1.5, d ** -0.5, H, compute=compute)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(reps):
    ops.injected_attention(xq, *f, m, 0.5, 1.5, d ** -0.5, H, compute=compute)
ev1.record(); ev1.synchronize()
t = ev0.elapsed_time(ev1) / reps
def timed(flags):
    for _ in range(3):
        ops.injected_attention(xq, *f, m, 0.5, 1.5, d ** -0.5, H, compute=compute, flags=flags)
    ev0.record()
    for _ in range(reps):
        ops.injected_attention(xq, *f, m, 0.5, 1.5, d ** -0.5, H, compute=compute, flags=flags)
    ev1.record(); ev1.synchronize()
    return ev0.elapsed_time(ev1) / reps
tp, tf, tm = timed(0x10), timed(0x20), timed(0x40)
kv = [ops.pack_kv(f[1 + 2 * i], f[2 + 2 * i], H, compute) for i in range(3)]
def packed(flags=0):
    return ops.injected_attention_packed(xq, f[0], kv[0], kv[1], kv[2], m, 0.5, 1.5, d ** -0.5, H, flags=flags)
for _ in range(reps):
    packed()
ev0.record()
for _ in range(reps):
    packed()
ev1.record(); ev1.synchronize()
tpk = ev0.elapsed_time(ev1) / reps
for _ in range(3):
    packed(0x10)
ev0.record()
for _ in range(reps):
    packed(0x10)
ev1.record(); ev1.synchronize()
tqp = ev0.elapsed_time(ev1) / reps
fl = 12.0 * B * H * N * N * d
print("N=%d d=%d B=%d: call %.1f us (%.1f TF/s) | pack %.1f us | merge %.1f us | flash %.1f us (%.1f TF/s = %.1f%% of 2.5 PF)" % (
    N, d, B, t * 1e3, fl / (t * 1e-3) / 1e12, tp * 1e3, tm * 1e3, tf * 1e3, fl / (tf * 1e-3) / 1e12, fl / (tf * 1e-3) / 2.5e15 * 100))
print("   pre-packed K/V: call %.1f us (%.1f TF/s = %.1f%% of 2.5 PF) | query pack %.1f us" % (
    tpk * 1e3, fl / (tpk * 1e-3) / 1e12, fl / (tpk * 1e-3) / 2.5e15 * 100, tqp * 1e3))
