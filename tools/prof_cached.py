"""Run the cached-content injected attention a few times at one shape (for rocprofv3 --kernel-trace --stats):
the partial once, then `reps` calls of sid_injected_attn_fwd_cached.
    python tools/prof_cached.py [N d B reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops
N, d, B, reps = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 4096), (2, 40), (3, 1), (4, 50)))
H = 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
xq, q_cnt = rn(B, N, H * d), rn(B * H, N, d)
kv = [ops.pack_kv(rn(B * H, N, d), rn(B * H, N, d), H, "bf16") for _ in range(3)]
m = torch.sign(rn(B, N)).contiguous()
partial = ops.content_partial(q_cnt, kv[2], d ** -0.5, H)
out = torch.empty_like(xq)
for _ in range(reps):
    ops.injected_attention_cached(xq, q_cnt, kv[0], kv[1], partial, m, 0.5, 1.5, d ** -0.5, H, out=out)
torch.cuda.synchronize()
print("done")
