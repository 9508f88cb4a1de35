"""Run plain attention calls at one shape (for rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops
N, Nk, d, B, reps = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 4096), (2, 4096), (3, 40), (4, 1), (5, 20)))
H = 8
dev = torch.device("cuda:0")
q = torch.randn(B, N, H * d, device=dev); k = torch.randn(B, Nk, H * d, device=dev); v = torch.randn(B, Nk, H * d, device=dev)
for _ in range(reps):
    ops.plain_attention(q, k, v, d ** -0.5, H)
torch.cuda.synchronize()
