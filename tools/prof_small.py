"""Run plain attention calls at the UNet's small shapes (for rocprofv3 --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops
H = 8
dev = torch.device("cuda:0")
for (N, Nk, d) in [(256, 256, 160), (64, 64, 160), (4096, 77, 40), (1024, 77, 80), (256, 77, 160), (1024, 1024, 80), (4096, 4096, 40)]:
    q = torch.randn(1, N, H * d, device=dev); k = torch.randn(1, Nk, H * d, device=dev); v = torch.randn(1, Nk, H * d, device=dev)
    for _ in range(20):
        ops.plain_attention(q, k, v, d ** -0.5, H)
    torch.cuda.synchronize()
