"""Aggregate rate of the flash stage (N=4096, d=40, B=1) when S streams issue it concurrently.
    python tools/prof_conc.py [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, d, B, H = 4096, 40, 1, 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
xq = rn(B, N, H * d); f = [rn(B * H, N, d) for _ in range(7)]; m = torch.sign(rn(B, N)).contiguous()
fl = 12.0 * B * H * N * N * d
for thr in (0, ops.FLAG_THROUGHPUT):
    for S in (1, 2, 3, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        for s in streams:
            with torch.cuda.stream(s):
                ops.injected_attention(xq, *f, m, 0.5, 1.5, d ** -0.5, H, flags=thr)       # fills this stream's workspace
        torch.cuda.synchronize()
        graphs = []
        for s in streams:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s, capture_error_mode="thread_local"):
                for _ in range(reps):
                    ops.injected_attention(xq, *f, m, 0.5, 1.5, d ** -0.5, H, flags=thr | 0x20)
            graphs.append(gr)
        for s, gr in zip(streams, graphs):
            with torch.cuda.stream(s):
                gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s, gr in zip(streams, graphs):
            with torch.cuda.stream(s):
                gr.replay()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / (S * reps)
        # the same graphs replayed three times each, round-robin over the streams (what bench.py does)
        t0 = time.perf_counter()
        R = int(os.environ.get('REPLAYS', '3'))
        for _ in range(R):
            for s, gr in zip(streams, graphs):
                with torch.cuda.stream(s):
                    gr.replay()
        torch.cuda.synchronize()
        t3 = (time.perf_counter() - t0) / (R * S * reps)
        print("throughput flag %d, %d stream(s): %.1f us per call aggregate, %.0f TFLOP/s; several replays each: %.1f us per call" % (
            thr != 0, S, t * 1e6, fl / t / 1e12, t3 * 1e6), flush=True)
