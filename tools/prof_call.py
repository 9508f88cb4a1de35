"""The injected attention call as a feature store serves it (key/value pairs pre-packed), timed the way the
path runs it: a HIP graph of back-to-back calls.  Prints the whole call (query pack + flash + merge) and
the flash stage alone.
    python tools/prof_call.py [N d B reps compute flags sigma mask]      mask: random | halves | disc | real:<name of a golden content mask>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_amd import ops
N, d, B, reps = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 4096), (2, 40), (3, 1), (4, 200)))
compute = sys.argv[5] if len(sys.argv) > 5 else "bf16"
base_flags = int(sys.argv[6]) if len(sys.argv) > 6 else 0
H = 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
sigma = float(sys.argv[7]) if len(sys.argv) > 7 else 1.0
rn = lambda *s: torch.randn(*s, device=dev, generator=g) * sigma
xq = rn(B, N, H * d)
q_cnt = rn(B * H, N, d)
kv = [ops.pack_kv(rn(B * H, N, d), rn(B * H, N, d), H, compute) for _ in range(3)]
m = torch.sign(rn(B, N)).contiguous()
if len(sys.argv) > 8 and sys.argv[8] == "halves":
    m = torch.ones(B, N, device=dev); m[:, N // 2:] = -1
if len(sys.argv) > 8 and sys.argv[8] == "disc":
    from style_transfer_amd import synth
    side = int(round(N ** 0.5))
    m = (torch.from_numpy(synth.disc_mask(side, 0.3)).reshape(1, N) * 2 - 1).to(dev).repeat(B, 1).contiguous()
if len(sys.argv) > 8 and sys.argv[8].startswith("real:"):      # one of the reference's content masks (tests/golden)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from helpers import load_masks
    side = int(round(N ** 0.5))
    m = ops.mask_downsample(torch.from_numpy(load_masks()[sys.argv[8][5:]]).to(dev), side, side, 1.0).reshape(1, N).repeat(B, 1).contiguous()
out = torch.empty_like(xq)


def call(flags=0):
    return ops.injected_attention_packed(xq, q_cnt, kv[0], kv[1], kv[2], m, 0.5, 1.5, d ** -0.5, H, out=out, flags=flags | base_flags)


def graph_time(fn, first=None):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        (first or call)()           # a full call of the SAME entry point first: this stream's workspace must hold its
                                    # query images and, with block skipping, its token map
        for _ in range(3):          # (a flash stage fed zeros draws less power and runs 17 % faster)
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


fl = 12.0 * B * H * N * N * d
t_call = graph_time(call)
t_flash = graph_time(lambda: call(0x20))
if base_flags & 1:
    call(); print("skip counters", ops.skip_counts(xq, H, compute, packed=True), end=" ")
print("N=%d d=%d B=%d %s: call %.1f us (%.0f TF/s = %.1f%% of 2.5 PF) | flash stage alone %.1f us (%.1f%%)" % (
    N, d, B, compute, t_call, fl / t_call / 1e6, fl / t_call / 2.5e9 * 100, t_flash, fl / t_flash / 2.5e9 * 100))
if os.environ.get("SID_TUNE_W") or os.environ.get("SID_TUNE_KS") or os.environ.get("PROF_NO_CACHED"):
    sys.exit(0)                             # (a content partial is one run per head: not under a forced decomposition)
# the same call with the content block taken from a partial computed once (sid_injected_attn_fwd_cached)
partial = ops.content_partial(q_cnt, kv[2], d ** -0.5, H)
cached = lambda flags=0: ops.injected_attention_cached(xq, q_cnt, kv[0], kv[1], partial, m, 0.5, 1.5, d ** -0.5, H, out=out,
                                                       flags=flags | base_flags)
if os.environ.get("PROF_CACHED_W"):         # force the runs per head of the cached call only (measurement)
    os.environ["SID_TUNE_W"] = os.environ["PROF_CACHED_W"]
cached()
if base_flags & 1:
    print("cached call, skip counters", ops.skip_counts(xq, H, compute, flags=base_flags, packed="cached"), end=" ")
t_cached = graph_time(cached, cached)
t_cflash = graph_time(lambda: cached(0x20), cached)
os.environ.pop("SID_TUNE_W", None)
t_partial = graph_time(lambda: ops.content_partial(q_cnt, kv[2], d ** -0.5, H))
print("cached content block: call %.1f us | flash stage %.1f us | making the partial %.1f us" % (t_cached, t_cflash, t_partial))
