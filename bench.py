#!/usr/bin/env python3
"""Hot-path benchmark: stylised images per second on the injection path of one 512x512 image.

A step is the hot path of ONE stylised image (50 DDIM steps = 49 UNet evaluations): per
evaluation 6 injected three-block attentions (3 x N=4096,d=40 and 3 x N=1024,d=80), the 10
non-injected self-attentions of the UNet (7 plain, 3 temperature-scaled), and, on the 19
evaluations the reference's schedule reaches, the fused high-frequency filter of the 6 decoder
ResBlocks (3 x (640,32,32), 3 x (320,64,64)).  The convolutions, GroupNorms, cross-attention, VAE
and sampler arithmetic of the UNet are NOT part of the path and are not run in `value` (the `e2e`
record times the whole UNet loop separately).  Inputs are synthetic (seeded normal), already
resident in HBM; every evaluation reads its own feature, query and latent tensors, as the
precomputed feature lists and the evolving latent do.  The feature store keeps the key/value pairs
in the kernels' operand layout (FeatureStore.pack / sid_pack_kv, done when a feature file is
loaded, outside the timed region); `--store tensors` keeps them as fp32 tensors instead.

    python bench.py --gpus N --steps K --warmup W

prints one JSON line on rank 0.  With N > 1 and no WORLD_SIZE in the environment the command starts
its own N ranks (one process per GPU) before anything touches a GPU; under torch.distributed.run it
uses the ranks it is given.  Every rank stylises its own images (the path shards over images, no
data-path collective) and rank 0 gathers the output latents once at the end over RCCL.

Images are independent, so each GPU keeps `--in-flight` (default 4) of them going at once, one HIP
stream and one workload each, and the ~2300 kernel launches of an image are captured once as a HIP
graph and replayed.  `--in-flight 1 --no-graph` is the plain one-image-at-a-time launch sequence.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HEADS = 8
GAMMA, TEMP, RATIO = 0.5, 1.5, 0.5
EVALS = 49
# The reference's callback hands the ResBlocks ddim_timesteps[i] of loop position i after step i
# (hf_run.py:256-270), so position i + 1 filters when that timestep is among the first 40 % of the
# ascending list: positions 2..20 of 0..49, i.e. evaluations 1..19 of the 49 (position 0 is skipped).
HF_FIRST, HF_EVALS = 1, 19
INJECTED = [(1024, 80)] * 3 + [(4096, 40)] * 3            # output blocks 6-8, 9-11
PLAIN = [(4096, 40, 1.0)] * 2 + [(1024, 80, 1.0)] * 2 + [(256, 160, 1.0)] * 2 + [(64, 160, 1.0)] + \
        [(256, 160, TEMP)] * 3                            # input 1,2,4,5,7,8; middle; output 3,4,5 (T)
HF_MAPS = [(640, 32)] * 3 + [(320, 64)] * 3
MFMA_PEAK = 2.5e15                # dense bf16 MFMA, FLOP/s (MI355X)
HBM_PEAK = 8.0e12


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--compute", default="f16", choices=["bf16", "f16", "fp32"],
                    help="f16 (the package default) = half-precision query/key operands, bf16 probabilities and values; "
                         "bf16 = bf16 operands throughout; fp32 = split-bf16")
    ap.add_argument("--io", default="f32", choices=["f32", "bf16"], help="storage type of the query / latent tensors")
    ap.add_argument("--store", default="packed", choices=["packed", "tensors"],
                    help="feature store layout of the key/value pairs: the kernels' operand layout, or tensors")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end record (host UNet around the path)")
    ap.add_argument("--no-block-skip", action="store_true", help="skip the SID_FLAG_BLOCK_SKIP on/off record")
    ap.add_argument("--no-content-cache", action="store_true", help="skip the cached-content-block record")
    ap.add_argument("--no-feature-scale", action="store_true", help="skip the record at feature-sized logits (both arithmetic modes)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the feature-store sweep record (load + sample over 8 triples)")
    ap.add_argument("--no-default-queues", action="store_true", help="skip the child run with the runtime's default hardware queues")
    ap.add_argument("--headline-only", action="store_true", help="print the line with `value` and `config` only (no side records)")
    ap.add_argument("--in-flight", type=int, default=4, help="images processed concurrently per GPU (one stream each)")
    ap.add_argument("--hw-queues", type=int, default=8, help="GPU_MAX_HW_QUEUES for this process and its ranks (0: leave the environment alone)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernel by kernel instead of replaying a HIP graph")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI; gloo only to rehearse on one GPU")
    ap.add_argument("--rank-probe", action="store_true", help=argparse.SUPPRESS)   # launcher test: print the rank environment and exit
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start `n` fresh processes of this script, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    the way torch.distributed.run sets them), relay rank 0's output and return the worst exit code.  Called
    before this process has made any GPU call: nothing here touches HIP."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    codes = [p.wait() for p in procs]
    return max((abs(c) for c in codes), default=0)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    # One hardware queue per image lane besides the default stream's: the runtime's default of four makes the
    # fourth lane share a queue (43 instead of 54 imgs/s).  Read by the HIP runtime when it starts, which in
    # this process (and in the ranks it starts) has not happened yet.
    if args.hw_queues > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.rank_probe:
        print(json.dumps({"rank": rank, "local_rank": local, "world": world, "master": os.environ.get("MASTER_ADDR"),
                          "port": os.environ.get("MASTER_PORT")}), flush=True)
        return
    run_rank(args, world, rank, local)


# ----------------------------------------------------------------------------------------------
# everything below runs inside one rank
# ----------------------------------------------------------------------------------------------
def make_image_workload(dev, gen, dtype, store, compute, sigma=1.0):
    """All tensors the hot path of one image reads, resident on `dev`: per evaluation its own query
    projection, feature tensors (or packed key/value pairs), plain-attention inputs and content maps.
    `sigma` scales the attention tensors (1: the headline's unit-variance features)."""
    import torch
    from style_transfer_amd import ops

    def rn(*shape):
        return (torch.randn(*shape, device=dev, dtype=torch.float32, generator=gen) * sigma).to(dtype)

    w = {"inj": [], "plain": [], "hf": [], "store": store}
    w["z"] = [rn(1, 4, 64, 64).float() for _ in range(3)]          # inverted latents: content / char / back
    yy, xx = torch.meshgrid(torch.arange(512, device=dev), torch.arange(512, device=dev), indexing="ij")
    disc = (((yy - 255.5) ** 2 + (xx - 255.5) ** 2) <= (0.3 * 512) ** 2).float().contiguous()
    w["mask512"] = disc
    for (N, d) in INJECTED:
        side = int(N ** 0.5)
        layer = {"N": N, "d": d, "xq": [rn(1, N, HEADS * d) for _ in range(EVALS)], "feats": [],
                 "mask": ops.mask_downsample(disc, side, side, 1.0).reshape(1, -1).contiguous()}
        for _ in range(EVALS):
            q_cnt = rn(HEADS, N, d)
            pairs = [(rn(HEADS, N, d), rn(HEADS, N, d)) for _ in range(3)]       # char, back, content
            if store == "packed":
                layer["feats"].append((q_cnt, [ops.pack_kv(k, v, HEADS, compute) for k, v in pairs]))
            else:
                layer["feats"].append((q_cnt, pairs))
        w["inj"].append(layer)
    for (N, d, T) in PLAIN:
        w["plain"].append({"N": N, "d": d, "T": T,
                           "qkv": [tuple(rn(1, N, HEADS * d) for _ in range(3)) for _ in range(EVALS)]})
    for (C, S) in HF_MAPS:
        w["hf"].append({"h_cnt": [rn(1, C, S, S) for _ in range(HF_EVALS)], "skip": rn(1, C, S, S), "h": rn(1, C, S, S)})
    torch.cuda.synchronize()
    return w


def injected_call(layer, e, compute, flags=0, cached=False):
    """`cached`: take the content block from the partial attached to the content pair (content_cache_record)."""
    from style_transfer_amd import ops
    q_cnt, kv = layer["feats"][e]
    scale = layer["d"] ** -0.5
    if cached:
        return ops.injected_attention_cached(layer["xq"][e], q_cnt, kv[0], kv[1], kv[2].partial, layer["mask"], GAMMA, TEMP,
                                             scale, HEADS, flags=flags)
    if isinstance(kv[0], ops.PackedKV):
        return ops.injected_attention_packed(layer["xq"][e], q_cnt, kv[0], kv[1], kv[2], layer["mask"], GAMMA, TEMP, scale,
                                             HEADS, flags=flags)
    return ops.injected_attention(layer["xq"][e], q_cnt, kv[0][0], kv[0][1], kv[1][0], kv[1][1], kv[2][0], kv[2][1],
                                  layer["mask"], GAMMA, TEMP, scale, HEADS, compute=compute, flags=flags)


def run_image(w, compute, flags=0, cached=False):
    """The hot path of one image: the masked AdaIN initial latent, then 49 evaluations in UNet order."""
    from style_transfer_amd import ops
    from style_transfer_amd.attention import resolve_compute
    outs = None
    ops.masked_adain(w["z"][0], w["z"][1], w["z"][2], w["mask512"])
    for e in range(EVALS):
        for p in w["plain"]:
            q, k, v = p["qkv"][e]
            ls = p["T"] * p["d"] ** -0.5
            ops.plain_attention(q, k, v, ls, HEADS, compute=resolve_compute(compute, p["N"], p["d"], ls), flags=flags)
        hf_on = HF_FIRST <= e < HF_FIRST + HF_EVALS
        for li, layer in enumerate(w["inj"]):
            if hf_on:
                m = w["hf"][li]
                ops.hf_filter(m["h_cnt"][e - HF_FIRST], RATIO, skip=m["skip"], h=m["h"])
            outs = injected_call(layer, e, compute, flags, cached)
    return outs


def events_per_call(fn, reps):
    """Average duration of `fn()`: `reps` back-to-back calls captured as one HIP graph on a stream of their
    own, HIP events on that stream around a replay (the host's launch path, about 10 us per call from
    Python, stays out of the figure; kernels keep their order and their launch gaps).  The graph runs once
    untimed first: after an idle gap the chip needs tens of milliseconds to return to its working clock."""
    import torch
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
        for _ in range(reps):
            fn()
    with torch.cuda.stream(st):
        graph.replay()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        graph.replay()
        ev1.record()
    ev1.synchronize()
    return ev0.elapsed_time(ev1) * 1e-3 / reps


def eager_events_per_call(fn, reps):
    """Average duration of `fn()` from HIP events on the current stream around `reps` eager back-to-back
    calls, after as many untimed ones.  For kernels much longer than the host's launch path (about 15 us
    per call from Python) the queue never drains, so this is the kernel's own launch-to-launch time; it
    agrees with rocprofv3's per-dispatch durations in the mixed kernel sequence of a real call.  The
    calls run on the stream whose workspace the preceding full call filled: a flash stage that reads an
    unfilled (zero) workspace draws less power and runs 17 % faster, which is not a measurement."""
    import torch
    for _ in range(reps):
        fn()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    ev1.synchronize()
    return ev0.elapsed_time(ev1) * 1e-3 / reps


def kernel_records(w, compute):
    """The dominant kernel and its call, alone on the GPU: the flash stage of the injected attention at
    N=4096, d=40, B=1 (HIP events around back-to-back launches of that stage), the whole call as the feature
    store serves it (query pack + flash + merge), and the fused high-frequency filter at (320,64,64)."""
    from style_transfer_amd import ops
    layer = w["inj"][-1]
    injected_call(layer, 0, compute)                                   # a full call fills the workspace
    t_flash = eager_events_per_call(lambda: injected_call(layer, 0, compute, flags=0x20), 300)      # flash stage only
    t_call = events_per_call(lambda: injected_call(layer, 0, compute), 300)
    flops = 12.0 * 1 * HEADS * 4096 ** 2 * 40           # 3 QK^T + 3 PV blocks, d = 40 (no padding counted)
    m = w["hf"][-1]
    t_hf = events_per_call(lambda: ops.hf_filter(m["h_cnt"][0], RATIO, skip=m["skip"], h=m["h"]), 300)
    hf_bytes = 16.0 * 320 * 64 * 64                     # three reads + one write of a (1,320,64,64) f32 map
    # cold operands: 16 distinct (h_cnt, skip, h, out) sets, 336 MB in all, visited in turn, so that no launch finds
    # its operands in the 256 MB Infinity Cache (the figure above replays one 21 MB set)
    import torch
    cold = [tuple(torch.randn(1, 320, 64, 64, device=m["skip"].device) for _ in range(4)) for _ in range(16)]
    rot = [0]

    def hf_cold():
        a, b, c, o = cold[rot[0] % 16]
        rot[0] += 1
        ops.hf_filter(a, RATIO, skip=b, h=c, out=o)
    t_hf_cold = events_per_call(hf_cold, 320)
    del cold
    hf_kept = ops.hf_filter(m["h_cnt"][0], RATIO)       # HF(h_cnt) made once per content image: the sum alone is left
    t_sum = events_per_call(lambda: ops.residual_sum(m["skip"], m["h"], hf_kept), 300)
    roof = {"bound": "mfma", "kernel": "sid_attn::flash_kernel<3,2,2,false,2,8,true,%s,false> injected attention N=4096 d=40 H=8 B=1 (%s operands for q.k^T)"
                      % (("true", "half-precision") if compute == "f16" else ("false", "bf16")),
            "achieved": flops / t_flash / 1e12, "peak": MFMA_PEAK / 1e12, "unit": "TFLOP/s",
            "frac": flops / t_flash / MFMA_PEAK, "kernel_ms": t_flash * 1e3, "algorithmic_flop": flops,
            "call_ms": t_call * 1e3, "call_frac": flops / t_call / MFMA_PEAK,
            "call": "whole injected-attention call at B=1, key/value pairs %s" %
                    ("pre-packed by the feature store: query pack + flash + merge" if w["store"] == "packed"
                     else "packed per call: pack + flash + merge")}
    roof.update(measured_traffic())
    if roof.get("clock"):
        roof["clock"]["frac_at_zero_operand_clock"] = roof["frac"] * roof["clock"]["zero_operands_GHz"] / roof["clock"]["random_operands_GHz"]
    if compute == "f16" and w["store"] == "packed":
        # the same stage with bf16 operands for q.k^T (the bf16-throughout arithmetic, selectable with
        # --compute bf16): the half-precision multipliers cost a few per cent of clock
        import torch
        gen = torch.Generator(device=layer["mask"].device); gen.manual_seed(99)
        rn = lambda *sh: torch.randn(*sh, device=layer["mask"].device, generator=gen)
        alt = {"N": 4096, "d": 40, "xq": [rn(1, 4096, HEADS * 40)], "mask": layer["mask"],
               "feats": [(rn(HEADS, 4096, 40), [ops.pack_kv(rn(HEADS, 4096, 40), rn(HEADS, 4096, 40), HEADS, "bf16") for _ in range(3)])]}
        injected_call(alt, 0, "bf16")
        t_alt = eager_events_per_call(lambda: injected_call(alt, 0, "bf16", flags=0x20), 300)
        roof["bf16_operands"] = {"kernel": "sid_attn::flash_kernel<3,2,2,false,2,8,true,false,false>", "kernel_ms": t_alt * 1e3,
                                 "achieved": flops / t_alt / 1e12, "frac": flops / t_alt / MFMA_PEAK}
    hf = {"bound": "hbm", "kernel": "sid_hf::hf2_kernel fused skip + h + HF(h_cnt), (1,320,64,64) f32", "kernel_ms": t_hf * 1e3,
          "achieved": hf_bytes / t_hf / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": hf_bytes / t_hf / HBM_PEAK,
          "algorithmic_bytes": hf_bytes,
          "cache_assisted": True,
          "cold": {"what": "the same launch over 16 operand sets in turn (336 MB > the 256 MB Infinity Cache): every byte from HBM",
                   "kernel_ms": t_hf_cold * 1e3, "achieved": hf_bytes / t_hf_cold / 1e9, "frac": hf_bytes / t_hf_cold / HBM_PEAK},
          "cached": {"what": "skip + h + HF(h_cnt) with HF(h_cnt) kept per content image (sid_residual_sum): same bytes, no transform "
                             "(back-to-back calls on the same 21 MB: the Infinity Cache serves part of them, here as in the fused figure)",
                     "kernel_ms": t_sum * 1e3, "achieved": hf_bytes / t_sum / 1e9, "frac": hf_bytes / t_sum / HBM_PEAK}}
    return roof, hf


def block_skip_record(dev, compute):
    """SID_FLAG_BLOCK_SKIP on and off on the top injected layer, (B,8,4096,40), key/value pairs pre-packed,
    the bench's disc mask, at three logit sizes: the unit-variance synthetic features of the headline
    workload (no block can be proven dead there: the flag only costs), features twice as large (the far
    style block is dead everywhere, the content block is not) and three times as large (the mask bias
    pushes the far style block and mostly the content block below f32 resolution).
    Not part of `value`: the headline runs without the flag."""
    import torch
    from style_transfer_amd import ops
    N, d = 4096, 40
    gen = torch.Generator(device=dev); gen.manual_seed(4242)
    yy, xx = torch.meshgrid(torch.arange(512, device=dev), torch.arange(512, device=dev), indexing="ij")
    disc = (((yy - 255.5) ** 2 + (xx - 255.5) ** 2) <= (0.3 * 512) ** 2).float().contiguous()
    mask1 = ops.mask_downsample(disc, 64, 64, 1.0).reshape(1, -1).contiguous()
    mode = compute if compute != "fp32" else "bf16"          # the flag does not apply to the f32-grade mode
    rec = {"layer": "injected attention (B,8,4096,40), disc mask r=0.3, pre-packed key/value pairs, %s" % mode, "cases": []}
    for sigma in (1.0, 2.0, 3.0):
        for B in (1, 16):
            rn = lambda *s: torch.randn(*s, device=dev, generator=gen) * sigma
            xq, q_cnt = rn(B, N, HEADS * d), rn(B * HEADS, N, d)
            kv = [ops.pack_kv(rn(B * HEADS, N, d), rn(B * HEADS, N, d), HEADS, mode) for _ in range(3)]
            mask = mask1.repeat(B, 1).contiguous()
            out = torch.empty_like(xq)
            call = lambda fl: ops.injected_attention_packed(xq, q_cnt, kv[0], kv[1], kv[2], mask, GAMMA, TEMP, d ** -0.5,
                                                            HEADS, out=out, flags=fl)
            reps = 200 if B == 1 else 20
            t_off = events_per_call(lambda: call(0), reps)
            t_on = events_per_call(lambda: call(ops.FLAG_BLOCK_SKIP), reps)
            call(ops.FLAG_BLOCK_SKIP)
            counts = ops.skip_counts(xq, HEADS, mode, packed=True)
            # the same call with the content block taken from a cached partial (two style blocks in the call)
            partial = ops.content_partial(q_cnt, kv[2], d ** -0.5, HEADS)
            ccall = lambda fl: ops.injected_attention_cached(xq, q_cnt, kv[0], kv[1], partial, mask, GAMMA, TEMP, d ** -0.5,
                                                             HEADS, out=out, flags=fl)
            tc_off = events_per_call(lambda: ccall(0), reps)
            tc_on = events_per_call(lambda: ccall(ops.FLAG_BLOCK_SKIP), reps)
            ccall(ops.FLAG_BLOCK_SKIP)
            ccounts = ops.skip_counts(xq, HEADS, mode, packed="cached")
            rec["cases"].append({"B": B, "feature_sigma": sigma, "call_ms_flag_off": t_off * 1e3, "call_ms_flag_on": t_on * 1e3,
                                 "time_ratio": t_on / t_off,
                                 "query_groups": counts[3], "groups_with_a_dead_style_block": counts[0],
                                 "content_segments_logits_only": counts[1], "of_those_redone_in_full": counts[2],
                                 "content_cached": {"call_ms_flag_off": tc_off * 1e3, "call_ms_flag_on": tc_on * 1e3,
                                                    "time_ratio": tc_on / tc_off,
                                                    "groups_with_a_dead_style_block": ccounts[0]}})
            del xq, q_cnt, kv, out, partial
            torch.cuda.empty_cache()
    # the whole hot path of an image, images in flight as in `value`, at feature-sized logits (x2 and x3): the headline's
    # loop with and without the flag (only the three N = 4096 injected layers are long enough to take it)
    n_lanes, images = max(len(_LANE_STREAMS), 1), 12
    rec["image"] = []
    for sigma in (2.0, 3.0):
        lanes = []
        for li in range(n_lanes):
            gen.manual_seed(4300 + li)
            w = make_image_workload(dev, gen, torch.float32, "packed", mode, sigma=sigma)
            st = lane_stream(dev, li)
            graphs = {}
            jobs = [(q_cnt, kv[2], layer["d"] ** -0.5) for layer in w["inj"] for q_cnt, kv in layer["feats"]]
            for (_, kvc, _), partial in zip(jobs, ops.content_partials(jobs, HEADS)):
                kvc.partial = partial
            torch.cuda.synchronize()
            both = ops.FLAG_THROUGHPUT | ops.FLAG_BLOCK_SKIP
            for name, fl, cached in (("off", ops.FLAG_THROUGHPUT, False), ("on", both, False),
                                     ("cached_off", ops.FLAG_THROUGHPUT, True), ("cached_on", both, True)):
                with torch.cuda.stream(st):
                    run_image(w, mode, fl, cached)
                torch.cuda.synchronize()
                graphs[name] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graphs[name], stream=st, capture_error_mode="thread_local"):
                    run_image(w, mode, fl, cached)
            lanes.append((st, graphs, w))
        rates = {}
        for name in ("off", "on", "cached_off", "cached_on"):
            for timed in (False, True):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(images):
                    st, graphs, _ = lanes[i % n_lanes]
                    with torch.cuda.stream(st):
                        graphs[name].replay()
                torch.cuda.synchronize()
                if timed:
                    rates[name] = images / (time.perf_counter() - t0)
        rec["image"].append({"what": "hot path of one image as in `value` (same lanes), attention tensors x%g (feature-sized logits)" % sigma,
                             "feature_sigma": sigma, "imgs_per_s_flag_off": rates["off"], "imgs_per_s_flag_on": rates["on"],
                             "ratio": rates["on"] / rates["off"],
                             "content_cached": {"imgs_per_s_flag_off": rates["cached_off"], "imgs_per_s_flag_on": rates["cached_on"],
                                                "ratio": rates["cached_on"] / rates["cached_off"],
                                                "ratio_to_three_blocks_flag_off": rates["cached_on"] / rates["off"]}})
        del lanes
        torch.cuda.empty_cache()
    return rec


# Streams of the image lanes, created first thing in a rank.  The runtime spreads a process's streams over a
# few hardware queues in creation order, so three streams made one after the other land on three queues,
# while streams made later, between other streams, may share one and then run their images one after the
# other (measured: 39 instead of 53 imgs/s).  The records below therefore run their lanes on these.
_LANE_STREAMS = []


def lane_stream(dev, li):
    import torch
    return _LANE_STREAMS[li] if li < len(_LANE_STREAMS) else torch.cuda.Stream(device=dev)


def feature_scale_record(dev, compute):
    """The headline loop again on features twice as large (q, k, v of the attention tensors at sigma 2: logits
    of the size real DDIM-inversion features produce), in both half-precision arithmetic modes, with the
    error of one top-layer call against the f32-grade kernels beside each rate.  Not part of `value`."""
    import torch
    from style_transfer_amd import ops
    sigma = 2.0
    gen = torch.Generator(device=dev)
    n_lanes, images = max(len(_LANE_STREAMS), 1), 12
    rec = {"what": "hot path of one image as in `value` (same lanes, same graphs), attention tensors x2 (feature-sized logits)",
           "feature_sigma": sigma, "modes": {}}
    for mode in ("f16", "bf16"):
        lanes = []
        for li in range(n_lanes):
            gen.manual_seed(6100 + li)
            w = make_image_workload(dev, gen, torch.float32, "packed", mode, sigma=sigma)
            st = lane_stream(dev, li)
            with torch.cuda.stream(st):
                run_image(w, mode, ops.FLAG_THROUGHPUT)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
                run_image(w, mode, ops.FLAG_THROUGHPUT)
            lanes.append((st, graph, w))
        rate = None
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(images):
                st, graph, _ = lanes[i % n_lanes]
                with torch.cuda.stream(st):
                    graph.replay()
            torch.cuda.synchronize()
            if timed:
                rate = images / (time.perf_counter() - t0)
        # one call of the top injected layer against the f32-grade kernels on the same tensors
        gen.manual_seed(6200)
        rn = lambda *s: torch.randn(*s, device=dev, generator=gen) * sigma
        N, d = 4096, 40
        xq, feats = rn(1, N, HEADS * d), [rn(HEADS, N, d) for _ in range(7)]
        mask = lanes[0][2]["inj"][-1]["mask"]
        call = lambda c: ops.injected_attention(xq, *feats, mask, GAMMA, TEMP, d ** -0.5, HEADS, compute=c)
        ref = call("fp32")
        got = call(mode)
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        row = ((got - ref).abs().amax(-1) / ref.abs().amax(-1).clamp_min(1e-30)).max().item()
        rec["modes"][mode] = {"imgs_per_s": rate, "rel_err_vs_f32_grade": err, "per_row_rel_err_vs_f32_grade": row}
        del lanes, xq, feats, ref, got
        torch.cuda.empty_cache()
    return rec


def default_queue_record(args):
    """The headline loop in a child process that leaves GPU_MAX_HW_QUEUES alone (the runtime's default of four
    hardware queues per process): what a caller who does not set the variable gets with the same lanes."""
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    cmd = [sys.executable, os.path.abspath(__file__), "--hw-queues", "0", "--headline-only", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--compute", args.compute, "--io", args.io, "--store", args.store,
           "--in-flight", str(args.in_flight)] + (["--no-graph"] if args.no_graph else [])
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        js = json.loads(r.stdout.strip().splitlines()[-1])
        return {"what": "same loop, child process without GPU_MAX_HW_QUEUES (runtime default: 4 queues)", "value": js["value"],
                "hardware_queues_per_process": js["hardware_queues_per_process"], "images_in_flight": js["images_in_flight"]}
    except Exception as e:       # the side record must not take the line down
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def content_cache_record(dev, compute):
    """The cached content block (sid_content_partial + sid_injected_attn_fwd_cached): the third key block of
    the injected attention reads the content image's feature file only, so a content image that meets
    several style pairs computes it once.  Call level at (B,8,4096,40), then the whole hot path of an image
    with three images in flight, content blocks cached against computed.  Not part of `value`: the headline
    is one content with one style pair and computes all three blocks in every call."""
    import torch
    from style_transfer_amd import ops
    N, d = 4096, 40
    gen = torch.Generator(device=dev); gen.manual_seed(5151)
    rec = {"layer": "injected attention (B,8,4096,40), pre-packed key/value pairs, %s" % compute, "cases": []}
    for B in (1, 16):
        rn = lambda *s: torch.randn(*s, device=dev, generator=gen)
        xq, q_cnt = rn(B, N, HEADS * d), rn(B * HEADS, N, d)
        kv = [ops.pack_kv(rn(B * HEADS, N, d), rn(B * HEADS, N, d), HEADS, compute) for _ in range(3)]
        mask = torch.sign(rn(B, N)).contiguous()
        out = torch.empty_like(xq)
        partial = ops.content_partial(q_cnt, kv[2], d ** -0.5, HEADS)
        reps = 200 if B == 1 else 20
        t3 = events_per_call(lambda: ops.injected_attention_packed(xq, q_cnt, kv[0], kv[1], kv[2], mask, GAMMA, TEMP, d ** -0.5,
                                                                   HEADS, out=out), reps)
        t2 = events_per_call(lambda: ops.injected_attention_cached(xq, q_cnt, kv[0], kv[1], partial, mask, GAMMA, TEMP, d ** -0.5,
                                                                   HEADS, out=out), reps)
        tp = events_per_call(lambda: ops.content_partial(q_cnt, kv[2], d ** -0.5, HEADS), max(reps // 4, 5))
        rec["cases"].append({"B": B, "call_ms_three_blocks": t3 * 1e3, "call_ms_content_cached": t2 * 1e3, "time_ratio": t2 / t3,
                             "partial_ms": tp * 1e3, "flop_computed_per_call": 8.0 * B * HEADS * N * N * d})
        del xq, q_cnt, kv, out, partial
        torch.cuda.empty_cache()
    n_lanes, images = max(len(_LANE_STREAMS), 1), 12
    lanes = []
    t_partials = None
    for li in range(n_lanes):
        gen.manual_seed(5200 + li)
        w = make_image_workload(dev, gen, torch.float32, "packed", compute)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        jobs = [(q_cnt, kv[2], layer["d"] ** -0.5) for layer in w["inj"] for q_cnt, kv in layer["feats"]]
        for (_, kvc, _), partial in zip(jobs, ops.content_partials(jobs, HEADS)):      # four side streams
            kvc.partial = partial
        torch.cuda.synchronize()
        t_partials = time.perf_counter() - t0
        st = lane_stream(dev, li)
        graphs = {}
        for name, cached in (("computed", False), ("cached", True)):
            with torch.cuda.stream(st):
                run_image(w, compute, ops.FLAG_THROUGHPUT, cached)
            torch.cuda.synchronize()
            graphs[name] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graphs[name], stream=st, capture_error_mode="thread_local"):
                run_image(w, compute, ops.FLAG_THROUGHPUT, cached)
        lanes.append((st, graphs, w))
    rates = {}
    for name in ("computed", "cached"):
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(images):
                st, graphs, _ = lanes[i % n_lanes]
                with torch.cuda.stream(st):
                    graphs[name].replay()
            torch.cuda.synchronize()
            if timed:
                rates[name] = images / (time.perf_counter() - t0)
    part_bytes = sum(kv[2].partial.nbytes() for layer in lanes[0][2]["inj"] for _, kv in layer["feats"])
    rec["image"] = {"what": "hot path of one image as in `value` (same lanes); `cached`: the content blocks of its 294 "
                            "injected calls come from partials made once per content image",
                    "imgs_per_s_content_computed": rates["computed"], "imgs_per_s_content_cached": rates["cached"],
                    "ratio": rates["cached"] / rates["computed"],
                    "partials_s_per_content_image": t_partials, "partials_GB_per_content_image": part_bytes / 1e9,
                    "style_pairs_to_break_even": t_partials / max(1.0 / rates["computed"] - 1.0 / rates["cached"], 1e-9)}
    del lanes
    torch.cuda.empty_cache()
    return rec


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE and WRITE_SIZE,
    each in its own rocprofv3 run, corrected for gfx950): counters cannot be read inside this run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_flash_traffic.json")))
    if not files:
        return {"traffic": None}
    with open(files[-1]) as f:
        js = json.load(f)
    # mfma_busy: matrix-pipe busy cycles over the dispatch's SIMD-cycles (SQ_VALU_MFMA_BUSY_CYCLES against
    # GRBM_GUI_ACTIVE), the hardware's own utilisation figure beside the algorithmic `frac` (which counts
    # neither the padding of head_dim 40 to the MFMA shapes nor the clock the chip sustains)
    out = {"traffic": js.get("traffic_bytes"), "mfma_busy": js.get("mfma_busy"), "valu_issue_busy": js.get("valu_issue_busy"),
           "traffic_source": "profiles/" + os.path.basename(files[-1])}
    # Where a workgroup's time goes, from the diagnostic build's clock stamps (tools/diag_flash.py; not measurable in
    # the product build, which executes no stamp): time inside the stage loops against the kernel's span, and the
    # rest of a workgroup's lifetime (first fill, reference pre-pass and row store of each of its segments).
    phases = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_flash_phases.json")))
    if phases:
        with open(phases[-1]) as f:
            ph = json.load(f)
        out.update(steady_loop_frac=ph.get("steady_loop_frac"), fixed_cost_us=ph.get("fixed_cost_us_per_workgroup"),
                   phases_source="profiles/" + os.path.basename(phases[-1]))
    # The clock the launch is given: the same cycle count whatever the operands hold, at a clock that follows the
    # switching in the multipliers (same diagnostic build, operands scaled to zero): `frac` is set by the power
    # management, not by the schedule.  frac_at_zero_operand_clock scales the live figure by clock alone.
    clocks = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_clock_by_operands.json")))
    if clocks:
        with open(clocks[-1]) as f:
            ck = json.load(f)
        out["clock"] = {"random_operands_GHz": ck["B1"]["random_operands_GHz"], "zero_operands_GHz": ck["B1"]["zero_operands_GHz"],
                        "cycles_per_workgroup": ck["B1"]["cycles_per_workgroup"], "peak_assumes_GHz": ck["peak_assumes_GHz"],
                        "source": "profiles/" + os.path.basename(clocks[-1])}
    return out


_HOST = {}     # the host UNet of the e2e record, kept for the sweep record (860 M parameters: built once)


def host_unet(dev, compute):
    import torch
    from style_transfer_amd import unet as U
    if "model" not in _HOST:
        torch.manual_seed(0)
        model = U.UNetModel().eval().to(dev)
        for a in model.attention_modules():
            a.compute = compute
        _HOST["model"] = model
    return _HOST["model"]


def e2e_record(dev, compute, store):
    """The whole stylisation loop of run_ori.py around the path, driver-timed: this package's host UNet
    (SD-v1 topology and size, seeded random fp32 weights: no checkpoint exists offline) under DDIM, masked
    AdaIN init, synthetic feature lists in the precomputed_feats_k layout, 49 evaluations per image; one
    image at a time and four in flight (one graphed evaluation, stream and workspace per lane)."""
    import tempfile
    import types
    import numpy as np
    import torch
    from style_transfer_amd import feature_store as fs, ops, synth
    from style_transfer_amd.features import masked_adain_init
    from style_transfer_amd.graphed import GraphedUNet
    from style_transfer_amd.sampler import DDIMSampler
    g = torch.Generator(device=dev)

    def feature_list(seed):
        g.manual_seed(seed)
        out = []
        for _ in range(50):
            e = {"config": {"T": TEMP}, "z_enc": torch.randn(1, 4, 64, 64, device=dev, generator=g)}
            for b in range(6, 12):
                n, d = (1024, 80) if b < 9 else (4096, 40)
                for t in "qkv":
                    e["output_block_%d_self_attn_%s" % (b, t)] = torch.randn(8, n, d, device=dev, generator=g)
            out.append(e)
        return out

    fstore = fs.FeatureStore(feature_list(1), feature_list(2), feature_list(3))
    z = fstore.z_enc()
    if store == "packed":
        fstore.pack(HEADS, compute, T=TEMP)
    merged = fstore.merged(types.SimpleNamespace(gamma=GAMMA, T=TEMP), start_step=49)
    mask = synth.disc_mask()
    td = tempfile.mkdtemp()
    np.save(os.path.join(td, "img_mask.npy"), mask)
    model = host_unet(dev, compute)
    model.set_content_name(os.path.join(td, "img.jpg"))
    s = DDIMSampler(model)
    s.make_schedule(50, 0.0)
    ctx = torch.randn(1, 77, 768, device=dev, generator=g)
    mask_dev = torch.from_numpy(mask).to(dev)
    x_T = lambda: masked_adain_init(z[0], z[1], z[2], mask_dev)

    def timed(sampler, images=2):
        one = lambda: sampler.sample(x_T(), ctx, injected_features=merged, start_step=49)
        one()                            # warm-up: the convolution library picks its kernels here (and graph capture)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(images):
            out = one()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        return (time.perf_counter() - t0) / images

    def timed_lanes(n, rounds=2):
        """`n` images in flight: sample_lanes over n graphed evaluations of the one network."""
        lanes = [GraphedUNet(model) for _ in range(n)]
        streams = [lane_stream(dev, i) for i in range(n)]
        batch = lambda: s.sample_lanes(lanes, [x_T() for _ in range(n)], [ctx] * n, [merged] * n, start_step=49, streams=streams)
        batch()                          # capture per lane
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(rounds):
            outs = batch()
        torch.cuda.synchronize()
        assert all(torch.isfinite(o).all() for o in outs)
        return (time.perf_counter() - t0) / (rounds * n)

    dt = timed(s)                        # launched kernel by kernel
    lanes_rec = {}
    for n in (1, 4):
        for a in model.attention_modules():
            a.kernel_flags = ops.FLAG_THROUGHPUT if n > 1 else 0
        per = {"fp32": timed_lanes(n)}
        with torch.autocast("cuda", dtype=torch.bfloat16):     # the host network's convolutions and linears in bf16
            per["bf16_autocast"] = timed_lanes(n)
        lanes_rec[str(n)] = {k: {"imgs_per_s": 1.0 / v, "s_per_image": v} for k, v in per.items()}
    for a in model.attention_modules():
        a.kernel_flags = 0
    best = lanes_rec["4"]["fp32"]
    _HOST.update(ctx=ctx, tmp=td)
    return {"value": best["imgs_per_s"], "unit": "imgs/s", "s_per_image": best["s_per_image"],
            "lanes": lanes_rec,
            "eager_host_unet": {"value": 1.0 / dt, "s_per_image": dt},
            "what": "masked AdaIN init + 49 evaluations of the host UNet (SD-v1 size, random fp32 weights, stock PyTorch "
                    "convolutions) with the HIP attention kernels, every evaluation replayed as one HIP graph; `value`: four "
                    "images in flight, fp32 network; lanes: images in flight x {fp32, network under bf16 autocast}; "
                    "eager_host_unet: one image at a time launched kernel by kernel; VAE and text encoder not run"}


def sweep_record(dev, compute, n_cnt=2, n_char=2, n_back=2, reference_lines=3):
    """configs[3] in small: the run.py loop (style_transfer_amd.run.run_lines: loader, masked AdaIN, merge, 49
    graphed evaluations per line) over 8 lines = 2 contents x 2 character styles x 2 background styles, whose
    six SD-v1-sized feature files (3.54 GB pickles written by save_feature_file) sit in tmpfs; imgs/s
    INCLUDING the feature load, by route, beside the compute-only rate.  Skipped when tmpfs has no room."""
    import shutil
    import tempfile
    import numpy as np
    import torch
    from style_transfer_amd import feature_store as fs, run as cli, synth
    from style_transfer_amd.sampler import DDIMSampler
    base_dir = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    free = shutil.disk_usage(base_dir).free
    need = int((n_cnt + n_char + n_back) * 3.6e9 * 1.6) + (8 << 30)          # pickles + side-cars + slack
    if free < need:
        return {"skipped": "%s has %.0f GB free; the sweep needs %.0f" % (base_dir, free / 1e9, need / 1e9)}
    root = tempfile.mkdtemp(prefix="sid_sweep_", dir=base_dir)
    try:
        data, pre = os.path.join(root, "data"), os.path.join(root, "pre")
        for sub in ("cnt", "sty/char", "sty/back"):
            os.makedirs(os.path.join(data, sub))
        os.makedirs(pre)
        g = torch.Generator(device=dev)

        def feature_list(seed):
            g.manual_seed(seed)
            out = []
            for _ in range(50):
                e = {"config": {"T": TEMP}, "z_enc": torch.randn(1, 4, 64, 64, device=dev, generator=g)}
                for b in range(6, 12):
                    n, d = (1024, 80) if b < 9 else (4096, 40)
                    for t in "qkv":
                        e["output_block_%d_self_attn_%s" % (b, t)] = torch.randn(8, n, d, device=dev, generator=g)
                out.append(e)
            return out

        t0 = time.perf_counter()
        names = {"cnt": ["c%d" % i for i in range(1, n_cnt + 1)], "sty/char": ["a%d" % i for i in range(1, n_char + 1)],
                 "sty/back": ["b%d" % i for i in range(1, n_back + 1)]}
        for sub, lst in names.items():
            for i, n in enumerate(lst):
                open(os.path.join(data, sub, n + ".png"), "wb").write(b"x")
                fs.save_feature_file(os.path.join(pre, n + ("_cnt.pkl" if sub == "cnt" else "_sty.pkl")),
                                     feature_list(7000 + 10 * len(sub) + i))
        for n in names["cnt"]:
            np.save(os.path.join(data, "cnt", n + "_mask.npy"), synth.disc_mask())
        lines = ["%s.png %s.png %s.png" % (c, a, b) for a in names["sty/char"] for b in names["sty/back"] for c in names["cnt"]]
        with open(os.path.join(data, "dataset.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
        pkl_bytes = sum(os.path.getsize(os.path.join(pre, n)) for n in os.listdir(pre))
        t_write = time.perf_counter() - t0
        model = host_unet(dev, compute)
        sampler = DDIMSampler(model)
        sampler.make_schedule(50, 0.0)
        uc = _HOST.get("ctx")
        if uc is None:
            uc = torch.zeros(1, 77, 768, device=dev)
        triples = cli.list_triples(data)
        common = ["--data_root", data, "--precomputed", pre, "--output_path", os.path.join(root, "out"), "--gamma", str(GAMMA),
                  "--T", str(TEMP), "--compute", compute, "--graph_unet"] + (["--progress"] if "--progress" in sys.argv else [])
        os.makedirs(os.path.join(root, "out"))

        def run(extra, lines_idx):
            sys.stderr.write("sweep: %d lines, %s\n" % (len(lines_idx), " ".join(extra)))      # (a long silent run looks hung)
            sys.stderr.flush()
            opt = cli.build_parser().parse_args(common + extra)
            torch.cuda.synchronize()
            t = time.perf_counter()
            _, st = cli.run_lines(opt, model, sampler, triples, lines_idx, uc, dev)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            return {"lines": len(lines_idx), "s": dt, "imgs_per_s": len(lines_idx) / dt, "files_read": st["files"],
                    "GB_read": st["file_bytes"] / 1e9, "lists_served_from_hbm": st["hits"], "side_cars_read": st["routes"]["sidecar"],
                    "load_GB_per_s": st["file_bytes"] / 1e9 / max(st["load_seconds"], 1e-9),
                    "sampler_waited_s": st["wait_seconds"]}

        every = list(range(len(triples)))
        rec = {"what": "run.py loop over %d lines (%d contents x %d character x %d background styles), %d feature pickles of 3.54 GB in "
                       "%s, one image at a time unless stated, every evaluation a HIP graph; imgs/s include reading, uploading and "
                       "packing the feature files" % (len(lines), n_cnt, n_char, n_back, n_cnt + n_char + n_back, base_dir),
               "pickle_GB": pkl_bytes / 1e9, "files_written_s": t_write}
        run(["--no_prefetch", "--sidecar", "off", "--store_cache_gb", "0"], every[:1])       # warm-up: graph capture, conv kernels
        # the reference's order: every line reads its three pickles when its turn comes, nothing is kept (run_ori.py:168-222)
        rec["reference_order"] = run(["--no_prefetch", "--sidecar", "off", "--store_cache_gb", "0"], every[:reference_lines])
        rec["reference_order"]["note"] = "first %d lines only (every line costs the same: 3 pickles)" % reference_lines
        rec["prefetch_lru_pickles"] = run(["--sidecar", "write"], every)                     # also writes the side-cars
        rec["prefetch_lru_sidecars"] = run(["--sidecar", "use"], every)
        rec["prefetch_lru_sidecars_4_lanes"] = run(["--sidecar", "use", "--lanes", "4"], every)
        rec["prefetch_lru_sidecars_4_lanes_network_autocast"] = run(["--sidecar", "use", "--lanes", "4", "--network_autocast"], every)
        rec["sidecar_GB"] = sum(os.path.getsize(os.path.join(pre, n)) for n in os.listdir(pre) if n.endswith(".sidpack")) / 1e9
        # compute only: the same lines with every list already resident (a second pass over a warm loader is not what
        # run_lines does, so: the e2e record's rate for one image at a time)
        return rec
    finally:
        shutil.rmtree(root, ignore_errors=True)


def cpu_baseline():
    """The C oracle (oracle/, a port of the reference arithmetic) timed on this host's cores on a
    bounded sample of query rows / feature maps of the same workload, scaled to one image."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sid_oracle_c as oc
    rng = np.random.default_rng(0)
    per_image = 0.0
    sample, factors = [], {}

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    for (N, d), count in (((4096, 40), 3), ((1024, 80), 3)):
        rows = 512 if N == 4096 else 256
        a = [rng.standard_normal((1, N, HEADS * d), dtype=np.float32)[0]] + \
            [rng.standard_normal((HEADS, N, d), dtype=np.float32) for _ in range(7)]
        m = np.sign(rng.standard_normal(N)).astype(np.float32)
        oc.injected_attention_rows(0, 16, *a, m, GAMMA, TEMP, d ** -0.5)
        t = timed(lambda: oc.injected_attention_rows(0, rows, *a, m, GAMMA, TEMP, d ** -0.5))
        per_image += EVALS * count * t * N / rows
        factors["injected_N%d" % N] = EVALS * count * N / rows
        sample.append("injected N=%d d=%d: %d of %d rows of one call" % (N, d, rows, N))
    for (N, d, T) in sorted(set(PLAIN)):
        count = PLAIN.count((N, d, T))
        rows = min(N, 256)
        q, k, v = (rng.standard_normal((HEADS, N, d), dtype=np.float32) for _ in range(3))
        t = timed(lambda: oc.plain_attention_rows(0, rows, q, k, v, d ** -0.5, T))
        per_image += EVALS * count * t * N / rows
        factors["plain_N%d_T%g" % (N, T)] = EVALS * count * N / rows
    sample.append("plain self-attention: <=256 rows of one call per shape")
    for (C, S) in sorted(set(HF_MAPS)):
        maps = 64
        x, s, h = (rng.standard_normal((1, maps, S, S), dtype=np.float32) for _ in range(3))
        t = timed(lambda: oc.hf_filter(x, RATIO, s, h))
        per_image += HF_EVALS * 3 * t * C / maps
        factors["hf_%dx%d" % (S, S)] = HF_EVALS * 3 * C / maps
    sample.append("HF filter: 64 maps of one call per shape")
    return {"value": 1.0 / per_image, "unit": "imgs/s", "cores": oc.threads(), "kind": "port",
            "sample": "; ".join(sample) + "; every timed sample is multiplied by the factor below to give its share of "
                      "one image (49 evaluations)",
            "extrapolation_factors": factors}


def run_rank(args, world, rank, local):
    import torch
    from style_transfer_amd import lib as sid_lib
    from style_transfer_amd import ops
    from style_transfer_amd import sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the path has no CPU fallback)")
    sid_lib.load()
    if os.environ.get("SID_BENCH_ONE_DEVICE") == "1":      # rehearsal of the N-rank flow on a one-GPU box (with --backend gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    gen = torch.Generator(device=dev)
    dtype = torch.float32 if args.io == "f32" else torch.bfloat16

    # one lane per image in flight: a stream, its own workload, and the image's launch sequence as a graph
    n_lanes = max(1, args.in_flight)
    kflags = ops.FLAG_THROUGHPUT if n_lanes > 1 else 0      # images share the GPU: least-work decomposition
    lanes = []
    graph_error = None
    for li in range(n_lanes):
        gen.manual_seed(1234 + 100 * rank + li)
        w = make_image_workload(dev, gen, dtype, args.store, args.compute)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            run_image(w, args.compute, kflags)       # sizes this stream's workspace; first-use kernel attributes
        torch.cuda.synchronize()
        graph, gout = None, None
        if not args.no_graph and graph_error is None:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
                    gout = run_image(w, args.compute, kflags)
            except Exception as e:                   # same kernels, launched one by one instead
                graph, gout, graph_error = None, None, "%s: %s" % (type(e).__name__, e)
                torch.cuda.synchronize()
        lanes.append((st, graph, gout, w))
    torch.cuda.synchronize()
    _LANE_STREAMS[:] = [lane[0] for lane in lanes]

    # the process group comes after graph capture, so no communicator thread touches the device meanwhile
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == world

    # which collectives ran for the first time inside the timed region (their first use sets up channels)
    seen_ops, first_in_timed, clock = set(), [], {"on": False}

    def note(op):
        if dist is not None and op not in seen_ops:
            seen_ops.add(op)
            if clock["on"]:
                first_in_timed.append(op)

    def barrier():
        if dist is not None:
            note("barrier")
            dist.barrier()
        torch.cuda.synchronize()

    def one_image(i):
        st, graph, gout, w = lanes[i % n_lanes]
        with torch.cuda.stream(st):
            if graph is not None:
                graph.replay()
                out = gout
            else:
                out = run_image(w, args.compute, kflags)
            return out.reshape(-1)[:4 * 64 * 64].clone()    # stand-in for the (1,4,64,64) latent, 64 KiB

    def to_wire(t):
        return t.cpu() if dist is not None and args.backend == "gloo" else t

    for i in range(args.warmup):
        one_image(i)
    # Rehearsal of the region's tail with the shapes it will have: the stack of `steps` latents (the allocator
    # gets its block now, not inside the region) and the job's one collective: the first gather of a process
    # group sets up its point-to-point channels, and the first all_reduce its ring; none of that belongs to the
    # timed region (with 20 images the region is under half a second).
    dummy = to_wire(torch.stack([torch.zeros(4 * 64 * 64, device=dev) for _ in range(args.steps)]).float())
    if dist is not None:
        note("gather")
    sharding.gather_latents(dummy, dist, dst=0)
    del dummy
    if dist is not None:
        tw = torch.zeros(1, device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        note("all_reduce")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    barrier()
    clock["on"] = True
    t0 = time.perf_counter()
    latents = [one_image(i) for i in range(args.steps)]
    torch.cuda.synchronize()
    t_images = time.perf_counter() - t0                    # this rank's own images, before the gather
    mine = to_wire(torch.stack(latents).float())
    note("gather")
    gathered = sharding.gather_latents(mine, dist, dst=0)  # the one collective of the job (RCCL gather)
    if rank == 0:
        assert gathered.shape[0] == world and gathered.shape[1] == args.steps
    barrier()
    elapsed = time.perf_counter() - t0
    clock["on"] = False
    rank_rates = [args.steps / t_images]
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine_rate = torch.tensor([args.steps / t_images], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        rates = [torch.zeros_like(mine_rate) for _ in range(world)]
        dist.all_gather(rates, mine_rate)                  # after the clock has stopped
        rank_rates = [float(r.item()) for r in rates]

    if rank == 0 and args.headline_only:
        print(json.dumps({"value": world * args.steps / elapsed, "unit": "imgs/s", "n_gpus": world, "steps": args.steps,
                          "ms_per_step": 1e3 * elapsed / args.steps,
                          "hardware_queues_per_process": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                          "images_in_flight": n_lanes, "compute": args.compute}), flush=True)
    elif rank == 0:
        roof, hf = kernel_records(lanes[0][3], args.compute)
        store_bytes = sum(kv.nbytes() if isinstance(kv, ops.PackedKV) else kv[0].numel() * kv[0].element_size() * 2
                          for layer in lanes[0][3]["inj"] for (_, kvs) in layer["feats"] for kv in kvs)
        line = {
            "metric": "stylized imgs/sec (512^2, 50 DDIM), injection hot path",
            "value": world * args.steps / elapsed,
            "unit": "imgs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"bf16": "bf16", "f16": "f16 (q.k^T operands of the long-sequence kernels) + bf16",
                      "fp32": "bf16x3 (f32-grade)"}[args.compute],
            "data": "synthetic",
            "config": {"workload": "configs[1]: one content + char/back style pair, 512^2, gamma=0.5 T=1.5 ratio=0.5, "
                                   "49 UNet evaluations: 6 injected + 10 self attention calls each, fused HF filter "
                                   "on 6 ResBlocks for 19 evaluations; hot path only (UNet convs/VAE not run; see e2e)",
                       "query_storage": args.io,
                       "feature_store": ("key/value pairs pre-packed to bf16 operand layout at load time (%.2f GB per image)"
                                         if args.store == "packed" else "fp32 tensors (%.2f GB per image)") % (store_bytes / 1e9),
                       "images_per_rank": args.steps, "images_in_flight": n_lanes, "workloads_per_rank": n_lanes,
                       "hardware_queues_per_process": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "ranks": world if dist is None else dist.get_world_size(),
                       "launch": ("kernel by kernel" if args.no_graph else "one HIP graph per image") if graph_error is None
                       else "kernel by kernel (graph capture failed: %s)" % graph_error[:120],
                       "collective": "one gather of 64 KiB latents per image" if world > 1 else "none"},
            # every rank's own rate over its images (before the gather): a straggler shows here, `value` divides by
            # the slowest rank's whole region
            "per_rank": {"imgs_per_s": rank_rates, "min": min(rank_rates), "max": max(rank_rates),
                         "collectives_warmed_before_t0": sorted(seen_ops - set(first_in_timed)) if world > 1 else None,
                         "first_use_collectives_in_timed_region": first_in_timed},
            "roofline": roof,
            "hf_filter": hf,
        }
        if world == 1:
            del lanes[1:]
            torch.cuda.empty_cache()
            line["feature_scale"] = None if args.no_feature_scale else feature_scale_record(dev, args.compute)
            line["block_skip"] = None if args.no_block_skip else block_skip_record(dev, args.compute)
            line["content_cache"] = None if args.no_content_cache else content_cache_record(dev, args.compute)
            line["e2e"] = None if args.no_e2e else e2e_record(dev, args.compute, args.store)
            line["sweep"] = None if args.no_sweep else sweep_record(dev, args.compute)
            _HOST.clear()
            torch.cuda.empty_cache()
            line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline()
            line["default_hw_queues"] = None if args.no_default_queues or args.hw_queues == 0 else default_queue_record(args)
        else:
            line["e2e"] = None
            line["sweep"] = None
            line["feature_scale"] = None
            line["default_hw_queues"] = None
            line["block_skip"] = None
            line["content_cache"] = None
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
