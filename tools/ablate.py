"""Throughput of bench.py's default configuration with parts of the per-image work left out
(measurement only; the rows computed are not meaningful):  python tools/ablate.py
Prints imgs/s for: everything | no merge stage | no pack stage | flash stage only | no plain attentions |
no HF sums | only the N=4096 injected attentions."""
import io, json, os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from style_transfer_amd import ops

full = bench.run_image


def variant(extra_flags=0, plain=True, hf=True, inj=(True, True), plain_keep=lambda N: True):
    state = {"calls": 0}

    def run(w, compute, flags=0, cached=False):
        # the eager call bench.py makes per lane before capture runs everything once, so that the
        # workspaces hold valid packed images and partials for the stage-subset replays
        if not torch.cuda.is_current_stream_capturing():
            return full(w, compute, flags)
        outs = None
        ops.masked_adain(w["z"][0], w["z"][1], w["z"][2], w["mask512"])
        for e in range(bench.EVALS):
            if plain:
                for p in w["plain"]:
                    if not plain_keep(p["N"]):
                        continue
                    q, k, v = p["qkv"][e]
                    ops.plain_attention(q, k, v, p["T"] * p["d"] ** -0.5, bench.HEADS, compute=compute,
                                        flags=flags | extra_flags)
            hf_on = bench.HF_FIRST <= e < bench.HF_FIRST + bench.HF_EVALS
            for li, layer in enumerate(w["inj"]):
                if hf_on and hf:
                    m = w["hf"][li]
                    ops.hf_filter(m["h_cnt"][e - bench.HF_FIRST], bench.RATIO, skip=m["skip"], h=m["h"])
                if not inj[0 if layer["N"] == 4096 else 1]:
                    continue
                outs = bench.injected_call(layer, e, compute, flags | extra_flags)
        return outs
    return run


lanes = os.environ.get("ABLATE_LANES", "4")
CASES = [("everything", full), ("no merge stage", variant(0x30)), ("no pack stage", variant(0x60)),
         ("flash stage only", variant(0x20)), ("no plain attentions", variant(0, plain=False)),
         ("no plain N=4096", variant(0, plain_keep=lambda N: N != 4096)),
         ("no plain N=1024", variant(0, plain_keep=lambda N: N != 1024)),
         ("no plain N<=256", variant(0, plain_keep=lambda N: N > 256)),
         ("no HF sums", variant(0, hf=False)), ("only N=4096 injected", variant(0, plain=False, hf=False, inj=(True, False))),
         ("only N=4096 injected, flash only", variant(0x20, plain=False, hf=False, inj=(True, False)))]
if os.environ.get("ABLATE_NO_THROUGHPUT_FLAG") == "1":
    ops.FLAG_THROUGHPUT = 0
if os.environ.get("ABLATE_CASES"):
    CASES = [c for i, c in enumerate(CASES) if str(i) in os.environ["ABLATE_CASES"].split(",")]
else:
    # one fresh process per case: the runtime spreads a process's streams over its hardware queues in creation
    # order, and the lane streams of a second bench.main() in one process can land on shared queues (lanes that
    # share a queue run one after the other), which would be measured as part of the case
    import subprocess
    for i in range(len(CASES)):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, ABLATE_CASES=str(i)),
                           capture_output=True, text=True)
        sys.stdout.write(r.stdout if r.returncode == 0 else "case %d failed: %s\n" % (i, r.stderr[-300:]))
        sys.stdout.flush()
    sys.exit(0)
sys.argv = [sys.argv[0], "--headline-only", "--steps", os.environ.get("ABLATE_STEPS", "45"), "--warmup", "15", "--in-flight", lanes]
for name, fn in CASES:
    bench.run_image = fn
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(sys.argv[1:])
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    print("%-36s %6.1f imgs/s  %6.2f ms per image" % (name, line["value"], line["ms_per_step"]), flush=True)
