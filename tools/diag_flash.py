"""Diagnostic build only (exp/diag.so, -DSID_DIAG): per-workgroup clock stamps of the flash kernel.
stamps per segment: 0 start, 1 first stages landed, 2 reference pre-pass done, 3 steady loop done, 4 tails done, 5 stored.
memrealtime ticks at 100 MHz (10 ns).
    python tools/diag_flash.py [N d B [plain]]      plain: the single-block call instead of the injected one"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
N, d, B = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 1024), (2, 80), (3, 1)))
dev = torch.device("cuda:0")
diag = torch.zeros(24 * 65536, dtype=torch.int64, device=dev)
os.environ["SID_DIAG_PTR"] = str(diag.data_ptr())
os.environ.setdefault("SID_LIB_OVERRIDE", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exp", "diag.so"))      # (exp/ does not travel to the GPU box: copy the build to ab/ and point SID_LIB_OVERRIDE at it)
from style_transfer_amd import ops
H = 8
g = torch.Generator(device=dev); g.manual_seed(0)
scale = float(os.environ.get("SID_DIAG_SCALE", "1"))      # 0: all-zero operands (same instructions, no switching in the multipliers)
rn = lambda *s: torch.randn(*s, device=dev, generator=g) * scale
xq = rn(B, N, H * d); f = [rn(B * H, N, d) for _ in range(7)]; m = torch.sign(rn(B, N)).contiguous()
plain = len(sys.argv) > 4 and sys.argv[4] == "plain"
pq, pk, pv = (rn(B, N, H * d) for _ in range(3))
call = (lambda: ops.plain_attention(pq, pk, pv, d ** -0.5, H)) if plain else (lambda: ops.injected_attention(xq, *f, m, 0.5, 1.5, d ** -0.5, H))
for _ in range(5):
    call()
torch.cuda.synchronize()
diag.zero_()
call()
torch.cuda.synchronize()
raw = diag.cpu().numpy().reshape(-1, 3, 8).astype(np.float64)      # [workgroup][segment][stamp]
raw = raw[raw[:, 0, 0] > 0]
nseg = (raw[:, :, 5] > 0).sum(axis=1).astype(int)                   # segments a workgroup ran
last = raw[np.arange(len(raw)), nseg - 1]                           # the phases below are those of the last segment
cyc = last[:, 7] - last[:, 6]
t = last[:, :6].copy()
t[:, 0] = raw[:, 0, 0]                                              # ... but a workgroup starts with its first
t0 = t[:, 0].min()
print("workgroups", len(t), " kernel span %.2f us" % ((t[:, 5].max() - t0) / 100))
names = ["entry->stage0 landed", "first stage", "steady loop", "tails", "store"]
for i, nm in enumerate(names):
    dlt = (t[:, i + 1] - t[:, i]) / 100
    print("  %-22s mean %.2f us  min %.2f  max %.2f" % (nm, dlt.mean(), dlt.min(), dlt.max()))
print("  wg lifetime            mean %.2f us  max %.2f" % (((t[:, 5] - t[:, 0]) / 100).mean(), ((t[:, 5] - t[:, 0]) / 100).max()))
print("  entry time spread      first %.2f us last %.2f us" % (0.0, (t[:, 0].max() - t0) / 100))
print("  end time spread        first %.2f us last %.2f us" % ((t[:, 5].min() - t0) / 100, (t[:, 5].max() - t0) / 100))
# all segments of a workgroup: time inside the stage loops (pipelined steady loop + tail stages) against its lifetime
seg_ok = raw[:, :, 5] > 0
loop_us = (((raw[:, :, 4] - raw[:, :, 2]) * seg_ok).sum(axis=1)) / 100
life_us = (t[:, 5] - t[:, 0]) / 100
span_us = (t[:, 5].max() - t0) / 100
summary = {"N": N, "d": d, "B": B, "call": "plain" if plain else "injected", "workgroups": int(len(t)), "kernel_span_us": float(span_us),
           "stage_loops_us_per_workgroup": float(loop_us.mean()), "lifetime_us_per_workgroup": float(life_us.mean()),
           "fixed_cost_us_per_workgroup": float((life_us - loop_us).mean()),
           "steady_loop_frac": float(loop_us.mean() / span_us), "segments_per_workgroup": float(nseg.mean())}
steady = (t[:, 3] - t[:, 2]) * 10e-9
ok = steady > 0
summary["shader_clock_GHz_in_the_loop"] = float((cyc[ok] / steady[ok]).mean() / 1e9)
if os.environ.get("SID_DIAG_JSON"):
    import json
    with open(os.environ["SID_DIAG_JSON"], "w") as jf:
        json.dump(summary, jf)
print("  summary", summary)
print("  shader clock in the steady loop: %.2f GHz (mean over workgroups), cycles per workgroup %.0f" % ((cyc[ok] / steady[ok]).mean() / 1e9, cyc[ok].mean()))
# per-run breakdown (the mapping of the kernel: ids equal mod 8 share a (head, run) group)
W = int(os.environ.get("SID_TUNE_W", "0"))
if W:
    QT = 2 if d <= 48 else 1
    NW = 8 if -(-(-(-N // 32)) // QT) >= 64 else 4
    nQG = -(-(-(-N // 32) // QT) // NW)
    groups = B * H * W
    ids = np.arange(len(t))
    grp = np.where(groups % 8 == 0, (ids & 7) + 8 * ((ids >> 3) // nQG), ids // nQG)
    run = grp % W
    life = (t[:, 5] - t[:, 0]) / 100
    for w in range(W):
        sel = run == w
        print("  run %d: %d workgroups, lifetime mean %.2f us (min %.2f max %.2f), last segment's steady loop mean %.2f us" % (
            w, sel.sum(), life[sel].mean(), life[sel].min(), life[sel].max(), ((t[sel, 3] - t[sel, 2]) / 100).mean()))
    xcd = ids & 7
    for x in range(8):
        sel = xcd == x
        print("  xcd %d: %d workgroups, lifetime mean %.2f us (min %.2f max %.2f), end of last %.2f us" % (
            x, sel.sum(), life[sel].mean(), life[sel].min(), life[sel].max(), (t[sel, 5].max() - t0) / 100))
    if os.environ.get("SID_DIAG_DUMP"):
        sel = np.where(xcd == 0)[0]
        print("  xcd 0 workgroups in id order: (start us, lifetime us)")
        print("   " + " ".join("%.0f/%.0f" % ((t[i, 0] - t0) / 100, life[i]) for i in sel))
    # segment by segment, for the runs that have more than one
    for w in range(W):
        sel = run == w
        if nseg[sel].max() < 2:
            continue
        r = raw[sel]
        for sg in range(int(nseg[sel].max())):
            d = [(r[:, sg, i + 1] - r[:, sg, i]).mean() / 100 for i in range(5)]
            print("  run %d segment %d: load %.2f  pre-pass %.2f  steady %.2f  tails %.2f  store %.2f us" % (w, sg, *d))
