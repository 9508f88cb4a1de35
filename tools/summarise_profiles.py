"""Turn gpurun_out/<tag> (tools/collect_profiles.sh) into the committed summaries under profiles/."""
import collections, csv, glob, io, json, os, re, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles")      # (on the GPU box: a directory under gpurun_out/)
os.makedirs(dst, exist_ok=True)

def kstats(path, skip=None):
    return subprocess.run([sys.executable, os.path.join(root, "tools", "kstats.py"), path] + ([skip] if skip else []),
                          capture_output=True, text=True).stdout

with open(os.path.join(dst, tag + "_bench_kernel_stats.txt"), "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-block-skip --no-content-cache --no-feature-scale --no-sweep --no-default-queues\n")
    f.write("(4 images: 1 warm-up + 3 timed; the torch normal_ kernels that create the synthetic workload are left out)\n\n")
    f.write(kstats(os.path.join(src, "bench", "bench_kernel_stats.csv"), "at::native"))
    f.write("\nbench line printed under the profiler:\n" + open(os.path.join(src, "bench_under_rocprof.json")).read())
with open(os.path.join(dst, tag + "_attn4096_kernel_stats.txt"), "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python3 tools/prof_attn.py 4096 40 1 20   (injected attention, N=4096 d=40 H=8 B=1)\n\n")
    f.write(kstats(os.path.join(src, "attn", "attn_kernel_stats.csv"), "at::native"))
    f.write("\n" + open(os.path.join(src, "attn.txt")).read())

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(src + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines, flash = [], None
for name, cs in acc.items():
    if "at::" in name:
        continue
    short = re.sub(r"\(anonymous namespace\)::", "", name)[:90]
    lines.append(short)
    for c, v in sorted(cs.items()):
        lines.append("   %-30s n=%-3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
    if "flash_kernel" in name:
        flash = {c: sum(v) / len(v) for c, v in cs.items()}
with open(os.path.join(dst, tag + "_attn4096_pmc.txt"), "w") as f:
    f.write("rocprofv3 --pmc <one pass per run> -- python3 tools/prof_attn.py 4096 40 1 3; mean per dispatch\n")
    f.write("FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (uncorrected)\n\n" + "\n".join(lines) + "\n")
if flash and "SQ_VALU_MFMA_BUSY_CYCLES" in flash and "GRBM_GUI_ACTIVE" in flash:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip -> SIMD-cycles of the dispatch
    simd_cycles = flash["GRBM_GUI_ACTIVE"] / 8 * 1024
    util = flash["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
    with open(os.path.join(dst, tag + "_attn4096_pmc.txt"), "a") as f:
        f.write("\nflash kernel: matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) = %.3f\n" % util)
        f.write("              VALU issue busy  = 4*SQ_ACTIVE_INST_VALU / SIMD-cycles = %.3f\n" % (4 * flash.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles))
    print("mfma busy", util)
if flash and "FETCH_SIZE" in flash:
    # gfx950: FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes -> double it;
    # WRITE_SIZE is exact for 16-byte-per-lane stores.  Both are reported in KiB.
    js = {"kernel": "flash_kernel, injected attention N=4096 d=40 H=8 B=1",
          "fetch_bytes": flash["FETCH_SIZE"] * 1024 * 2, "write_bytes": flash.get("WRITE_SIZE", 0.0) * 1024,
          "correction": "FETCH_SIZE x2 (gfx950 counts 128 B read requests as 64 B), WRITE_SIZE x1; KiB -> bytes"}
    js["traffic_bytes"] = js["fetch_bytes"] + js["write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in flash and "GRBM_GUI_ACTIVE" in flash:
        # hardware-counter utilisation of the matrix pipe: busy cycles over the SIMD-cycles of the dispatch
        js["mfma_busy"] = flash["SQ_VALU_MFMA_BUSY_CYCLES"] / (flash["GRBM_GUI_ACTIVE"] / 8 * 1024)
        js["valu_issue_busy"] = 4 * flash.get("SQ_ACTIVE_INST_VALU", 0) / (flash["GRBM_GUI_ACTIVE"] / 8 * 1024)
    json.dump(js, open(os.path.join(dst, tag + "_flash_traffic.json"), "w"), indent=1)
    print(js)
print("\n".join(lines[:60]))
# microbenchmarks and diagnostic-build runs (tools/collect_micro.sh)
import shutil
for name in ("micro_issue.txt", "micro_slot.txt", "micro_diag.txt"):
    if os.path.isfile(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, tag + "_" + name))
# matrix-pipe occupancy in shader cycles: GRBM_GUI_ACTIVE does not tick at the shader clock, so take
# the kernel's duration from the kernel trace and the shader clock the diagnostic build measured
try:
    dur = None
    for r in csv.DictReader(open(os.path.join(src, "attn", "attn_kernel_stats.csv"))):
        if "flash_kernel" in r["Name"]:
            dur = float(r["AverageNs"]) * 1e-9
    m = re.search(r"shader clock in the steady loop: ([0-9.]+) GHz", open(os.path.join(src, "micro_diag.txt")).read())
    if dur and m and flash:
        ghz = float(m.group(1))
        busy = flash["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (dur * ghz * 1e9)
        with open(os.path.join(dst, tag + "_attn4096_pmc.txt"), "a") as f:
            f.write("              in shader cycles: kernel %.1f us (kernel trace) x %.2f GHz (diagnostic build, steady loop) = %.0f cycles;\n"
                    "              matrix-pipe busy = %.3f, i.e. %.0f MFMA cycles per SIMD (32 per 32x32x16 MFMA)\n"
                    % (dur * 1e6, ghz, dur * ghz * 1e9, busy, flash["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024))
        print("mfma busy (shader cycles)", busy)
except Exception as e:   # the summaries above do not depend on this
    print("no shader-cycle occupancy:", e)
