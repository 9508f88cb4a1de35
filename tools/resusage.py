"""Print per-kernel register/LDS/scratch usage of a HIP source (gfx950)."""
import re, subprocess, sys
src = sys.argv[1]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form", "-c", src,
                    "-o", "/tmp/_ru.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = t.split(":")[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
    print(f"{name:70s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} sgpr={v.get('SGPRs')} spillV={v.get('VGPRs Spill')} scratch={v.get('ScratchSize [bytes/lane]')} occ={v.get('Occupancy [waves/SIMD]')} lds={v.get('LDS Size [bytes/block]')}")
