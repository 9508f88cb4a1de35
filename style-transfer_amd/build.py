"""Build libsid.so (gfx950) in-tree with hipcc.  Usage: python -m style_transfer_amd.build [--force]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsid.so")
SOURCES = ["sid_common.cpp", "sid_attn.hip", "sid_attn_bf16.hip", "sid_attn_x3.hip", "sid_attn_f16.hip", "sid_attn_small.hip", "sid_mask.hip", "sid_hf.hip", "sid_adain.hip"]
HEADERS = [os.path.join(CSRC, "sid_common.h"), os.path.join(CSRC, "sid_attn_kernel.hpp"), os.path.join(ROOT, "include", "sid.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in the VGPRs the VALU softmax reads and rescales;
# without it hipcc parks them in AGPRs and copies 16 registers in and out around every MFMA.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wall", "-Wno-unused-function",
         "-Wno-inline-asm"]      # the stage copy declares m0 clobbered on purpose (see glds16)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src + ".o")
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
