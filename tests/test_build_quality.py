"""What the compiler made of the kernels in the built library, read from the code objects' metadata (no GPU):
a kernel that starts to spill vector registers keeps passing every parity test and loses a fifth of its speed
(a shared code path in the block-skipping probe once cost the three-block kernel 143 spilled registers)."""
import os
import re
import shutil
import subprocess

import pytest

from style_transfer_amd import lib

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_metadata(tmp_path):
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf) and os.path.exists(lib.LIB_PATH)):
        pytest.skip("needs the built library and the ROCm llvm tools")
    so = shutil.copy(lib.LIB_PATH, str(tmp_path / "libsid.so"))      # the bundles are extracted next to the file
    subprocess.run([objdump, "--offloading", so], check=True, capture_output=True)
    meta = {}
    for f in sorted(os.listdir(str(tmp_path))):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.name:\s+(\S+)", line)
            if m and m.group(1).startswith("_Z"):
                name = m.group(1)
                meta[name] = {}
            m = re.match(r"\s*\.(vgpr_spill_count|vgpr_count|private_segment_fixed_size):\s+(\d+)", line)
            if m and name:
                meta[name][m.group(1)] = int(m.group(2))
    return meta


def test_hot_kernels_do_not_spill(tmp_path):
    meta = kernel_metadata(tmp_path)
    flash = {k: v for k, v in meta.items() if "flash_kernel" in k}
    small = {k: v for k, v in meta.items() if "small_attn" in k}
    assert len(flash) > 100 and len(small) > 20 and any("hf" in k for k in meta) and any("merge_kernel" in k for k in meta)
    spilled = {k: v["vgpr_spill_count"] for k, v in meta.items() if v.get("vgpr_spill_count", 0) > 0}
    # the widest head-dim class (145-160: ten k-steps, six value blocks) is the only one allowed to spill, and little
    wide = lambda k: "flash_kernelILi10ELi6E" in k or "small_attn_kernelILi10E" in k
    offenders = {k: n for k, n in spilled.items() if not wide(k)}
    assert not offenders, offenders
    assert all(n <= 48 for n in spilled.values()), spilled
    # the workload's two injected shapes (head_dim 40 and 80), every decomposition, flag and arithmetic
    for cls in ("flash_kernelILi3ELi2ELi2ELb0E", "flash_kernelILi5ELi3ELi2ELb0E"):
        hot = [k for k in flash if cls in k]
        assert len(hot) >= 8, cls
        assert all(flash[k].get("private_segment_fixed_size", 0) == 0 for k in hot), cls
