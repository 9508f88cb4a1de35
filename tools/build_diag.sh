# Diagnostic build of libsid (per-workgroup clock stamps in the flash kernel, read by tools/diag_flash.py):
#   bash tools/build_diag.sh   -> exp/diag.so
# Every unit that sees FlashParams is compiled with -DSID_DIAG; the rest comes from the in-tree build.
set -e
mkdir -p exp
OBJ=style-transfer_amd/csrc/_obj
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm -DSID_DIAG"
for u in sid_attn sid_attn_bf16 sid_attn_x3 sid_attn_f16; do
  /opt/rocm/bin/hipcc $F -c style-transfer_amd/csrc/$u.hip -o exp/d_$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/diag.so $OBJ/sid_common.cpp.o exp/d_sid_attn.o exp/d_sid_attn_bf16.o exp/d_sid_attn_x3.o exp/d_sid_attn_f16.o \
  $OBJ/sid_attn_small.hip.o $OBJ/sid_mask.hip.o $OBJ/sid_hf.hip.o $OBJ/sid_adain.hip.o
rm exp/d_*.o
echo exp/diag.so
