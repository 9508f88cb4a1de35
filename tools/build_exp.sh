# Build an experimental libsid variant: bash tools/build_exp.sh NAME -DSOMETHING ...  -> exp/NAME.so
# Only sid_attn_bf16.hip is recompiled with the extra flags; the other objects come from the in-tree build.
NAME=$1; shift
OBJ=style-transfer_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm "$@" -c style-transfer_amd/csrc/sid_attn_bf16.hip -o exp/$NAME.bf16.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/$NAME.so $OBJ/sid_common.cpp.o $OBJ/sid_attn.hip.o exp/$NAME.bf16.o $OBJ/sid_attn_x3.hip.o $OBJ/sid_attn_small.hip.o $OBJ/sid_mask.hip.o $OBJ/sid_hf.hip.o $OBJ/sid_adain.hip.o
rm exp/$NAME.bf16.o
echo exp/$NAME.so
