# Build an experimental libsid variant: bash tools/build_ab.sh NAME -DSOMETHING ...  -> tools/_bin/NAME.so
# sid_attn.hip (host side) and the bf16 / f16 kernel units are recompiled with the extra flags; the other
# objects come from the in-tree build (run python -m style_transfer_amd.build first).  tools/_bin/ is
# git-ignored but travels to the GPU box.
NAME=$1; shift
OBJ=style-transfer_amd/csrc/_obj
mkdir -p tools/_bin
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm"
/opt/rocm/bin/hipcc $FL "$@" -c style-transfer_amd/csrc/sid_attn_bf16.hip -o tools/_bin/$NAME.bf16.o &
/opt/rocm/bin/hipcc $FL "$@" -c style-transfer_amd/csrc/sid_attn.hip -o tools/_bin/$NAME.host.o &
/opt/rocm/bin/hipcc $FL "$@" -c style-transfer_amd/csrc/sid_attn_f16.hip -o tools/_bin/$NAME.f16.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/$NAME.so $OBJ/sid_common.cpp.o tools/_bin/$NAME.host.o tools/_bin/$NAME.bf16.o $OBJ/sid_attn_x3.hip.o tools/_bin/$NAME.f16.o $OBJ/sid_attn_small.hip.o $OBJ/sid_mask.hip.o $OBJ/sid_hf.hip.o $OBJ/sid_adain.hip.o || exit 1
rm -f tools/_bin/$NAME.bf16.o tools/_bin/$NAME.host.o tools/_bin/$NAME.f16.o
echo tools/_bin/$NAME.so
