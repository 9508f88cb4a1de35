# Build an experimental libsid variant into ab/: bash tools/build_ab.sh NAME -DSOMETHING ...  -> ab/NAME.so
# sid_attn.hip (host side) and sid_attn_bf16.hip (bf16 kernels) are recompiled with the extra flags; the other
# objects come from the in-tree build (run python -m style_transfer_amd.build first).
NAME=$1; shift
OBJ=style-transfer_amd/csrc/_obj
mkdir -p ab
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm"
/opt/rocm/bin/hipcc $FL "$@" -c style-transfer_amd/csrc/sid_attn_bf16.hip -o ab/$NAME.bf16.o &
/opt/rocm/bin/hipcc $FL "$@" -c style-transfer_amd/csrc/sid_attn.hip -o ab/$NAME.host.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$NAME.so $OBJ/sid_common.cpp.o ab/$NAME.host.o ab/$NAME.bf16.o $OBJ/sid_attn_x3.hip.o $OBJ/sid_attn_f16.hip.o $OBJ/sid_attn_small.hip.o $OBJ/sid_mask.hip.o $OBJ/sid_hf.hip.o $OBJ/sid_adain.hip.o || exit 1
rm -f ab/$NAME.bf16.o ab/$NAME.host.o
echo ab/$NAME.so
