#!/bin/bash
# Compile-time ablations of dec_upscale_kernel (csrc/decoder_fused.hip: UP_ABL): one library variant per mask, built here (CPU), timed on the
# GPU box by tools/upscale_ablate.py with SABER_AMD_LIB=<variant>.   bash tools/upscale_ablate.sh build | run
set -e
cd "$(dirname "$0")/.."
MASKS="1 16 32 64 8 384 385 465"     # GELU | hyper | X loads | LN reductions | stores | MFMAs | MFMAs+GELU | all compute (GELU+hyper+LN+MFMA)
CS=saber_amd/csrc
if [ "$1" = build ]; then
  make -C $CS -j8 > /dev/null
  for m in $MASKS; do
    mkdir -p $CS/build_abl
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -DUP_ABL=$m -DSABER_OP_NS=op_bf16 -DSABER_OP_SRC='"decoder_fused.hip"' \
        -I$CS -c $CS/op_wrap.hip -o $CS/build_abl/decoder_fused_$m.o &
  done
  wait
  for m in $MASKS; do
    OBJS=$(ls $CS/build/*.o $CS/build/f16/*.o $CS/build/bf16/*.o | grep -v "build/bf16/decoder_fused.o")
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $CS/build_abl/decoder_fused_$m.o -o saber_amd/libsaber_amd_abl$m.so
  done
  ls -la saber_amd/libsaber_amd_abl*.so
else
  for m in 0 $MASKS; do
    lib=saber_amd/libsaber_amd_abl$m.so; [ $m = 0 ] && lib=saber_amd/libsaber_amd.so
    echo "== UP_ABL=$m"; SABER_AMD_LIB=$lib UP_ONLY_FULL=1 python tools/upscale_ablate.py
  done
fi
