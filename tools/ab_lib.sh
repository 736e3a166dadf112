#!/bin/bash
# same-box A/B of two builds of the library: bash tools/ab_lib.sh saber_amd/libsaber_amd_variant.so
V=${1:?path of the variant library}
for lib in "" "$V" "" "$V"; do
  SABER_AMD_LIB=$lib python bench.py --workers 1 --steps 4 --warmup 1 --no-cpu-baseline --no-encoder-only 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes_ms_per_slice']; print('lib', '${lib:-default}', round(d['value'],3), round(d['ms_per_step'],1), 'i2t', k['decoder_i2t'], 't2i', k['decoder_t2i'], 'up', k['decoder_upscale'], 'gemm', k['gemm_bf16'], 'att', k['hiera_attention'])"
done
