#!/bin/bash
# same-box A/B of one saber_k_set_debug bit on the whole step: bash tools/ab_flag.sh <bit-value> [class]
BIT=${1:-8}
for d in 0 $BIT 0 $BIT; do
  SABER_AMD_DEBUG=$d python bench.py --workers 1 --steps 4 --warmup 1 --no-cpu-baseline --no-encoder-only 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug', $d, round(d['value'],3), round(d['ms_per_step'],1), d['kernel_classes_ms_per_slice'])"
done
