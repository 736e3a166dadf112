#!/bin/bash
# same-box A/B of a GEMM debug flag on the Hiera shapes (M0=21: the 21-crop batch of one slice): bash tools/gemm_diag.sh
# DBG bit 2 (value 2): gemm_bf16_glds2_kernel computes the wave columns beyond N too (the pre-"cols_live" behaviour)
for d in 0 2; do
  echo "== fp32 + residual (proj / fc2), DBG=$d"; DBG=$d RES=1 M0=21 python tools/gemm_bench.py 2>&1 | grep -E "N= *(576|288|144|1152) "
done
