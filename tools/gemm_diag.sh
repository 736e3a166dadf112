#!/bin/bash
echo "== bf16 out"; M0=21 python tools/gemm_bench.py 2>&1 | grep -E "M=  86016|M= 344064|M=1376256|M=  21504"
echo "== fp32 + residual"; RES=1 M0=21 python tools/gemm_bench.py 2>&1 | grep -E "M=  86016|M= 344064|M=1376256|M=  21504"
