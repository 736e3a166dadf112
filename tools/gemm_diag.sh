#!/bin/bash
for d in 0 64 66; do echo "== DBG=$d"; DBG=$d M0=21 python tools/gemm_bench.py 2>&1 | grep -E "N= 1728|N= 2304|N= 3456|N= 4608|M=   4096|M=   8192" ; echo "-- gelu"; ACT=1 DBG=$d M0=21 python tools/gemm_bench.py 2>&1 | grep -E "N= 2304|N= 4608"; done
