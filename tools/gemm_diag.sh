#!/bin/bash
for d in 0 0x200 0x400 0x600 0x800 0xc00; do echo "== DBG=$d"; DBG=$d M0=21 python tools/gemm_bench.py 2>&1 | grep -E "M=  86016" ; done
