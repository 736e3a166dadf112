#!/bin/bash
# SQ counters under dec_upscale_kernel: one pass of 8 SQ counters over a decode-only run (tools/decode_bench.py), summarised per kernel.
# bash tools/upscale_pmc.sh <tag>      (GPU box; rocprofv3 gets the python program itself after --)
set -u
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o dec -- python3 tools/decode_bench.py 1024 > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o dec -- python3 tools/decode_bench.py 1024 > $OUT/pmc_sq2.log 2>&1
python3 tools/summarize_sq.py $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/sq_summary.txt
cat $OUT/sq_summary.txt
