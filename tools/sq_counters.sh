#!/bin/bash
# SQ counters (two passes of 8) under any of the repo's micro-benchmarks, summarised per kernel by tools/summarize_sq.py:
#   bash tools/sq_counters.sh <tag> tools/decode_bench.py 1024        (GPU box; rocprofv3 gets the python program itself after --)
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o run -- python3 "$@" > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o run -- python3 "$@" > $OUT/pmc_sq2.log 2>&1
python3 tools/summarize_sq.py $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/sq_summary.txt
cat $OUT/sq_summary.txt
