#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  bash tools/collect_profile.sh r01
# (--workers 1: with the default two slices in flight the kernels of the two streams overlap and stretch each other's durations;
#  bench.py's own roofline figures are measured on a single-handle step too)
# 1) kernel-trace + stats of bench.py, 2) two separate PMC passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 2 --warmup 1 --workers 1 --no-cpu-baseline --no-encoder-only --no-tail --no-video > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --steps 1 --warmup 0 --workers 1 --no-cpu-baseline --no-profile --no-video > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --steps 1 --warmup 0 --workers 1 --no-cpu-baseline --no-profile --no-video > $OUT/bench_write.log 2>&1
# MFMA utilisation (north star: "rocprof MFMA utilisation"): matrix-core busy cycles summed over the SIMDs, and the GPU-active cycles
# (summed over the 8 XCDs) of the same dispatches; SQ and GRBM counters share a pass
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o bench -- python3 bench.py --steps 1 --warmup 0 --workers 1 --no-cpu-baseline --no-profile --no-video > $OUT/bench_mfma.log 2>&1
python3 tools/summarize_pmc.py $OUT $TAG
ls -la $OUT
