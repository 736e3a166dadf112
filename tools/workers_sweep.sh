#!/bin/bash
# same-box sweep of engine handles per GPU (slices in flight): bash tools/workers_sweep.sh
for w in 2 3 4; do
  python bench.py --workers $w --steps 12 --warmup 3 --no-cpu-baseline --no-profile --no-encoder-only 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('workers', d['config']['engine_handles_per_gpu'], d['value'], d['ms_per_step'])"
done
