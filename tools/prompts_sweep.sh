#!/bin/bash
# same-box sweep of the decode batch (prompts per batched decoder pass): per-prompt image-token state is 2 MB, the Infinity Cache 256 MB
for w in ${WORKERS:-1 2}; do for p in ${PROMPTS:-48 96 192 384 1024}; do
  python bench.py --workers $w --max-prompts $p --steps 4 --warmup 1 --no-cpu-baseline --no-encoder-only 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes_ms_per_slice']; print('workers', $w, 'max_prompts', $p, round(d['value'],3), round(d['ms_per_step'],1), 't2i', k['decoder_t2i'], 'i2t', k['decoder_i2t'], 'up', k['decoder_upscale'], 'ew', k['elementwise'], 'gemm', k['gemm_bf16'])"
done; done
