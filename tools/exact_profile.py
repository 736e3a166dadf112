"""Time the exact (fp32-operand) mode on one cfgAMG-default slice (21 crops, 3 072 grid prompts + m2m): the workload of
tests/test_gpu_exact.py::test_exact_default_grid_amg_golden, without the oracle.  Run it under
`rocprofv3 --kernel-trace --stats -- python3 tools/exact_profile.py` for the per-kernel split (tools/trace_by_kernel.py)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from saber_amd.engine import Engine, make_amg_params
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    steps = int(os.environ.get("STEPS", "2"))
    W = seeded_weights(get_config("large"), 0)
    eng = Engine("large", device=0, weights=W, max_images=21, max_prompts=1024, precision="exact")
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 1, (1024, 1024)).astype(np.float32)
    yy, xx = np.mgrid[:1024, :1024]
    for _ in range(12):
        cy, cx = rng.integers(100, 924, 2)
        r = rng.integers(30, 120)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] *= 0.3
    x = eng.prepare(torch.from_numpy(img).cuda())
    amg = make_amg_params(dict(npoints=32, crop_n_layers=2, pred_iou_thresh=0.8055, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0))
    for it in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.time()
        bits, meta = eng.amg_generate(x, amg, max_masks=4096)
        torch.cuda.synchronize()
        print(f"exact default-grid AMG pass {it}: {time.time() - t0:.3f} s, {len(meta)} masks", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
