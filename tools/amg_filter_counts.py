"""How many masks cfgAMG's own filters (stability_score_thresh 0.92, box_nms_thresh 0.7, crop_nms_thresh 0.7) leave on the golden slice with the
seeded Hiera-L weights, as a function of pred_iou_thresh - in the engine's EXACT mode (= the fp32 oracle to 1e-6, tests/test_gpu_exact.py),
to choose the parameters of the default-filters golden (oracle/make_golden_amg.py FILTERS=1).  python tools/amg_filter_counts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from saber_amd.engine import Engine, make_amg_params
from oracle import saber_ref
eng = Engine("large", device=0, seed=0, max_images=21, max_prompts=1024, precision="exact")
img = eng.prepare(torch.from_numpy(saber_ref.synthetic_slice(seed=0)).cuda())
for stab, bn, cn in ((0.92, 0.7, 0.7), (0.92, 1.0, 1.0), (0.0, 0.7, 0.7), (0.85, 0.7, 0.7), (0.92, 0.9, 0.9)):
    for thr in (0.0, 0.7, 0.78, 0.8):
        amg = dict(npoints=32, crop_n_layers=2, pred_iou_thresh=thr, stability_score_thresh=stab, box_nms_thresh=bn, crop_nms_thresh=cn)
        bits, meta = eng.amg_generate(img, make_amg_params(amg), max_masks=16384)
        print(f"stability {stab} box_nms {bn} crop_nms {cn} pred_iou {thr}: {len(meta)} masks", flush=True)
eng.close()
