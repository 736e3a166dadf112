"""GPU box: how many masks survive the AMG filters with the seeded Hiera-L weights for a few cfgAMG settings, and what a slice costs
then (the benchmark's default thresholds leave ~0.5 masks per slice, so the post-filter tail - NMS, compaction, pair intersections,
dedup, paint - runs on nothing; bench.py --tail picks a setting from this sweep)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd.engine import Engine, make_amg_params
from saber_amd.model_config import get_config
from saber_amd.weights import seeded_weights
from saber_amd.segmenters.slice_driver import segment_slice_to_plane
from oracle import saber_ref
eng = Engine("large", device=0, weights=seeded_weights(get_config("large"), 0), max_images=21, max_prompts=1024)
sl = [torch.from_numpy(saber_ref.synthetic_slice(seed=i)).cuda() for i in range(2)]
for amg in [dict(), dict(pred_iou_thresh=0.5, stability_score_thresh=0.8), dict(pred_iou_thresh=0.5, stability_score_thresh=0.8, box_nms_thresh=0.9),
            dict(pred_iou_thresh=0.5, stability_score_thresh=0.8, box_nms_thresh=0.95), dict(pred_iou_thresh=0.3, stability_score_thresh=0.6, box_nms_thresh=0.95),
            dict(pred_iou_thresh=0.5, stability_score_thresh=0.85, box_nms_thresh=0.98)]:
    a = dict(npoints=32, crop_n_layers=2); a.update(amg)
    prm = make_amg_params(a)
    for i in range(2):
        img = eng.prepare(sl[i])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bits, meta = eng.amg_generate(img, prm, max_masks=4096)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        plane, n = segment_slice_to_plane(eng, sl[i], prm, min_mask_area=50, max_masks=4096)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(amg, "slice", i, "amg masks", len(meta), f"amg {1e3*(t1-t0):.1f} ms; full slice -> plane {1e3*(t2-t1):.1f} ms, painted {n}", flush=True)
