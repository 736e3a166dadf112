"""How many masks the mask generator's filters leave on the golden slice with the seeded Hiera-L weights - in the engine's EXACT mode (= the fp32
oracle to 1e-6, tests/test_gpu_exact.py) - to choose the parameters of the filters golden (oracle/make_golden_amg.py FILTERS=1).
cfgAMG's own thresholds (stability 0.92, both NMS 0.7) leave 0-1 masks whatever pred_iou_thresh is (the seeded decoder's masks are unstable,
crop-sized blobs), so the golden uses the thresholds at which each filter removes a real share of the candidates.  python tools/amg_filter_counts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from saber_amd.engine import Engine, make_amg_params
from oracle import saber_ref
eng = Engine("large", device=0, seed=0, max_images=21, max_prompts=1024, precision="exact")
img = eng.prepare(torch.from_numpy(saber_ref.synthetic_slice(seed=0)).cuda())
base = dict(npoints=32, crop_n_layers=2)
_, meta = eng.amg_generate(img, make_amg_params(dict(base, pred_iou_thresh=0.0, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0)), max_masks=16384)
st = np.array([m.stability_score for m in meta]); pi = np.array([m.predicted_iou for m in meta])
print(len(meta), "candidates with every filter off; stability quantiles 10/25/50/75/90/99 %:", np.round(np.quantile(st, [.1, .25, .5, .75, .9, .99]), 4),
      "; predicted IoU quantiles:", np.round(np.quantile(pi, [.1, .25, .5, .75, .9, .99]), 4))
for stab_q in (0.5, 0.75):
    stab = float(np.quantile(st, stab_q))
    for thr_q in (0.0, 0.5):
        thr = float(np.quantile(pi, thr_q)) if thr_q > 0 else 0.0
        for bn, cn in ((1.0, 1.0), (0.98, 0.98), (0.95, 0.95), (0.9, 0.9), (0.8, 0.8), (0.7, 0.7), (0.95, 0.7), (0.7, 0.95)):
            amg = dict(base, pred_iou_thresh=thr, stability_score_thresh=stab, box_nms_thresh=bn, crop_nms_thresh=cn)
            _, m2 = eng.amg_generate(img, make_amg_params(amg), max_masks=16384)
            print(f"stability >= {stab:.4f} (q{stab_q}) pred_iou > {thr:.4f} box_nms {bn} crop_nms {cn}: {len(m2)} masks", flush=True)
eng.close()
