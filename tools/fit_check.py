import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from saber_amd.engine import Engine, make_amg_params
from saber_amd.model_config import get_config
from saber_amd.weights import fitted_decoder_weights
from oracle import saber_ref
cfg = get_config("large"); W = fitted_decoder_weights(cfg, 0)
for prec in ("exact", "fp16", "bf16"):
    eng = Engine("large", device=0, weights=W, max_images=21, max_prompts=1024, precision=prec)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for seed in (0, 7):
            img = eng.prepare(torch.from_numpy(saber_ref.synthetic_slice(seed=seed)).cuda())
            bits, meta = eng.amg_generate(img, make_amg_params(dict(npoints=32, crop_n_layers=2)), max_masks=4096)
            a = np.array([m.area for m in meta]); pi = np.array([m.predicted_iou for m in meta]); ss = np.array([m.stability_score for m in meta])
            print(prec, "seed", seed, "masks", len(meta), "areas", np.sort(a)[:5], "...", np.sort(a)[-5:], "pred_iou %.2f..%.2f" % (pi.min() if len(pi) else 0, pi.max() if len(pi) else 0), "stab %.3f..%.3f" % (ss.min() if len(ss) else 0, ss.max() if len(ss) else 0), flush=True)
            bits2, meta2 = eng.amg_generate(img, make_amg_params(dict(npoints=32, crop_n_layers=2, pred_iou_thresh=0.0, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0)), max_masks=16384)
            pi = np.array([m.predicted_iou for m in meta2]); ss = np.array([m.stability_score for m in meta2])
            print("   all candidates", len(meta2), "pred_iou quantiles", np.quantile(pi, [0.1, 0.5, 0.9]).round(3), "stability quantiles", np.quantile(ss, [0.1, 0.5, 0.9]).round(3), "pass both", int(((pi > 0.7) & (ss >= 0.92)).sum()), flush=True)
    eng.close()
