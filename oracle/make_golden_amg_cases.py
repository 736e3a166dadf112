"""Test infrastructure (not product): the oracle's AMG results for the four large-trunk cases of tests/test_gpu_engine.py and
tests/test_gpu_parity_bf16.py, computed ONCE here (minutes of CPU time each on the GPU box's host share) and committed as
tests/golden/amg_cases_large_seed0.npz.  The GPU tests compare the engine with these masks under the assertions they always had; the
inputs (seeded Hiera-L weights, the synthetic images below) are regenerated identically in the tests.

    fp32_l0      oracle/sam2_ref + oracle/amg_ref, npoints=6, crop_n_layers=0, box NMS off        (test_amg_parity[0-1.0])
    fp32_l1      the same with crop_n_layers=1, box_nms_thresh=0.95                               (test_amg_parity[1-0.95])
    fp32_ragged  600 x 840 image, crop_n_layers=0                                                 (test_amg_parity_non_square_image)
    emul_l0      oracle/sam2_bf16_emul (bf16-operand-emulating predictor), as fp32_l0             (test_amg_masks_vs_bf16_emulating_oracle)

Masks are stored at quarter resolution (the samples [2::4, 2::4] of every mask, bit-packed along x, as tests/golden/amg_default_grid_seed0.npz
does): 1/16 of the pixels estimate a mask pair's IoU to a few 1e-4, an order below the differences the tests bound.
Usage: python oracle/make_golden_amg_cases.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def image_1024():
    rng = np.random.default_rng(7)
    img = rng.uniform(0, 1, (1024, 1024)).astype(np.float32)
    yy, xx = np.mgrid[:1024, :1024]
    for _ in range(10):
        cy, cx = rng.integers(100, 924, 2)
        r = rng.integers(30, 120)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] *= 0.3
    return img


def image_ragged():
    rng = np.random.default_rng(21)
    H, W = 600, 840
    img = rng.uniform(0, 1, (H, W)).astype(np.float32)
    yy, xx = np.mgrid[:H, :W]
    for _ in range(8):
        cy, cx, r = rng.integers(60, H - 60), rng.integers(60, W - 60), rng.integers(25, 90)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] *= 0.3
    return img


def main():
    from oracle import sam2_bf16_emul as E
    from oracle import sam2_ref
    from oracle.amg_ref import amg_from_saber_cfg
    from oracle.sam2_ref import ImagePredictorRef
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    cfg = get_config("large")
    Wnp = seeded_weights(cfg, 0)
    Wt = sam2_ref.to_torch(Wnp)
    base = dict(npoints=6, box_nms_thresh=1.0, pred_iou_thresh=0.5, stability_score_thresh=0.8)
    cases = {
        "fp32_l0": (lambda: ImagePredictorRef(Wnp, cfg), dict(base, crop_n_layers=0), image_1024()),
        "fp32_l1": (lambda: ImagePredictorRef(Wnp, cfg), dict(base, crop_n_layers=1, box_nms_thresh=0.95), image_1024()),
        "fp32_ragged": (lambda: ImagePredictorRef(Wnp, cfg), dict(base, crop_n_layers=0), image_ragged()),
        "emul_l0": (lambda: E.ImagePredictorEmul(Wt, cfg), dict(base, crop_n_layers=0), image_1024()),
    }
    out = {}
    for name, (make, amg, img) in cases.items():
        t0 = time.time()
        ref = amg_from_saber_cfg(make(), amg).generate(np.repeat(img[..., None], 3, 2))
        seg = np.stack([r["segmentation"] for r in ref]).astype(bool)[:, 2::4, 2::4]
        out[name + "_bits"] = np.packbits(seg, axis=-1)
        out[name + "_width"] = np.array(seg.shape[-1])
        out[name + "_predicted_iou"] = np.array([r["predicted_iou"] for r in ref], dtype=np.float32)
        print(f"{name}: {len(ref)} masks in {time.time() - t0:.0f} s", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "amg_cases_large_seed0.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
