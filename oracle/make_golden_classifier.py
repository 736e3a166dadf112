"""Generates tests/golden/saber_classifier_glue.npz by IMPORTING the reference's own functions (authoring container only) for the
torch / numpy glue of the classifier filter:
  crop_and_resize_adaptive          saber/classifier/datasets/RandMaskCrop.py
  Predictor.apply_crops / preprocess (called unbound on a stand-in `self`: they only read min_area, model.input_mode, device)
  SAM2Classifier.apply_mask_to_features (unbound: no state)
  convert_predictions_to_masks (instance and semantic branches)   saber/filters/masks.py

    python -m oracle.make_golden_classifier
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

from oracle.make_golden import _stub

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "saber_classifier_glue.npz")


def scene(seed=5, H=200, W=240):
    rng = np.random.default_rng(seed)
    img = rng.normal(0.4, 0.15, (H, W)).astype(np.float32)
    yy, xx = np.mgrid[:H, :W]
    masks = []
    for (cy, cx, ry, rx) in [(60, 70, 18, 25), (150, 180, 30, 12), (100, 120, 95, 115), (20, 20, 4, 3), (190, 230, 15, 15), (64, 76, 20, 22)]:
        masks.append((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1).astype(np.uint8))
    masks.append(np.zeros((H, W), np.uint8))                       # empty mask: whole image resized
    one = np.zeros((H, W), np.uint8); one[77, 13] = 1               # single pixel: bbox side max(1, 0)
    masks.append(one)
    return img, np.stack(masks)


def main():
    _stub()
    for name in ["monai.transforms", "saber.classifier.models.common"]:
        sys.modules.setdefault(name, MagicMock())
    from saber.classifier.datasets.RandMaskCrop import crop_and_resize_adaptive
    from saber.classifier.models.predictor import Predictor
    from saber.classifier.models.SAM2 import SAM2Classifier
    from saber.filters import masks as fm
    img, masks = scene()
    G = {"image": img, "masks": masks}
    ti, tm = torch.from_numpy(img), torch.from_numpy(masks)
    ic, mc = crop_and_resize_adaptive(ti[None], tm[1])
    G["crop1_image"], G["crop1_mask"] = ic.numpy(), mc.numpy()
    fake = types.SimpleNamespace(min_area=250, model=types.SimpleNamespace(input_mode="separate"), device="cpu")
    ci, cm = Predictor.apply_crops(fake, ti, tm)
    G["crops_invalid_image"], G["crops_mask_area"] = ci.numpy()[7], (cm.numpy() > 0).sum(axis=(1, 2))     # (the valid ones are in "batch")
    batch, valid = Predictor.preprocess(fake, ci, cm)
    G["batch"], G["valid"] = batch.numpy(), np.array(valid)
    rng = np.random.default_rng(9)
    feats = torch.from_numpy(rng.normal(0, 1, (len(valid), 2, 64, 64)).astype(np.float32))
    G["feats"] = feats.numpy()
    G["masked_feats"] = SAM2Classifier.apply_mask_to_features(None, feats, batch[:, 1:2]).numpy()
    # resolution of predictions into masks
    dicts = [{"segmentation": m.astype(bool), "area": int(m.sum())} for m in masks[:6]]
    pred = np.array([[0.1, 0.8, 0.1], [0.2, 0.7, 0.1], [0.6, 0.3, 0.1], [0.1, 0.2, 0.7], [0.05, 0.9, 0.05], [0.3, 0.6, 0.1]], dtype=np.float32)
    G["pred"] = pred
    inst = fm.convert_predictions_to_masks(pred, list(dicts), 1, 32)
    G["inst_n"] = np.array(len(inst))
    G["inst_seg"] = np.stack([m["segmentation"] for m in inst]).astype(np.uint8)
    G["inst_area"] = np.array([m["area"] for m in inst])
    G["inst_bbox"] = np.array([m["bbox"] for m in inst])
    G["inst_conf"] = np.array([m["predicted_iou"] for m in inst], dtype=np.float64)
    sem = fm.convert_predictions_to_masks(pred, list(dicts), 0, 32)
    G["sem_seg"] = np.stack([np.asarray(m["segmentation"]) for m in sem]).astype(np.uint8)
    G["sem_area"] = np.array([m["area"] for m in sem])
    G["sem_label"] = np.array([m["label"] for m in sem])
    np.savez_compressed(OUT, **G)
    print("wrote", OUT, {k: v.shape for k, v in G.items()})


if __name__ == "__main__":
    main()
