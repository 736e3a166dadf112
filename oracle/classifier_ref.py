"""TEST INFRASTRUCTURE (CPU oracle) - restatement of the reference's domain-expert classifier filter, fp32 torch on
the CPU.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(saber_amd/classifier/) never does.

What is restated, and where it lives in the reference:
  normalize_intensity           monai.transforms.NormalizeIntensity() with its defaults, as constructed at
                                saber/classifier/models/predictor.py:60 (whole-image mean, population std, std==0 -> 1)
  crop_and_resize_adaptive      saber/classifier/datasets/RandMaskCrop.py:44-171 (apply_translation=False branch), resize_image :173-205
  apply_crops / preprocess      saber/classifier/models/predictor.py:221-249 / :62-115
  head / mask_features          saber/classifier/models/SAM2.py:58-92 (projection, classifier), :118-167 (forward), :169-197
  predict / batch_predict       saber/classifier/models/predictor.py:117-219
  convert_predictions_to_masks, _consensus_based_resolution, _semantic_segmentation, apply_classifier
                                saber/filters/masks.py:8-62, :64-122, :124-158
Pinned by tests/golden/saber_classifier_glue.npz (made by oracle/make_golden_classifier.py from the reference's own functions for the
torch/numpy glue: crops, preprocess, mask features, consensus / semantic resolution).  The head's layer list is restated from the source
(the reference module cannot be constructed offline: its __init__ downloads the SAM2 checkpoint), under the state_dict names the
reference's nn.Sequential indices give, so a trained checkpoint of the reference loads unchanged."""
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F
from scipy import ndimage

from oracle import sam2_ref

BN_EPS = 1e-5
LN_EPS = 1e-5


def normalize_intensity(img: torch.Tensor) -> torch.Tensor:
    img = img.float()
    mean = img.mean()
    std = img.std(unbiased=False)
    if float(std) == 0.0:
        return img - mean
    return (img - mean) / std


def _resize_gray(image: torch.Tensor, size) -> torch.Tensor:
    """image [1,h,w] -> [1,H,W], bilinear, align_corners=False, no antialias (resize_image's grayscale branch)."""
    return F.interpolate(image.unsqueeze(0), size=size, mode="bilinear", align_corners=False).squeeze(0)


def crop_box(mask: torch.Tensor, margin: float = 1.5, full_mask_thresh: float = 0.9):
    """The (top, left, h, w) window crop_and_resize_adaptive cuts for this [H,W] mask, or None when it resizes the whole image (empty mask,
    or a bounding box covering >= 90 % of both sides)."""
    H, W = mask.shape
    nz = torch.nonzero(mask)
    if nz.numel() == 0:
        return None
    y0, y1 = int(nz[:, 0].min()), int(nz[:, 0].max())
    x0, x1 = int(nz[:, 1].min()), int(nz[:, 1].max())
    bh, bw = max(1, y1 - y0), max(1, x1 - x0)
    if bh / H >= full_mask_thresh and bw / W >= full_mask_thresh:
        return None
    ch, cw = int(bh * (1 + margin)), int(bw * (1 + margin))
    top = (y0 + y1) // 2 - ch // 2
    left = (x0 + x1) // 2 - cw // 2
    top = max(0, min(top, H - ch))
    left = max(0, min(left, W - cw))
    return top, left, min(ch, H), min(cw, W)


def crop_and_resize_adaptive(image: torch.Tensor, mask: torch.Tensor, margin: float = 1.5, output_size=(320, 320), full_mask_thresh: float = 0.9):
    """image [1,H,W] float, mask [H,W] -> (image [1,oh,ow] float, mask [1,oh,ow] of the mask's dtype)."""
    if mask.dim() == 2:
        mask = mask.unsqueeze(0)
    box = crop_box(mask[0], margin, full_mask_thresh)
    if box is not None:
        t, l, h, w = box
        image, mask = image[:, t:t + h, l:l + w], mask[:, t:t + h, l:l + w]
    return _resize_gray(image, output_size), F.interpolate(mask.unsqueeze(0), size=output_size, mode="nearest").squeeze(0)


def apply_crops(image: torch.Tensor, masks: torch.Tensor):
    image0 = image.unsqueeze(0) if image.ndim == 2 else image
    out = []
    for m in masks:
        ic, mc = crop_and_resize_adaptive(image0, m)
        out.append(torch.cat([ic, mc.to(image.dtype)], dim=0))
    out = torch.stack(out)
    return out[:, 0], out[:, 1]


def preprocess(image: torch.Tensor, masks: torch.Tensor, min_area: int = 250):
    """-> ((n_valid, 2, H, W) [image, mask] batch or None, valid indices)"""
    if image.ndim == 2:
        image = image.unsqueeze(0)
    elif image.ndim == 3 and image.shape[0] == masks.shape[0]:
        image = image.unsqueeze(1)
    binm = (masks > 0).to(torch.uint8)
    areas = binm.sum(dim=[1, 2])
    valid = (areas >= min_area).nonzero(as_tuple=False).squeeze(1).tolist()
    binm = binm[valid]
    if binm.shape[0] == 0:
        return None, []
    binm = binm.unsqueeze(1)
    im = image.expand(binm.shape[0], -1, -1, -1) if image.shape[0] == 1 else image[valid]
    return torch.cat([im, binm], dim=1), valid


def mask_features(feature_map: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    m = F.interpolate(mask.to(feature_map.dtype), size=feature_map.shape[2:], mode="nearest")
    return torch.cat([feature_map * m, feature_map * (1 - m)], dim=1)


def head_param_specs(num_classes: int, hidden: int = 256):
    """state_dict entries of SAM2Classifier (the SAM2 backbone is not a registered sub-module, so these are all of them)."""
    d0, d1 = hidden, hidden // 2
    specs = {}

    def conv(i, cout, cin, k):
        specs[f"projection.{i}.weight"] = (cout, cin, k, k)
        specs[f"projection.{i}.bias"] = (cout,)

    def bn(i, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            specs[f"projection.{i}.{n}"] = (c,)

    conv(0, d0, 512, 1); bn(1, d0); specs["projection.2.weight"] = (1,)
    conv(4, d0, d0, 3); bn(5, d0); specs["projection.6.weight"] = (1,)
    conv(9, d1, d0, 3); bn(10, d1); specs["projection.11.weight"] = (1,)
    specs["classifier.0.weight"] = (64, d1); specs["classifier.0.bias"] = (64,)
    specs["classifier.1.weight"] = (64,); specs["classifier.1.bias"] = (64,)
    specs["classifier.2.weight"] = (1,)
    specs["classifier.4.weight"] = (num_classes, 64); specs["classifier.4.bias"] = (num_classes,)
    return specs


def seeded_head(num_classes: int, seed: int = 0, hidden: int = 256) -> Dict[str, np.ndarray]:
    """Synthetic head weights (no trained checkpoint offline): fan-in scaled normals, BatchNorm running statistics away from (0, 1) and
    PReLU slopes away from the 0.25 default so that every term of the folded arithmetic is exercised."""
    rng = np.random.default_rng(1000 + seed)
    W = {}
    for name, shape in head_param_specs(num_classes, hidden).items():
        if name.endswith("running_var"):
            a = rng.uniform(0.5, 2.0, shape)
        elif name.endswith("running_mean"):
            a = rng.normal(0, 0.3, shape)
        elif len(shape) == 1 and shape[0] == 1:
            a = rng.uniform(0.1, 0.4, shape)
        elif ".1." in name or ".5." in name or ".10." in name:          # BN / LN affine
            a = rng.uniform(0.7, 1.3, shape) if name.endswith("weight") else rng.normal(0, 0.1, shape)
        elif name.endswith("bias"):
            a = rng.normal(0, 0.05, shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = rng.normal(0, 1.6 / np.sqrt(fan_in), shape)
        W[name] = a.astype(np.float32)
    return W


def head(Wh: Dict[str, torch.Tensor], feats512: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """projection (eval mode: BatchNorm on running statistics, dropouts off) -> global average pool -> classifier.  -> logits (B, classes)"""
    def block(x, ic, ib, ip, pad):
        x = F.conv2d(x, Wh[f"projection.{ic}.weight"], Wh[f"projection.{ic}.bias"], padding=pad)
        x = F.batch_norm(x, Wh[f"projection.{ib}.running_mean"], Wh[f"projection.{ib}.running_var"], Wh[f"projection.{ib}.weight"],
                         Wh[f"projection.{ib}.bias"], training=False, eps=BN_EPS)
        return F.prelu(x, Wh[f"projection.{ip}.weight"])
    x = block(feats512, 0, 1, 2, 0)
    if taps is not None: taps["p0"] = x
    x = F.max_pool2d(block(x, 4, 5, 6, 1), 2, 2)
    if taps is not None: taps["p1"] = x
    x = F.max_pool2d(block(x, 9, 10, 11, 1), 2, 2)
    if taps is not None: taps["p2"] = x
    v = F.adaptive_avg_pool2d(x, (1, 1)).view(x.size(0), -1)
    if taps is not None: taps["pooled"] = v
    h = F.linear(v, Wh["classifier.0.weight"], Wh["classifier.0.bias"])
    h = F.layer_norm(h, (64,), Wh["classifier.1.weight"], Wh["classifier.1.bias"], LN_EPS)
    h = F.prelu(h, Wh["classifier.2.weight"])
    return F.linear(h, Wh["classifier.4.weight"], Wh["classifier.4.bias"])


class PredictorRef:
    """saber.classifier.models.predictor.Predictor with the SAM2 backbone and the head restated (fp32, CPU)."""

    def __init__(self, sam_weights: Dict[str, np.ndarray], cfg, head_weights: Dict[str, np.ndarray], num_classes: int, min_area: int = 250):
        self.W = sam2_ref.to_torch(sam_weights)
        self.cfg = cfg
        self.Wh = {k: torch.from_numpy(np.asarray(v)).float() for k, v in head_weights.items()}
        self.num_classes = num_classes
        self.min_area = min_area
        self.taps = None

    @torch.no_grad()
    def image_embed(self, gray: torch.Tensor) -> torch.Tensor:
        """SAM2Classifier.forward's backbone step for one [H,W] crop: grey -> 3 channels -> SAM2Transforms -> image_embed [1,256,64,64]"""
        rgb = np.repeat(gray.numpy()[..., None], 3, axis=2)
        return sam2_ref.encode_image(self.W, self.cfg, sam2_ref.sam2_transforms(rgb, self.cfg.image_size))["image_embed"]

    @torch.no_grad()
    def logits(self, crops: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
        """crops (B,1,H,W) float, masks (B,1,H,W) float 0/1 -> logits (B, classes)"""
        emb = torch.cat([self.image_embed(c[0]) for c in crops])
        f = mask_features(emb, masks)
        if self.taps is not None:
            self.taps["embed"], self.taps["feats512"] = emb, f
        return head(self.Wh, f, self.taps)

    @torch.no_grad()
    def predict(self, image, masks) -> np.ndarray:
        image = torch.as_tensor(np.asarray(image), dtype=torch.float32)
        masks = torch.as_tensor(np.asarray(masks), dtype=torch.uint8)
        n = masks.shape[0]
        image = normalize_intensity(image)
        ci, cm = apply_crops(image, masks)
        batch, valid = preprocess(ci, cm, self.min_area)
        full = np.zeros((n, self.num_classes), dtype=np.float32)
        if batch is None:
            return full
        probs = torch.softmax(self.logits(batch[:, 0:1], batch[:, 1:2]), dim=1).numpy()
        full[valid] = probs
        return full

    def batch_predict(self, image, masks, batch_size: int = 32) -> np.ndarray:
        masks = np.asarray(masks)
        out = np.zeros((masks.shape[0], self.num_classes), dtype=np.float32)
        for s in range(0, masks.shape[0], batch_size):
            out[s:s + batch_size] = self.predict(image, masks[s:s + batch_size])
        return out


# --------------------------------------------------------------------------------------------------- saber/filters/masks.py
def consensus_based_resolution(image_shape, masks: List[dict], confidences) -> List[dict]:
    h, w = image_shape
    conf_map = np.zeros((h, w), dtype=np.float32)
    count = np.zeros((h, w), dtype=np.int32)
    for m, c in zip(masks, confidences):
        conf_map += m["segmentation"] * c
        count += m["segmentation"]
    with np.errstate(divide="ignore", invalid="ignore"):
        avg = np.nan_to_num(np.divide(conf_map, count))
    lab, n = ndimage.label(count > 0)
    out = []
    for k in range(1, n + 1):
        comp = lab == k
        conf = np.mean(avg[comp])
        ys, xs = np.where(comp)
        y0, y1, x0, x1 = ys.min(), ys.max(), xs.min(), xs.max()
        out.append({"segmentation": comp, "area": int(comp.sum()), "bbox": [int(x0), int(y0), int(x1 - x0), int(y1 - y0)],
                    "predicted_iou": float(conf), "point_coords": [[int((x0 + x1) / 2), int((y0 + y1) / 2)]],
                    "stability_score": float(conf), "crop_box": [int(x0), int(y0), int(x1), int(y1)]})
    return out


def semantic_segmentation(masks: List[dict], predictions: np.ndarray) -> List[dict]:
    cls = np.argmax(predictions, axis=1)
    out = [{"segmentation": np.zeros(masks[0]["segmentation"].shape, dtype=np.uint8), "area": 0, "label": k} for k in range(1, predictions.shape[1])]
    for m, c in zip(masks, cls):
        if c > 0:
            out[c - 1]["segmentation"] = np.logical_or(out[c - 1]["segmentation"], m["segmentation"]).astype(bool)
            out[c - 1]["area"] += m["area"]
    return out


def convert_predictions_to_masks(predictions: np.ndarray, masks: List[dict], desired_class: Optional[int] = None, min_mask_area: int = 100):
    cls = np.argmax(predictions, axis=1)
    if desired_class > 0 and desired_class is not None:      # (the reference's own order of the two tests: None raises TypeError there too)
        conf = predictions[:, desired_class]
        idx = [i for i, c in enumerate(cls) if c == desired_class]
        masks = [masks[i] for i in idx]
        conf = conf[idx]
        if len(masks) > 0:
            masks = consensus_based_resolution(masks[0]["segmentation"].shape, masks, conf)
            masks = sorted([m for m in masks if m["area"] >= min_mask_area], key=lambda m: m["area"])
        return masks
    if len(masks) == 0:
        return np.array([])
    return semantic_segmentation(masks, predictions)


def apply_classifier(image, masks: List[dict], classifier, desired_class: Optional[int] = None, min_mask_area: int = 100, batchsize: int = 32):
    segs = np.array([m["segmentation"].astype(np.uint8) for m in masks])
    return convert_predictions_to_masks(classifier.batch_predict(image, segs, batchsize), masks, desired_class, min_mask_area)
