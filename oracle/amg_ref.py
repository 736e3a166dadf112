"""ORACLE (test infrastructure, never shipped): CPU restatement of
`sam2.SAM2AutomaticMaskGenerator.generate` as the reference drives it
(reference call site: saber/adapters/sam2/predictor.py:70 via
saber/adapters/sam2/amg.py:161-183; construction parameters
saber/adapters/sam2/automask.py:66-78; un-passed upstream defaults mask_threshold=0.0,
crop_nms_thresh=0.7, crop_overlap_ratio=512/1500, min_mask_region_area=0).

Parity status: parity unpinned (third-party `sam2` absent; see oracle/sam2_ref.py).
The helper functions restate upstream sam2/utils/amg.py semantics:
point grids, crop boxes, stability score, mask->box, near-crop-edge filter and
torchvision's greedy NMS (stable descending score sort, IoU > thr suppresses).
"""
import math
from itertools import product
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .sam2_ref import ImagePredictorRef


# ------------------------------------------------------------------ upstream sam2/utils/amg.py
def build_point_grid(n_per_side: int) -> np.ndarray:
    offset = 1 / (2 * n_per_side)
    side = np.linspace(offset, 1 - offset, n_per_side)
    px = np.tile(side[None, :], (n_per_side, 1))
    py = np.tile(side[:, None], (1, n_per_side))
    return np.stack([px, py], axis=-1).reshape(-1, 2)


def build_all_layer_point_grids(n_per_side: int, n_layers: int, scale_per_layer: int) -> List[np.ndarray]:
    return [build_point_grid(int(n_per_side / (scale_per_layer ** i))) for i in range(n_layers + 1)]


def generate_crop_boxes(im_size, n_layers: int, overlap_ratio: float):
    crop_boxes, layer_idxs = [], []
    im_h, im_w = im_size
    short_side = min(im_h, im_w)
    crop_boxes.append([0, 0, im_w, im_h])
    layer_idxs.append(0)

    def crop_len(orig_len, n_crops, overlap):
        return int(math.ceil((overlap * (n_crops - 1) + orig_len) / n_crops))

    for i_layer in range(n_layers):
        n_side = 2 ** (i_layer + 1)
        overlap = int(overlap_ratio * short_side * (2 / n_side))
        crop_w = crop_len(im_w, n_side, overlap)
        crop_h = crop_len(im_h, n_side, overlap)
        x0s = [int((crop_w - overlap) * i) for i in range(n_side)]
        y0s = [int((crop_h - overlap) * i) for i in range(n_side)]
        for x0, y0 in product(x0s, y0s):
            crop_boxes.append([x0, y0, min(x0 + crop_w, im_w), min(y0 + crop_h, im_h)])
            layer_idxs.append(i_layer + 1)
    return crop_boxes, layer_idxs


def calculate_stability_score(masks: torch.Tensor, mask_threshold: float, offset: float) -> torch.Tensor:
    inter = (masks > (mask_threshold + offset)).flatten(-2).sum(-1).to(torch.int32)
    union = (masks > (mask_threshold - offset)).flatten(-2).sum(-1).to(torch.int32)
    return inter / union  # 0/0 -> nan, which fails the >= filter exactly as upstream


def batched_mask_to_box(masks: torch.Tensor) -> torch.Tensor:
    if masks.numel() == 0:
        return torch.zeros(*masks.shape[:-2], 4)
    h, w = masks.shape[-2:]
    in_h, _ = torch.max(masks, dim=-1)
    hc = in_h * torch.arange(h)[None, :]
    bottom, _ = torch.max(hc, dim=-1)
    top, _ = torch.min(hc + h * (~in_h), dim=-1)
    in_w, _ = torch.max(masks, dim=-2)
    wc = in_w * torch.arange(w)[None, :]
    right, _ = torch.max(wc, dim=-1)
    left, _ = torch.min(wc + w * (~in_w), dim=-1)
    empty = (right < left) | (bottom < top)
    out = torch.stack([left, top, right, bottom], dim=-1)
    return out * (~empty).unsqueeze(-1)


def uncrop_boxes_xyxy(boxes: torch.Tensor, crop_box) -> torch.Tensor:
    x0, y0, _, _ = crop_box
    return boxes + torch.tensor([[x0, y0, x0, y0]])


def is_box_near_crop_edge(boxes, crop_box, orig_box, atol: float = 20.0):
    cb = torch.as_tensor(crop_box, dtype=torch.float)
    ob = torch.as_tensor(orig_box, dtype=torch.float)
    b = uncrop_boxes_xyxy(boxes, crop_box).float()
    near_crop = torch.isclose(b, cb[None, :], atol=atol, rtol=0)
    near_img = torch.isclose(b, ob[None, :], atol=atol, rtol=0)
    return torch.any(near_crop & ~near_img, dim=1)


def nms(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """torchvision.ops.nms semantics (single category), float32 arithmetic."""
    n = len(boxes)
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    b = boxes.astype(np.float32)
    order = np.argsort(-scores.astype(np.float32), kind="stable")
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(b[i, 0], b[rest, 0]); yy1 = np.maximum(b[i, 1], b[rest, 1])
        xx2 = np.minimum(b[i, 2], b[rest, 2]); yy2 = np.minimum(b[i, 3], b[rest, 3])
        w = np.maximum(np.float32(0), xx2 - xx1); h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > np.float32(thr)]] = True
    return np.asarray(keep, dtype=np.int64)


# ------------------------------------------------------------------ the generator
class AutomaticMaskGeneratorRef:
    def __init__(self, predictor: ImagePredictorRef, points_per_side=32, points_per_batch=64,
                 pred_iou_thresh=0.8, stability_score_thresh=0.95, stability_score_offset=1.0,
                 mask_threshold=0.0, box_nms_thresh=0.7, crop_n_layers=0, crop_nms_thresh=0.7,
                 crop_overlap_ratio=512 / 1500, crop_n_points_downscale_factor=1,
                 use_m2m=False, multimask_output=True):
        self.predictor = predictor
        self.point_grids = build_all_layer_point_grids(points_per_side, crop_n_layers, crop_n_points_downscale_factor)
        self.points_per_batch = points_per_batch
        self.pred_iou_thresh = pred_iou_thresh
        self.stability_score_thresh = stability_score_thresh
        self.stability_score_offset = stability_score_offset
        self.mask_threshold = mask_threshold
        self.box_nms_thresh = box_nms_thresh
        self.crop_n_layers = crop_n_layers
        self.crop_nms_thresh = crop_nms_thresh
        self.crop_overlap_ratio = crop_overlap_ratio
        self.use_m2m = use_m2m
        self.multimask_output = multimask_output

    @torch.no_grad()
    def generate(self, image: np.ndarray) -> List[Dict[str, Any]]:
        d = self._generate_masks(image)
        anns = []
        for i in range(len(d["masks"])):
            x0, y0, x1, y1 = [float(v) for v in d["boxes"][i]]
            cb = d["crop_boxes"][i]
            anns.append({
                "segmentation": d["masks"][i],
                "area": int(d["masks"][i].sum()),
                "bbox": [x0, y0, x1 - x0, y1 - y0],
                "predicted_iou": float(d["iou_preds"][i]),
                "point_coords": [[float(d["points"][i][0]), float(d["points"][i][1])]],
                "stability_score": float(d["stability_score"][i]),
                "crop_box": [float(cb[0]), float(cb[1]), float(cb[2] - cb[0]), float(cb[3] - cb[1])],
            })
        return anns

    def _generate_masks(self, image):
        orig_size = image.shape[:2]
        crop_boxes, layer_idxs = generate_crop_boxes(orig_size, self.crop_n_layers, self.crop_overlap_ratio)
        parts = [self._process_crop(image, cb, li, orig_size) for cb, li in zip(crop_boxes, layer_idxs)]
        d = _cat(parts)
        if len(crop_boxes) > 1 and len(d["boxes"]):
            cbx = d["crop_boxes"].float()
            scores = 1 / ((cbx[:, 2] - cbx[:, 0]) * (cbx[:, 3] - cbx[:, 1]))
            keep = nms(d["boxes"].float().numpy(), scores.numpy(), self.crop_nms_thresh)
            d = _filter(d, torch.from_numpy(keep))
        d["masks"] = [m.numpy() for m in d["masks"]] if isinstance(d["masks"], list) else list(d["masks"].numpy())
        return d

    def _process_crop(self, image, crop_box, layer_idx, orig_size):
        x0, y0, x1, y1 = crop_box
        crop = image[y0:y1, x0:x1, :]
        crop_hw = crop.shape[:2]
        self.predictor.set_image(crop)
        pts_scale = np.array(crop_hw)[None, ::-1]
        pts_img = self.point_grids[layer_idx] * pts_scale
        parts = []
        for s in range(0, len(pts_img), self.points_per_batch):
            parts.append(self._process_batch(pts_img[s:s + self.points_per_batch], crop_hw, crop_box, orig_size))
        self.predictor.reset_predictor()
        d = _cat(parts)
        keep = nms(d["boxes"].float().numpy(), d["iou_preds"].numpy(), self.box_nms_thresh)
        d = _filter(d, torch.from_numpy(keep))
        d["boxes"] = uncrop_boxes_xyxy(d["boxes"], crop_box)
        d["points"] = d["points"] + torch.tensor([[x0, y0]], dtype=d["points"].dtype)
        d["crop_boxes"] = torch.tensor([crop_box for _ in range(len(d["boxes"]))], dtype=torch.int64).reshape(-1, 4)
        return d

    def _process_batch(self, points, im_size, crop_box, orig_size):
        orig_h, orig_w = orig_size
        P = self.predictor
        points = torch.as_tensor(points, dtype=torch.float32)
        in_pts = P.transform_coords(points, normalize=True, orig_hw=im_size)
        labels = torch.ones(in_pts.shape[0], dtype=torch.int64)
        masks, iou, low = P._predict(in_pts[:, None, :], labels[:, None], multimask_output=self.multimask_output)
        nm = masks.shape[1]
        d = {"masks": masks.flatten(0, 1), "iou_preds": iou.flatten(0, 1),
             "points": points.repeat_interleave(nm, dim=0), "low_res_masks": low.flatten(0, 1)}
        if self.use_m2m:
            in_pts = P.transform_coords(d["points"], normalize=True, orig_hw=im_size)
            labels = torch.ones(in_pts.shape[0], dtype=torch.int64)
            new_m, new_i = [], []
            for s in range(0, len(in_pts), self.points_per_batch):
                e = s + self.points_per_batch
                m2, i2, _ = P._predict(in_pts[s:e, None, :], labels[s:e, None],
                                       mask_input=d["low_res_masks"][s:e, None], multimask_output=False)
                new_m.append(m2); new_i.append(i2)
            d["masks"] = torch.cat(new_m, 0).squeeze(1)
            d["iou_preds"] = torch.cat(new_i, 0).squeeze(1)
        if self.pred_iou_thresh > 0.0:
            d = _filter(d, d["iou_preds"] > self.pred_iou_thresh)
        d["stability_score"] = calculate_stability_score(d["masks"], self.mask_threshold, self.stability_score_offset)
        if self.stability_score_thresh > 0.0:
            d = _filter(d, d["stability_score"] >= self.stability_score_thresh)
        d["masks"] = d["masks"] > self.mask_threshold
        d["boxes"] = batched_mask_to_box(d["masks"])
        keep = ~is_box_near_crop_edge(d["boxes"], crop_box, [0, 0, orig_w, orig_h])
        if not torch.all(keep):
            d = _filter(d, keep)
        x0, y0, x1, y1 = crop_box
        full = torch.zeros(d["masks"].shape[0], orig_h, orig_w, dtype=torch.bool)
        full[:, y0:y1, x0:x1] = d["masks"]
        d["masks"] = full
        del d["low_res_masks"]
        return d


def _filter(d, keep):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v[keep]
        else:
            raise TypeError(k)
    return out


def _cat(parts):
    keys = parts[0].keys()
    return {k: torch.cat([p[k] for p in parts], dim=0) for k in keys}


def amg_from_saber_cfg(predictor: ImagePredictorRef, amg: Optional[dict] = None) -> AutomaticMaskGeneratorRef:
    """Parameter plumbing of reference build_amg (saber/adapters/sam2/automask.py:66-78)
    from a cfgAMG dict (saber/adapters/sam2/amg.py:7-17)."""
    a = dict(npoints=32, points_per_batch=64, pred_iou_thresh=0.7, stability_score_thresh=0.92,
             stability_score_offset=0.7, crop_n_layers=2, box_nms_thresh=0.7,
             crop_n_points_downscale_factor=2, use_m2m=True, multimask_output=True)
    a.update(amg or {})
    return AutomaticMaskGeneratorRef(
        predictor, points_per_side=a["npoints"], points_per_batch=a["points_per_batch"],
        pred_iou_thresh=a["pred_iou_thresh"], stability_score_thresh=a["stability_score_thresh"],
        stability_score_offset=a["stability_score_offset"], crop_n_layers=a["crop_n_layers"],
        box_nms_thresh=a["box_nms_thresh"], crop_n_points_downscale_factor=a["crop_n_points_downscale_factor"],
        use_m2m=a["use_m2m"], multimask_output=a["multimask_output"])
