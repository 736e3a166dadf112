"""Host-side mask utilities with the semantics of the reference's saber/segmenters/utils.py.

remove_duplicate_masks  <- saber/segmenters/utils.py:5-86   (pairwise area-ratio >= 0.9 and IoU >= 0.9 => duplicates;
                           groups are formed greedily in list order; the group's max stability_score survives)
separate_masks          <- saber/segmenters/utils.py:88-131 (26-connected 3-D components, drop < min_mask_area*10 voxels,
                           compact relabel to uint32)

`duplicate_groups_from_counts` is the same grouping driven by integer pixel counts (areas + pairwise intersections),
so the device path can decide duplicates from bit-packed masks without materialising bool arrays on the host.
"""
from typing import Any, Dict, List, Sequence

import numpy as np
from scipy import ndimage as ndi


def duplicate_groups_from_counts(areas: Sequence[int], inter: np.ndarray, stability: Sequence[float],
                                 iou_threshold: float = 0.9, area_threshold: float = 0.9,
                                 pixel_counts: Sequence[int] = None) -> List[int]:
    """Indices kept by the reference's duplicate removal, in output order.

    areas[i] = the dict's 'area' (used for the area-ratio test), inter[i, j] = |mask_i & mask_j| and
    pixel_counts[i] = |mask_i| (exact integer counts, default = areas) for the IoU, evaluated exactly as the
    reference does on bool arrays (integer counts divided in float64)."""
    n = len(areas)
    counts = areas if pixel_counts is None else pixel_counts
    done = [False] * n
    keep: List[int] = []
    for i in range(n):
        if done[i]:
            continue
        group = [i]
        for j in range(i + 1, n):
            if done[j]:
                continue
            hi = max(areas[i], areas[j])
            ratio = min(areas[i], areas[j]) / hi if hi > 0 else 0
            it = int(inter[i, j])
            union = int(counts[i]) + int(counts[j]) - it
            iou = it / union if union > 0 else 0.0
            if ratio < area_threshold or iou < iou_threshold:
                continue
            group.append(j)
            done[j] = True
        best = group[0]
        for g in group[1:]:  # the reference's max() keeps the first maximal element
            if stability[g] > stability[best]:
                best = g
        keep.append(best)
        done[i] = True
    return keep


DEVICE_ROW_KEY = "_device_row"


class DeviceMaskRows:
    """The bit-packed device masks an AMG dict list was unpacked from (EngineMaskGenerator.generate).  A dict that carries
    (rows, i) under DEVICE_ROW_KEY says "my `segmentation` is row i of rows.bits"; the holder must drop the key when it changes the
    array (saber2D strips it before the list leaves _apply_classifier, which also lets the device buffer go)."""

    def __init__(self, engine, bits, H: int, W: int):
        self.engine, self.bits, self.H, self.W = engine, bits, H, W

    def intersections(self, idx: Sequence[int]) -> np.ndarray:
        import torch
        sel = self.bits[torch.as_tensor(list(idx), dtype=torch.long, device=self.bits.device)].contiguous()
        return self.engine.pair_intersections(sel, self.H, self.W).cpu().numpy().astype(np.int64)


def _shared_device_rows(masks):
    refs = [m.get(DEVICE_ROW_KEY) for m in masks]
    if any(r is None for r in refs) or any(r[0] is not refs[0][0] for r in refs):
        return None
    return refs[0][0], [r[1] for r in refs]


def strip_device_rows(masks):
    for m in masks:
        m.pop(DEVICE_ROW_KEY, None)
    return masks


def remove_duplicate_masks(masks: List[Dict[str, Any]], iou_threshold: float = 0.9, area_threshold: float = 0.9,
                           verbose: bool = False) -> List[Dict[str, Any]]:
    """Drop-in for the reference function on SAM-AMG dict lists with full bool `segmentation` arrays.  When every dict still refers to
    its bit-packed row on the device the pixel counts come from the pair-intersection kernel (same integers, no n x HW host matrix)."""
    if len(masks) == 0:
        return []
    shared = _shared_device_rows(masks)
    if shared is not None:
        inter = shared[0].intersections(shared[1])
        keep = duplicate_groups_from_counts([m["area"] for m in masks], inter, [m.get("stability_score", 0) for m in masks],
                                            iou_threshold, area_threshold, pixel_counts=inter.diagonal())
        return [masks[i] for i in keep]
    flat = np.stack([np.asarray(m["segmentation"], dtype=bool).ravel() for m in masks]).astype(np.float32)
    inter = np.rint(flat @ flat.T).astype(np.int64)  # exact: counts < 2^24
    keep = duplicate_groups_from_counts([m["area"] for m in masks], inter, [m.get("stability_score", 0) for m in masks],
                                        iou_threshold, area_threshold, pixel_counts=inter.diagonal())
    return [masks[i] for i in keep]


def separate_masks(combined_mask: np.ndarray, min_mask_area: int = 100) -> np.ndarray:
    """3-D 26-connected components of the foreground, small components removed, labels compacted (uint32)."""
    fg = np.ascontiguousarray(np.asarray(combined_mask).astype(bool))
    out = np.zeros(fg.shape, dtype=np.uint32)
    if not fg.any():
        return out
    zz, yy, xx = np.nonzero(fg)
    sl = (slice(zz.min(), zz.max() + 1), slice(yy.min(), yy.max() + 1), slice(xx.min(), xx.max() + 1))
    lab, _ = ndi.label(fg[sl], structure=np.ones((3, 3, 3), dtype=bool))
    min_vol = min_mask_area * 10
    sizes = np.bincount(lab.ravel())
    alive = sizes > 0
    if min_vol > 1:
        alive &= sizes >= min_vol
    alive[0] = False
    remap = np.zeros(sizes.size, dtype=np.uint32)
    remap[alive] = np.arange(1, int(alive.sum()) + 1, dtype=np.uint32)
    out[sl] = remap[lab]
    return out


def resize_mask_nearest(mask: np.ndarray, shape) -> np.ndarray:
    """skimage.transform.resize(mask, shape, order=0, anti_aliasing=False) of a bool mask as SAM2Adapter.segment_volume applies it when
    the video resolution differs from the tomogram's (adapters/sam2/predictor.py:294-296): nearest source pixel of every output pixel centre."""
    H, W = mask.shape
    ys = np.clip(np.floor((np.arange(shape[0]) + 0.5) * H / shape[0]).astype(np.int64), 0, H - 1)
    xs = np.clip(np.floor((np.arange(shape[1]) + 0.5) * W / shape[1]).astype(np.int64), 0, W - 1)
    return mask[np.ix_(ys, xs)]
