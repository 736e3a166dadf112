"""saber.filters.masks.fast_3d_gaussian_smoothing (saber/filters/masks.py:230-309) on the MI355X.

segment_tomogram_core smooths the segmenter's label volume with it (scale=0.05) before writing the segmentation
(saber/entry_points/inference_core.py:68-74).  The reference runs three whole-volume conv3d passes per label; the device
version (csrc/smooth3d.hip) confines every label to its bounding box and runs all labels in three launches."""
import numpy as np
import torch

from ._context import handle

_TORCH_OK = (torch.uint8, torch.int16, torch.uint16, torch.int32, torch.uint32)


def _estimate_feature_size_3d(binary_volume, scale=0.075):
    """sigma = scale * diameter of the sphere with the mask's volume (masks.py:289-309); host arithmetic only."""
    volume = np.sum(binary_volume)
    approx_diameter = 2 * ((3 * volume) / (4 * np.pi)) ** (1 / 3)
    return scale * approx_diameter


def fast_3d_gaussian_smoothing(volume, scale=0.075, deviceID=None):
    """volume: 3-D label array (numpy, any integer dtype, or a device tensor).  Returns the smoothed uint8 label volume: numpy for
    numpy input (reference signature), a device tensor for tensor input."""
    is_tensor = isinstance(volume, torch.Tensor)
    if volume.ndim != 3:
        raise ValueError(f"Expected 3D input, got {volume.ndim}D")
    eng = handle(volume.device if is_tensor else deviceID)
    if is_tensor:
        lab = volume
        if lab.dtype == torch.bool:
            lab = lab.to(torch.uint8)
        elif lab.dtype == torch.int64:
            lab = lab.to(torch.int32)
        if lab.dtype not in _TORCH_OK:
            raise ValueError(f"fast_3d_gaussian_smoothing: unsupported label dtype {volume.dtype}")
    else:
        v = np.asarray(volume)
        if v.dtype == np.bool_:
            v = v.astype(np.uint8)
        if not np.issubdtype(v.dtype, np.integer):
            raise ValueError(f"fast_3d_gaussian_smoothing: label volumes are integer arrays, got {v.dtype}")
        if v.size and (v.min() < 0 or v.max() > 2 ** 22):
            raise ValueError("fast_3d_gaussian_smoothing: label values must lie in [0, 2^22]")
        if v.dtype.itemsize not in (1, 2, 4) or v.dtype.kind == "i" and v.dtype.itemsize == 1:
            v = v.astype(np.uint32)
        v = np.ascontiguousarray(v)
        view = {1: np.uint8, 2: np.int16, 4: np.int32}[v.dtype.itemsize]       # torch.from_numpy has no uint16 / uint32 on every build
        lab = torch.from_numpy(v.view(view)).to(eng.device)
    with torch.cuda.device(eng.device):
        out, _ = eng.smooth_labels(lab.contiguous(), scale)
    return out if is_tensor else out.cpu().numpy()


# ------------------------------------------------------------------------------------------------ classifier filter
def apply_classifier(image, masks, classifier, desired_class: int = None, min_mask_area: int = 100, batchsize: int = 32):
    """saber/filters/masks.py:8-21: class probabilities of every candidate mask (the device part, Predictor.batch_predict), then
    the host resolution below."""
    segs = np.array([m["segmentation"].astype(np.uint8) for m in masks])
    predictions = classifier.batch_predict(image, segs, batchsize)
    return convert_predictions_to_masks(predictions, masks, desired_class, min_mask_area)


def convert_predictions_to_masks(predictions, masks, desired_class: int = None, min_mask_area: int = 100):
    """saber/filters/masks.py:23-62.  desired_class > 0: the masks predicted as that class, merged where they overlap
    (_consensus_based_resolution), area-filtered, ascending area.  Otherwise one merged mask per non-background class."""
    if isinstance(masks, np.ndarray):
        masks = masks_to_list(masks)
    predicted = np.argmax(predictions, axis=1)
    if desired_class > 0 and desired_class is not None:        # (operand order of the reference: desired_class=None raises there as well)
        conf = predictions[:, desired_class]
        idx = [i for i, p in enumerate(predicted) if p == desired_class]
        masks = [masks[i] for i in idx]
        conf = conf[idx]
        if len(masks) > 0:
            masks = _consensus_based_resolution(masks[0]["segmentation"].shape, masks, conf)
            masks = sorted([m for m in masks if m["area"] >= min_mask_area], key=lambda m: m["area"], reverse=False)
        return masks
    if len(masks) == 0:
        return np.array([])
    return _semantic_segmentation(masks, predictions)


def _consensus_based_resolution(image_shape, masks, confidences):
    """saber/filters/masks.py:64-122: connected components of the union of the masks; each component's score is the mean, over its
    pixels, of the overlap-averaged class confidence."""
    from scipy import ndimage
    h, w = image_shape
    conf_map = np.zeros((h, w), dtype=np.float32)
    count = np.zeros((h, w), dtype=np.int32)
    for m, c in zip(masks, confidences):
        conf_map += m["segmentation"] * c
        count += m["segmentation"]
    with np.errstate(divide="ignore", invalid="ignore"):
        avg = np.nan_to_num(np.divide(conf_map, count))
    labeled, ncomp = ndimage.label(count > 0)
    out = []
    for lab in range(1, ncomp + 1):
        comp = labeled == lab
        score = float(np.mean(avg[comp]))
        ys, xs = np.where(comp)
        y0, y1, x0, x1 = int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())
        out.append({"segmentation": comp, "area": int(np.sum(comp)), "bbox": [x0, y0, x1 - x0, y1 - y0], "predicted_iou": score,
                    "point_coords": [[int((x0 + x1) / 2), int((y0 + y1) / 2)]], "stability_score": score, "crop_box": [x0, y0, x1, y1]})
    return out


def _semantic_segmentation(masks, predictions):
    """saber/filters/masks.py:124-158"""
    predicted = np.argmax(predictions, axis=1)
    out = [{"segmentation": np.zeros(masks[0]["segmentation"].shape, dtype=np.uint8), "area": 0, "label": k} for k in range(1, predictions.shape[1])]
    for m, p in zip(masks, predicted):
        if p > 0:
            out[p - 1]["segmentation"] = np.logical_or(out[p - 1]["segmentation"], m["segmentation"]).astype(bool)
            out[p - 1]["area"] += m["area"]
    return out


def masks_to_array(mask_list):
    """filters/masks.py:157-182: (n, H, W) stack with mask j holding the value j + 1, in the narrowest unsigned type that holds n.  The
    reference looks at mask_list[0] before it tests for an empty list, so an empty list raises IndexError there and here."""
    if not isinstance(mask_list, list):
        print("Returning None")
        return None
    nx, ny = mask_list[0]["segmentation"].shape
    n = len(mask_list)
    dtype = np.uint8 if n < 256 else (np.uint16 if n < 65536 else np.uint32)
    out = np.zeros((n, nx, ny), dtype=dtype)
    for j, m in enumerate(mask_list):
        out[j] = m["segmentation"].astype(dtype) * (j + 1)
    return out


def masks_to_list(masks):
    """saber/filters/masks.py:188-206"""
    if isinstance(masks, list):
        return masks
    return [{"segmentation": masks == v, "area": np.sum((masks == v) > 0)} for v in np.unique(masks)]
