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
