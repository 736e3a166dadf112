"""saber.filters.gaussian.gaussian_smoothing_3d (saber/filters/gaussian.py:76-138) on the MI355X."""
import numpy as np
import torch

from ._context import handle


def gaussian_smoothing_3d(volume, sigma, device=None):
    """Separable zero-padded 3-D Gaussian of a binary volume: int(6 sigma + 1) taps (made odd) along x, then y, then z, in fp32.
    volume: 3-D numpy array (bool / 0-1 values, as fast_3d_gaussian_smoothing passes it) or a device tensor; returns the float32
    field as numpy (reference signature) or, for tensor input, as a tensor on the same device."""
    is_tensor = isinstance(volume, torch.Tensor)
    if volume.ndim != 3:
        raise ValueError(f"Expected 3D input, got {volume.ndim}D")
    eng = handle(volume.device if is_tensor else device)
    if is_tensor:
        m = volume
    else:
        v = np.asarray(volume)
        if v.dtype != np.bool_ and not np.isin(v, (0, 1)).all():
            raise ValueError("gaussian_smoothing_3d: the MI355X filter takes a 0/1 mask (what fast_3d_gaussian_smoothing passes)")
        m = torch.from_numpy(np.ascontiguousarray(v.astype(np.uint8))).to(eng.device)
    if m.dtype not in (torch.bool, torch.uint8):
        m = (m != 0)
    with torch.cuda.device(eng.device):
        out = eng.gaussian_smoothing_3d(m.contiguous(), float(sigma))
    return out if is_tensor else out.cpu().numpy()
