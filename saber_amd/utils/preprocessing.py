"""Slice / slab preparation (reference: saber/utils/preprocessing.py).

`prepare` (contrast + min-max, :67-80) runs on the device through the engine's K0 kernel - it is part of the
hot path and has no host implementation here.  `normalize` (:20-37) and `project_tomogram` (:39-65) are
volume-level host glue outside the per-slice loop and stay numpy, like the reference."""
from typing import Optional

import numpy as np
import torch


def normalize(image: np.ndarray, rgb: bool = False) -> np.ndarray:
    if rgb:
        lo = image.min(axis=(0, 1), keepdims=True)
        hi = image.max(axis=(0, 1), keepdims=True)
    else:
        lo, hi = image.min(), image.max()
    return (image - lo) / (hi - lo + 1e-8)


def project_tomogram(vol, zSlice: Optional[int] = None, deltaZ: Optional[int] = None):
    if zSlice is None:
        return np.mean(vol, axis=0)
    if deltaZ is None:
        return vol[zSlice, ]
    z0 = int(max(zSlice - deltaZ, 0))
    z1 = int(min(zSlice + deltaZ, vol.shape[0]))
    return np.mean(vol[z0:z1, ], axis=0)


_K0_HANDLES = {}


def _k0_engine(device):
    """a weight-less handle per device: K0 needs the device binding and scratch only"""
    from saber_amd.engine import Engine
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _K0_HANDLES:
        _K0_HANDLES[idx] = Engine.bare(idx)
    return _K0_HANDLES[idx]


def prepare(image, to_rgb: bool = False, engine=None) -> torch.Tensor:
    """Reference signature `prepare(image, to_rgb)` (preprocessing.py:67); `engine` optionally names the handle (and device) to run on,
    default: a weight-less handle on torch's current device.
    image: (H,W) numpy or device tensor (uint16 / float) -> (H,W) float32 device tensor in [0,1]; `to_rgb` is accepted for
    signature parity: the 3x channel repeat of the reference is folded into the encoder's pixel kernel (a gray plane is broadcast
    to the three ImageNet-normalised channels).  (H,W,3) -> (H,W,3) float32 device tensor, filtered over all three axes with ONE
    global min/max exactly as the reference does for RGB arrays.
    K0 computes in float32: prepare(x) == prepare(x.astype(float32)) (the reference's loaders cast to float32, utils/io.py:34; for
    integer arrays the reference's own `image**2` wraps in the integer dtype, a behaviour nobody relies on and not reproduced -
    uint16 slices are widened exactly)."""
    if engine is None:
        dev = image.device if isinstance(image, torch.Tensor) and image.is_cuda else torch.device("cuda", torch.cuda.current_device())
        engine = _k0_engine(dev)
    if isinstance(image, np.ndarray):
        if image.dtype == np.uint16 and image.ndim == 2:
            t = torch.from_numpy(np.ascontiguousarray(image))
        else:
            t = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32))
        image = t.to(engine.device)
    elif image.dtype not in (torch.uint16, torch.float32) or image.dim() == 3:
        image = image.float()
    return engine.prepare(image.contiguous())
