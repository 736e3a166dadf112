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


def prepare(image, engine, to_rgb: bool = False) -> torch.Tensor:
    """image: (H,W) numpy or device tensor (uint16 / float).  Returns the (H,W) float32 device tensor in [0,1];
    `to_rgb` is accepted for signature parity: the 3x channel repeat of the reference is folded into the
    encoder's pixel kernel (a gray plane is broadcast to the three ImageNet-normalised channels)."""
    if isinstance(image, np.ndarray):
        if image.dtype == np.uint16:
            t = torch.from_numpy(np.ascontiguousarray(image))
        else:
            t = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32))
        image = t.to(engine.device)
    elif image.dtype not in (torch.uint16, torch.float32):
        image = image.float()
    return engine.prepare(image.contiguous())
