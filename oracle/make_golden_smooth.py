"""Generates tests/golden/saber_smooth.npz by IMPORTING the reference's own post-processing step from /root/reference
(authoring container only; the reference cannot travel to the GPU box):

    fast_3d_gaussian_smoothing / _estimate_feature_size_3d      saber/filters/masks.py:230-309
    gaussian_smoothing_3d                                       saber/filters/gaussian.py:76-138

    python -m oracle.make_golden_smooth

This is the step `segment_tomogram_core` applies to the stitched label volume right after the hot path
(saber/entry_points/inference_core.py:68-74, scale=0.05) - SURVEY.md section 8(f) rank 2.  Inputs and outputs only are stored.
"""
import os

import numpy as np
import torch

from oracle.make_golden import _stub

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "saber_smooth.npz")


def blob_volume(seed, shape, n, rmin, rmax, labels=None, dtype=np.uint32):
    """label volume of n ellipsoids (later ones overwrite), plus a single voxel and a border-touching blob"""
    rng = np.random.default_rng(seed)
    Z, H, W = shape
    zz, yy, xx = np.mgrid[:Z, :H, :W]
    lab = np.zeros(shape, dtype=dtype)
    for k in range(n):
        cz, cy, cx = rng.integers(0, Z), rng.integers(0, H), rng.integers(0, W)
        rz, ry, rx = rng.uniform(rmin, rmax, 3)
        m = ((zz - cz) / rz) ** 2 + ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
        # salt the surface so the smoothing has something to remove
        m &= rng.uniform(0, 1, shape) < 0.93
        lab[m] = (k + 1) if labels is None else labels[k]
    return lab


def main():
    _stub()
    import saber.filters.masks as fm
    import saber.filters.gaussian as fg
    cpu = torch.device("cpu")
    fm.io.get_available_devices = lambda deviceID=None: cpu
    fg.io.get_available_devices = lambda deviceID=None: cpu
    G = {}
    # case A: 10 blobs, default scale and the scale inference_core uses
    a = blob_volume(21, (24, 56, 64), 10, 3.0, 12.0)
    a[3, 5, 7] = 11                                      # a single voxel: sigma 0.09, kernel size 1 -> kept as is
    G["a_in"] = a
    G["a_out_s075"] = fm.fast_3d_gaussian_smoothing(a.copy(), scale=0.075)
    G["a_out_s05"] = fm.fast_3d_gaussian_smoothing(a.copy(), scale=0.05)
    # case B: label values beyond uint8 (the reference's result array is uint8) and a uint16 input
    b = blob_volume(22, (16, 40, 48), 5, 4.0, 10.0, labels=[3, 200, 300, 515, 70], dtype=np.uint16)
    G["b_in"] = b
    G["b_out_s05"] = fm.fast_3d_gaussian_smoothing(b.copy(), scale=0.05)
    # case C: one large label (kernel radius 8) next to a thin plate that the threshold erases, and an empty volume
    c = np.zeros((20, 64, 72), dtype=np.uint32)
    zz, yy, xx = np.mgrid[:20, :64, :72]
    c[(zz - 10) ** 2 * 9 + (yy - 30) ** 2 + (xx - 30) ** 2 < 26 ** 2] = 1
    c[4:5, 2:60, 60:70] = 2
    G["c_in"] = c
    G["c_out_s075"] = fm.fast_3d_gaussian_smoothing(c.copy(), scale=0.075)
    G["empty_out"] = fm.fast_3d_gaussian_smoothing(np.zeros((4, 8, 8), dtype=np.uint32))
    # the float field itself for two (mask, sigma) pairs + the sigma estimate
    for name, lab, sig in (("a3", a == 3, 1.3), ("c1", c == 1, 2.7)):
        G[f"field_{name}_sigma"] = np.array(sig)
        G[f"field_{name}"] = fg.gaussian_smoothing_3d(lab, sig, cpu)
    G["sigma_est"] = np.array([fm._estimate_feature_size_3d(a == k, 0.05) for k in range(1, 12)], dtype=np.float64)
    np.savez_compressed(OUT, **G)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(G), "arrays")
    for k in ("a_out_s075", "a_out_s05", "b_out_s05", "c_out_s075"):
        print(k, G[k].dtype, np.unique(G[k]))


if __name__ == "__main__":
    main()
