"""Generates tests/golden/saber_rgb_prepare.npz by IMPORTING the reference's own prep.prepare from /root/reference (authoring
container only) on an (H,W,3) array: the path SAM2Adapter.segment_image_2d takes for RGB input
(saber/adapters/sam2/predictor.py:58-59 -> saber/utils/preprocessing.py:67-80: uniform_filter(size=500) over ALL three axes,
clip +-3 sigma, one global min/max, no channel repeat).

    python -m oracle.make_golden_rgb
"""
import os

import numpy as np

from oracle.make_golden import _stub

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "saber_rgb_prepare.npz")


def rgb_input(seed=11, H=140, W=120):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[:H, :W]
    img = rng.normal(0.4, 0.1, (H, W, 3))
    img[..., 0] += 0.3 * np.sin(xx / 17.0)
    img[..., 1] += 0.2 * (yy / H)
    img[..., 2] *= 1.0 + 0.5 * ((yy - 60) ** 2 + (xx - 50) ** 2 < 900)
    return img.astype(np.float32)


def main():
    _stub()
    from saber.utils import preprocessing as prep
    x = rgb_input()
    y = prep.prepare(x, to_rgb=False)
    assert y.shape == x.shape
    np.savez_compressed(OUT, rgb_in=x, rgb_out=y.astype(np.float32), rgb_out_dtype=np.array(str(y.dtype)))
    print("wrote", OUT, y.dtype, float(y.min()), float(y.max()))


if __name__ == "__main__":
    main()
