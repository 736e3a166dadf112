"""Generates tests/golden/saber_video_glue.npz by IMPORTING the reference's own saber/filters/estimate_thickness.py (authoring container
only): fit_organelle_boundaries on object-score traces of the kind SAM2Adapter.segment_volume collects (a bump over z, a plateau cut
off at the volume edge, a flat trace that makes both fits fail, a noisy bump).

    python -m oracle.make_golden_video_glue
"""
import os

import numpy as np

from oracle.make_golden import _stub

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "saber_video_glue.npz")


def traces(n=40, seed=3):
    rng = np.random.default_rng(seed)
    z = np.arange(n, dtype=np.float64)
    t = np.zeros((n, 5))
    t[:, 0] = 6.0 * np.exp(-(z - 18.0) ** 2 / (2 * 5.0 ** 2)) - 1.0
    t[:, 1] = np.clip(4.0 - 0.02 * (z - 8.0) ** 2, -2.0, None)
    t[:, 2] = -1.5
    t[:, 3] = 5.0 * np.exp(-(z - 25.0) ** 2 / (2 * 3.0 ** 2)) + rng.normal(0, 0.4, n) - 0.5
    t[:, 4] = np.where((z > 10) & (z < 30), 3.0, -3.0) + rng.normal(0, 0.2, n)
    return t


def main():
    _stub()
    from saber.filters import estimate_thickness as et
    fs = traces()
    out = et.fit_organelle_boundaries(fs.copy(), plot=False)
    np.savez_compressed(OUT, frame_scores=fs, boundaries=out)
    print("wrote", OUT, out.shape, np.round(out.max(0), 3))


if __name__ == "__main__":
    main()
