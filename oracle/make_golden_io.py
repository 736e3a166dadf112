"""Generates tests/golden/saber_io_glue.npz by IMPORTING the reference's own functions from /root/reference (authoring container only)
for the host glue of the micrograph path:

    saber/filters/downsample.py   FourierRescale2D.run / run_resolution (odd and even sizes), FourierRescale3D.run
    saber/utils/zarr_writer.py    add_attributes (2-D and 3-D), _to_jsonable
    saber/filters/masks.py        masks_to_array (uint8 and uint16 label stacks)

    python -m oracle.make_golden_io
"""
import json
import os

import numpy as np
import torch

from oracle.make_golden import _stub

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "saber_io_glue.npz")


class _Attrs(dict):
    pass


class _Group:
    def __init__(self):
        self.attrs = _Attrs()


def inputs():
    rng = np.random.default_rng(31)
    return {"img_even": rng.normal(size=(64, 96)).astype(np.float32), "img_odd": rng.normal(size=(65, 97)).astype(np.float32),
            "vol": rng.normal(size=(9, 16, 20)).astype(np.float32)}


def main():
    _stub()
    from saber.filters import downsample as ds
    from saber.filters import masks as fm
    from saber.utils import zarr_writer as zw
    G = dict(inputs())
    cpu = torch.device("cpu")
    G["r2_even_2"] = ds.FourierRescale2D.run(G["img_even"], 2.0, device=cpu)
    G["r2_odd_1p7"] = ds.FourierRescale2D.run(G["img_odd"], 1.7, device=cpu)
    G["r2_res_even"] = ds.FourierRescale2D.run_resolution(G["img_even"], 1.5, 4.0, device=cpu)
    f3 = ds.FourierRescale3D(5.0, (10.0, 10.0, 7.5))
    f3.device = cpu
    G["r3"] = f3.run(G["vol"])
    g2, g3 = _Group(), _Group()
    zw.add_attributes(g2, 0.5)
    zw.add_attributes(g3, 0.5, True, 1.25)
    sample = {"a": np.int64(3), "b": np.float32(0.5), "c": np.arange(3), "d": (1, 2), "e": {"k": np.bool_(True)}, 7: None, "s": "x"}
    G["json"] = np.array(json.dumps({"attrs2d": dict(g2.attrs), "attrs3d": dict(g3.attrs), "jsonable": zw._to_jsonable(sample)}, sort_keys=True))
    rng = np.random.default_rng(5)
    segs = [{"segmentation": rng.uniform(size=(12, 10)) > 0.6} for _ in range(5)]
    G["m2a_in"] = np.stack([s["segmentation"] for s in segs])
    G["m2a_out"] = fm.masks_to_array(segs)
    G["m2a_out_300_dtype"] = np.array(str(fm.masks_to_array(segs * 60).dtype))
    np.savez_compressed(OUT, **G)
    print("wrote", OUT, {k: (v.shape, str(v.dtype)) for k, v in G.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()
