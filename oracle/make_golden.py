"""Generates tests/golden/saber_glue.npz by IMPORTING the reference's own glue code from /root/reference
(authoring container only; the reference cannot travel to the GPU box).  Third-party packages the reference
imports but that are absent here are stubbed with MagicMock - none of them is called by the functions captured.

    python -m oracle.make_golden

Captured (inputs and outputs of the reference's own functions):
  prepare / contrast / normalize / project_tomogram      saber/utils/preprocessing.py
  make_gaussian_kernel / gaussian_smoothing (z)          saber/filters/gaussian.py
  remove_duplicate_masks / separate_masks                saber/segmenters/utils.py
  saber2D.get_sliding_windows, _apply_classifier,
  propagationSegmenter.slice_by_slice (fake adapter)     saber/segmenters/{base,propagation}.py
  cfgAMG defaults, SAM2AdapterConfig defaults            saber/adapters/sam2/amg.py, saber/adapters/base.py
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "saber_glue.npz")


def _stub():
    for name in ["mrcfile", "skimage", "skimage.io", "skimage.transform", "cv2", "zarr", "copick", "copick_utils", "copick_utils.io",
                 "copick_utils.readers", "copick_utils.writers", "rich_click", "sam2", "sam2.build_sam", "sam2.automatic_mask_generator",
                 "sam2.sam2_image_predictor", "imageio", "matplotlib", "matplotlib.pyplot", "matplotlib.colors", "matplotlib.widgets",
                 "matplotlib.patches", "matplotlib.cm", "napari", "monai", "lightning", "tqdm"]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    sys.modules["tqdm"].tqdm = lambda x, **k: x
    sys.path.insert(0, REF)


def synthetic_masks(rng, n, H, W):
    """blobby bool masks, some near-duplicates of each other"""
    yy, xx = np.mgrid[:H, :W]
    out = []
    for i in range(n):
        cy, cx, r = rng.integers(10, H - 10), rng.integers(10, W - 10), rng.integers(5, 20)
        m = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
        out.append(m)
        if i % 3 == 0:  # a near duplicate: one-pixel dilation along x
            d = m.copy()
            d[:, 1:] |= m[:, :-1]
            out.append(d)
    return out


def main():
    _stub()
    from saber.utils import preprocessing as prep
    from saber.segmenters import utils as sutils
    from saber.filters import gaussian as gauss
    from saber.adapters.sam2.amg import cfgAMG
    from saber.adapters.base import SAM2AdapterConfig
    from oracle import saber_ref
    G = {}
    # ---- prepare on the config-2 uint16 slice
    raw = saber_ref.synthetic_slice(seed=0)
    G["prep_raw_seed"] = np.array(0)
    f = raw.astype(np.float32)
    p = prep.prepare(f, to_rgb=True)
    G["prep_out_sub"] = p[::16, ::16, 0].copy()
    G["prep_out_stats"] = np.array([p.min(), p.max(), p.mean(), p.std()], dtype=np.float64)
    G["prep_out_rgb_equal"] = np.array(bool((p[..., 0] == p[..., 1]).all() and (p[..., 0] == p[..., 2]).all()))
    small = np.random.default_rng(5).normal(100, 20, (600, 700)).astype(np.float32)
    G["prep_small_in"] = small
    G["prep_small_contrast"] = prep.contrast(small, std_cutoff=3)
    G["prep_small_out"] = prep.prepare(small, to_rgb=False)
    vol = np.random.default_rng(6).uniform(-1, 1, (20, 32, 40)).astype(np.float32)
    G["vol_in"] = vol
    G["proj_z10_d3"] = prep.project_tomogram(vol, 10, 3)
    G["proj_z1_d5"] = prep.project_tomogram(vol, 1, 5)
    G["proj_z7"] = prep.project_tomogram(vol, 7, None)
    G["proj_all"] = prep.project_tomogram(vol)
    G["vol_normalized"] = prep.normalize(vol)
    G["gauss_kernel_s5"] = gauss.make_gaussian_kernel(5).numpy()
    G["gauss_z_s5"] = gauss.gaussian_smoothing(vol, 5, dim=0)
    # ---- duplicate removal / separate_masks
    rng = np.random.default_rng(11)
    masks = synthetic_masks(rng, 12, 96, 128)
    stab = rng.uniform(0.9, 1.0, len(masks))
    dicts = [{"segmentation": m, "area": int(m.sum()), "stability_score": float(s), "id": i} for i, (m, s) in enumerate(zip(masks, stab))]
    kept = sutils.remove_duplicate_masks(dicts)
    G["dedup_masks"] = np.stack(masks)
    G["dedup_stab"] = stab
    G["dedup_kept_ids"] = np.array([d["id"] for d in kept])
    lab = np.zeros((24, 64, 64), dtype=np.uint16)
    r2 = np.random.default_rng(12)
    zz, yy, xx = np.mgrid[:24, :64, :64]
    for k in range(9):
        cz, cy, cx, r = r2.integers(2, 22), r2.integers(5, 59), r2.integers(5, 59), r2.integers(3, 9)
        lab[(zz - cz) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = k + 1
    G["sep_in"] = lab
    G["sep_out_default"] = sutils.separate_masks(lab)
    G["sep_out_min5"] = sutils.separate_masks(lab, min_mask_area=5)
    G["sep_out_empty"] = sutils.separate_masks(np.zeros((3, 8, 8), dtype=np.uint16))
    # ---- saber2D / propagationSegmenter with a fake adapter (the reference's own loop + paint order)
    import saber.segmenters.base as sbase
    import saber.segmenters.propagation as sprop
    vol_masks = [synthetic_masks(np.random.default_rng(100 + z), 4, 48, 64) for z in range(6)]
    vol_stab = [np.random.default_rng(200 + z).uniform(0.9, 1.0, len(vol_masks[z])) for z in range(6)]

    class FakeAdapter:
        def __init__(self): self.z = 0
        def segment_image_2d(self, image, text_prompt=None, threshold=None):
            z = int(round(float(image[0, 0])))
            return [{"segmentation": m, "area": int(m.sum()), "stability_score": float(s), "bbox": [0, 0, 1, 1]}
                    for m, s in zip(vol_masks[z], vol_stab[z])]
        def reset_state(self): pass

    sbase.get_adapter = lambda cfg, dev: FakeAdapter()
    sbase.io.get_available_devices = lambda d=None: "cpu"
    seg = sprop.propagationSegmenter(amg_cfg=cfgAMG(), min_mask_area=30)
    volume = np.zeros((6, 48, 64), dtype=np.float32)
    for z in range(6):
        volume[z] = z
    G["sbs_masks"] = np.array([np.stack(m) for m in vol_masks], dtype=object) if False else np.stack([np.stack(m) for m in vol_masks])
    G["sbs_stab"] = np.stack(vol_stab)
    G["sbs_out"] = seg.slice_by_slice(volume, None)
    planes = np.zeros((6, 48, 64), dtype=np.uint16)
    for z in range(6):
        ms = seg.segment_image(volume[z], display=False)
        for idx, m in enumerate(ms):
            planes[z][m["segmentation"]] = idx + 1
    G["sbs_planes"] = planes
    s2 = sbase.saber2D(amg_cfg=cfgAMG())
    for shp in [(1024, 1024), (600, 900), (300, 300)]:
        G[f"windows_{shp[0]}x{shp[1]}"] = np.array(s2.get_sliding_windows(shp))
    G["cfgamg_defaults_keys"] = np.array(sorted(cfgAMG().dict().keys()))
    G["cfgamg_defaults_vals"] = np.array([str(cfgAMG().dict()[k]) for k in sorted(cfgAMG().dict().keys())])
    c = SAM2AdapterConfig()
    G["adaptercfg_defaults"] = np.array([c.model_type, c.cfg, str(c.checkpoint), str(c.num_maskmem), str(c.light_modality), str(c.min_mask_area)])
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **G)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(G), "arrays")


if __name__ == "__main__":
    main()
