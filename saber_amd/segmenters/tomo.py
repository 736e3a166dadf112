"""tomoSegmenter (reference: saber/segmenters/tomo.py:14-139).  segment_slab's 2-D path (z-smoothing -> normalise ->
slab projection -> segment_image) feeds segment_vol, which continues into SAM2 video propagation on the engine
(SAM2Adapter.segment_volume, saber_amd/adapters/sam2/video.py).  The z Gaussian (filters/gaussian.py:17-74, sigma 5, conv1d) is host
glue outside the per-slice loop and is evaluated with scipy's separable filter of the same kernel."""
from typing import Optional

import numpy as np
import torch

from saber_amd.adapters.base import AdapterConfig
from saber_amd.adapters.sam2.amg import cfgAMG
from saber_amd.segmenters.base import saber3D
from saber_amd.utils import preprocessing as preprocess


def make_gaussian_kernel(sigma: float) -> np.ndarray:
    """Kernel of the reference (filters/gaussian.py:7-15): odd size max(round(3 sigma), 3), taps on linspace(-ks/2, ks/2, ks)."""
    ks = max(round(sigma * 3), 3)
    ks += 1 - ks % 2
    ts = np.linspace(-ks / 2, ks / 2, ks, dtype=np.float32)
    g = np.exp(-(ts / np.float32(sigma)) ** 2 / 2).astype(np.float32)
    return g / g.sum()


def gaussian_smoothing_z(vol: np.ndarray, sigma: float, dim: int = 0) -> np.ndarray:
    """1-D Gaussian along `dim`: conv1d with zero 'same' padding of the reference kernel (filters/gaussian.py:17-74)."""
    from scipy.ndimage import correlate1d
    return correlate1d(np.asarray(vol, dtype=np.float32), make_gaussian_kernel(sigma), axis=dim, mode="constant", cval=0.0)


class tomoSegmenter(saber3D):
    def __init__(self, deviceID: int = 0, cfg: Optional[AdapterConfig] = None, amg_cfg: Optional[cfgAMG] = None, min_mask_area: int = 50):
        super().__init__(deviceID=deviceID, cfg=cfg, amg_cfg=amg_cfg, min_mask_area=min_mask_area)
        self.filter_threshold = 0.5

    @torch.inference_mode()
    def segment_slab(self, vol, slab_thickness: int = 10, zSlice: Optional[int] = None, display: bool = True,
                     text: Optional[str] = None, target_class: Optional[int] = 1):
        self.vol = preprocess.normalize(gaussian_smoothing_z(vol, 5, dim=0))
        if zSlice is None:
            zSlice = int(self.vol.shape[0] // 2)
        self.image0 = preprocess.project_tomogram(self.vol, zSlice, slab_thickness)
        self.segment_image(self.image0, display=display, text_prompt=text, target_class=target_class)
        return self.masks

    def segment(self, vol, thickness: int = 10, zSlice: int = None, text: Optional[str] = None, target_class: Optional[int] = 1,
                save_run: str = None, display: bool = False):
        return self.segment_vol(vol, thickness, zSlice, text, target_class, save_run, display)

    @torch.inference_mode()
    def segment_vol(self, vol, thickness: int, zSlice: int = None, text: Optional[str] = None, target_class: Optional[int] = 1,
                    save_run: str = None, display: bool = False):
        self.is_tomogram_mode = True
        self.segment_slab(vol, thickness, zSlice, display=False, text=text, target_class=target_class)
        if len(self.masks) == 0:
            return None
        # the reference loads the volume once per segmenter (`_vol_loaded` is never reset, tomo.py:117-119), so a pooled segmenter
        # propagates its FIRST tomogram's frames for every later one (known reference bug): here every call loads its own
        self.video_predictor.set_volume(self.vol)
        self._vol_loaded = True
        nx = self.vol.shape[0]
        ny, nz = self.masks[0]["segmentation"].shape
        self.ann_frame_idx = zSlice if zSlice is not None else nx // 2
        return self.propagate((nx, ny, nz))


    def generate_multi_slab(self, vol, thickness, zSlice):
        """tomo.py:141-158 ("highly experimental"): three slab projections a third of a slab apart as the three channels of one image,
        local contrast with a 3-sigma clip and one global min-max; kept as host glue (scipy's uniform_filter over all three axes, like
        the reference), it only sets self.image."""
        from scipy.ndimage import uniform_filter
        planes = [preprocess.project_tomogram(vol, zSlice + d, thickness) for d in (-thickness / 3, 0, thickness / 3)]
        image = np.stack(planes, axis=-1)
        mean = uniform_filter(image, size=500)
        var = np.clip(uniform_filter(image ** 2, size=500) - mean ** 2, a_min=0, a_max=None)
        image = np.clip((image - mean) / (np.sqrt(var) + 1e-8), -3, 3)
        self.image = preprocess.normalize(image, rgb=True)


class multiDepthTomoSegmenter(tomoSegmenter):
    """tomo.py:161-258: segment_vol seeded at num_slabs depths delta_z apart around the centre, binary union, 3-D connected components."""

    def __init__(self, deviceID: int = 0, cfg: Optional[AdapterConfig] = None, amg_cfg: Optional[cfgAMG] = None, target_class: int = 1,
                 min_mask_area: int = 100, min_rel_box_size: float = 0.025):
        self.min_rel_box_size = min_rel_box_size
        self.target_class = target_class
        super().__init__(deviceID=deviceID, cfg=cfg, amg_cfg=amg_cfg, min_mask_area=min_mask_area)
        if target_class < 1:
            raise ValueError("Multi-Depth Tomogram Segmenter only supports Single-Class Segmentation currently.")    # (the reference prints and exits)

    def segment(self, vol, thickness: int, num_slabs: int = 3, delta_z: int = 30, save_run: str = None, display: bool = False):
        self.show_segments = display
        if self.target_class > 0 or self.classifier is None:
            return self.single_segment(vol, thickness, num_slabs, delta_z)
        print("Multiclass Segmentation is not implemented yet")     # reference behaviour (tomo.py:201-203)

    @torch.inference_mode()
    def single_segment(self, vol, thickness, num_slabs, delta_z):
        from saber_amd.segmenters import utils
        depth = vol.shape[0]
        combined = np.zeros(vol.shape, dtype=np.uint16)
        for i in range(num_slabs):
            centre = int(depth // 2 + (i - num_slabs // 2) * delta_z)
            if centre < 0 or centre >= depth:
                print(f"Skipping slab {i}: slab_center={centre} out of range (0-{depth - 1})")
                continue
            masks3d = self.segment_vol(vol, thickness, zSlice=centre, display=False)
            if masks3d is None:
                continue
            np.maximum(combined, (masks3d > 0).astype(np.uint16), out=combined)
        return utils.separate_masks(combined)
