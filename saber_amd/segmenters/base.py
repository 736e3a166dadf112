"""saber2D / saber3D (reference: saber/segmenters/base.py:18-280): the 2-D segmentation entry used by every
volumetric loop.  Behaviour kept: adapter construction from an AdapterConfig or a bare cfgAMG, the non-sliding and
sliding-window branches of segment_image, the min-area filter, duplicate removal and ascending-area ordering of
_apply_classifier (with and without a classifier), get_sliding_windows, rasterize_masks, and saber3D.propagate's contract."""
from typing import List, Optional, Tuple

import numpy as np
import torch

from saber_amd.adapters.base import AdapterConfig, SAM2AdapterConfig, get_adapter
from saber_amd.adapters.sam2.amg import cfgAMG
from saber_amd.segmenters import utils
from saber_amd.utils import io


class saber2D:
    def __init__(self, deviceID: int = 0, cfg: Optional[AdapterConfig] = None, amg_cfg: Optional[cfgAMG] = None,
                 min_mask_area: int = 50, window_size: int = 256, overlap_ratio: float = 0.25):
        if cfg is None and amg_cfg is None:
            raise ValueError("Either Provide an AdapterConfig or AMG Config!")
        if cfg is None:
            cfg = SAM2AdapterConfig(amg_cfg=amg_cfg, min_mask_area=min_mask_area)
        self.min_mask_area = min_mask_area
        self.window_size = window_size
        self.overlap_ratio = overlap_ratio
        self.device = io.get_available_devices(deviceID)
        self.deviceID = deviceID
        self.classifier = getattr(cfg, "classifier", None)
        self.batchsize = None if self.classifier is None else 32
        self.adapter_cfg = cfg
        self.adapter = get_adapter(cfg, self.device)
        self.image = None
        self.masks = []
        self.save_button = False
        self.remove_repeating_masks = True

    def segment(self, image: np.ndarray, target_class: Optional[int] = None, text: Optional[str] = None,
                threshold: Optional[float] = 0.5, display: bool = False, use_sliding_window: bool = False) -> list:
        return self.segment_image(image, display=display, use_sliding_window=use_sliding_window, text_prompt=text,
                                  threshold=threshold, target_class=target_class)

    @torch.inference_mode()
    def segment_image(self, image: np.ndarray, display: bool = True, use_sliding_window: bool = False,
                      text_prompt: Optional[str] = None, threshold: Optional[float] = 0.5, target_class: Optional[int] = 1):
        self.target_class = target_class
        if use_sliding_window:
            collected = []
            for (y1, x1, y2, x2) in self.get_sliding_windows(image.shape):
                window = image[y1:y2, x1:x2]
                found = self.adapter.segment_image_2d(window, text_prompt=text_prompt, threshold=threshold)
                local = []
                for m in found:
                    if m["area"] < self.min_mask_area:
                        continue
                    m["offset"] = (y1, x1)                      # segmentation stays window-sized
                    m["bbox"] = self._to_global_bbox(m["bbox"], y1, x1)
                    local.append(m)
                collected.extend(self._apply_classifier(window, local))
            self.masks = self.rasterize_masks(image, collected)
        else:
            self.masks = self.adapter.segment_image_2d(image, text_prompt=text_prompt, threshold=threshold)
            self.masks = self._apply_classifier(image, self.masks)
        # display / save hooks of the reference (matplotlib viewers) are outside the hot-path build
        self.image = image
        return self.masks

    def _apply_classifier(self, image, masks):
        masks = [m for m in masks if m["area"] >= self.min_mask_area]
        if self.remove_repeating_masks:
            masks = utils.remove_duplicate_masks(masks)
        utils.strip_device_rows(masks)
        if self.classifier is None:
            return sorted(masks, key=lambda m: m["area"], reverse=False)
        from saber_amd.filters import masks as filters
        gray = image[:, :, 0] if image.ndim == 3 else image
        # (positional arguments as in the reference, base.py:172-174: self.batchsize lands in apply_classifier's min_mask_area slot and
        # the group size stays at its default of 32)
        return filters.apply_classifier(gray, masks, self.classifier, self.target_class, self.batchsize)

    def get_sliding_windows(self, image_shape: Tuple[int, int]) -> List[Tuple[int, int, int, int]]:
        h, w = image_shape[:2]
        stride = int(self.window_size * (1 - self.overlap_ratio))
        half = self.window_size // 2
        wins = []
        for y in range(0, h, stride):
            for x in range(0, w, stride):
                y2, x2 = min(y + self.window_size, h), min(x + self.window_size, w)
                if (y2 - y) < half or (x2 - x) < half:
                    continue
                wins.append((y, x, y2, x2))
        return wins

    def _to_global_bbox(self, local_bbox, y0, x0):
        x, y, w, h = local_bbox
        return [x + x0, y + y0, w, h]

    def rasterize_masks(self, image, masks):
        H, W = image.shape[:2]
        out = []
        for m in masks:
            y0, x0 = m["offset"]
            seg = m["segmentation"]
            h, w = seg.shape
            full = np.zeros((H, W), dtype=bool)
            ya, xa = max(0, y0), max(0, x0)
            yb, xb = min(H, y0 + h), min(W, x0 + w)
            full[ya:yb, xa:xb] = seg[ya - y0:yb - y0, xa - x0:xb - x0]
            m2 = dict(m)
            m2["segmentation"] = full
            out.append(m2)
        return out


class saber3D(saber2D):
    def __init__(self, deviceID: int = 0, cfg: AdapterConfig = None, amg_cfg: cfgAMG = None, min_mask_area: int = 50):
        super().__init__(deviceID=deviceID, cfg=cfg, amg_cfg=amg_cfg, min_mask_area=min_mask_area)
        self.video_predictor = self.adapter
        self._vol_loaded = False
        self.min_logits = 0.5
        self.confidence_debug = False
        self.nframes = None
        self.filter_threshold = 0.5

    def propagate(self, mask_shape, target_class: Optional[int] = 1):
        """Seed masks into the adapter and propagate bidirectionally (reference :265-280)."""
        arrays = [m["segmentation"] for m in self.masks] if isinstance(self.masks[0], dict) else self.masks
        vol = self.video_predictor.segment_volume(start_frame_idx=self.ann_frame_idx, masks=arrays, vol_shape=mask_shape,
                                                  max_frame_num_to_track=self.nframes, min_presence_score=self.filter_threshold)
        self.video_predictor.reset_state()
        return vol
