"""SAM2Adapter on the MI355X engine (reference: saber/adapters/sam2/predictor.py).

Hot-path method: segment_image_2d (:48-70) = prep.prepare -> lazily built mask generator -> generate.
The video-propagation methods of the ABC (set_volume / add_new_mask / propagate_in_video / segment_volume,
:76-348) belong to the memory-attention path, a "next" row of the scope table; they raise
NotImplementedError naming that, instead of silently doing something else."""
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from saber_amd.adapters.base import BaseAdapter, SAM2AdapterConfig
from saber_amd.adapters.sam2.amg import cfgAMG
from saber_amd.adapters.sam2.automask import build_amg
from saber_amd.utils import preprocessing as prep

_NEXT = "SAM2 video propagation (memory attention) is a 'next' row of the hot-path scope (SURVEY.md 8f-1), not built yet"


class SAM2Adapter(BaseAdapter):
    def __init__(self, config: SAM2AdapterConfig, device="cuda"):
        if config.num_maskmem > 7:
            raise ValueError("num_maskmem must be less than 7")
        self._config = config
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        # the reference builds its video predictor from config.cfg here; that model belongs to the propagation path
        # (next row).  The AMG model is chosen by amg_cfg.sam2_cfg (automask.py:61) and is built lazily below.
        self.frame_metrics: Dict[int, Dict[int, Dict[str, Any]]] = {}
        self._vol_shape: Optional[Tuple[int, int, int]] = None
        self.inference_state = None
        self._mask_generator = None

    def _generator(self):
        if self._mask_generator is None:
            if self._config.amg_cfg is not None:
                amg = self._config.amg_cfg.dict()
            else:
                amg = cfgAMG(sam2_cfg=self._config.cfg).dict()
            self._mask_generator = build_amg(amg, self._config.min_mask_area, device=self.device, checkpoint=self._config.checkpoint)
        return self._mask_generator

    @property
    def engine(self):
        return self._generator().base_generator.engine

    @torch.inference_mode()
    def segment_image_2d(self, image: np.ndarray, text_prompt: str = None, threshold: float = None) -> List[Dict[str, Any]]:
        """(H,W) gray or (H,W,3) image of any float range -> SAM-AMG dict list (caller owns the arrays)."""
        gen = self._generator()
        if image.ndim not in (2, 3) or (image.ndim == 3 and image.shape[2] != 3):
            raise ValueError(f"segment_image_2d: expected (H,W) or (H,W,3), got {image.shape}")
        # (H,W): gray slice, the reference repeats it to 3 channels (to_rgb=True).  (H,W,3): the reference passes the array
        # through prepare(to_rgb=False): box filter over all three axes, one global min/max (predictor.py:58-59)
        img = prep.prepare(image, to_rgb=image.ndim == 2, engine=gen.base_generator.engine)
        return gen.generate(img)

    # ------------------------------------------------------------------ video path: next row
    def set_volume(self, tomogram: np.ndarray, offload_video_to_cpu: bool = False) -> None:
        raise NotImplementedError(_NEXT)

    def add_new_mask(self, frame_idx: int, obj_id: int, mask: np.ndarray, inference_state=None) -> Tuple:
        raise NotImplementedError(_NEXT)

    def add_new_points_or_box(self, frame_idx: int, obj_id: int, inference_state=None, **kwargs) -> Tuple:
        raise NotImplementedError(_NEXT)

    def propagate_in_video(self, start_frame_idx, max_frame_num_to_track=None, reverse=False, inference_state=None) -> Iterator:
        raise NotImplementedError(_NEXT)

    def segment_volume(self, start_frame_idx: int, masks=None, vol_shape=None, max_frame_num_to_track=None,
                       min_presence_score: float = 0.5, inference_state=None) -> np.ndarray:
        if self.inference_state is None:
            raise RuntimeError("call set_volume() first")
        raise NotImplementedError(_NEXT)

    def reset_state(self, inference_state=None) -> None:
        self.inference_state = None
        self.frame_metrics = {}
