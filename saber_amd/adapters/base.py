"""Adapter ABC, configs and factory - the drop-in boundary of the hot path.

Mirrors saber/adapters/base.py of the reference: SAM2AdapterConfig (:7-33), BaseAdapter (:48-89),
get_adapter (:92-97).  Field names, defaults and validation errors are kept so that segmenters and CLI
code written against the reference work unchanged.
"""
from abc import ABC, abstractmethod
from typing import Any, Dict, Iterator, List, Literal, Optional, Tuple, Union

import numpy as np
from pydantic import BaseModel, ConfigDict, Field, field_validator, model_validator

_TRUNKS = {"tiny", "small", "base", "large"}


class SAM2AdapterConfig(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)

    model_type: Literal["sam2"] = "sam2"
    cfg: str = Field("small", description="tiny / small / base / large")
    checkpoint: Optional[str] = None
    num_maskmem: int = 2
    light_modality: bool = False
    amg_cfg: Optional[Any] = None      # cfgAMG instance; None -> cfgAMG() defaults
    min_mask_area: int = 50
    classifier: Optional[Any] = None   # Predictor; fills cfg + amg_cfg from its config['amg_params']

    @model_validator(mode="after")
    def _derive_from_classifier(self) -> "SAM2AdapterConfig":
        if self.classifier is not None and self.amg_cfg is None:
            from saber_amd.adapters.sam2.amg import cfgAMG
            params = self.classifier.config["amg_params"]
            self.cfg = params.get("sam2_cfg", self.cfg)
            self.amg_cfg = cfgAMG(**params)
        return self

    @field_validator("cfg")
    @classmethod
    def _check_cfg(cls, v):
        if v not in _TRUNKS:
            raise ValueError(f"cfg must be one of tiny/small/base/large, got '{v}'")
        return v


class SAM3AdapterConfig(BaseModel):
    """Kept for signature compatibility; the SAM3 adapter is out of scope of this build."""
    model_type: Literal["sam3"] = "sam3"
    checkpoint_path: Optional[str] = None
    load_from_HF: bool = True
    light_modality: bool = False
    text_prompt: Optional[str] = None
    min_mask_area: int = 50


AdapterConfig = Union[SAM2AdapterConfig, SAM3AdapterConfig]


class BaseAdapter(ABC):
    """Common interface every tomogram adapter implements (reference: saber/adapters/base.py:48-89)."""

    frame_metrics: Dict[int, Dict[int, Dict[str, Any]]]

    @abstractmethod
    def segment_image_2d(self, image: np.ndarray, text_prompt: Optional[str] = None) -> List[Dict[str, Any]]:
        """2D segmentation; list of dicts with at least {'segmentation': (H,W) bool, 'area': int}."""

    @abstractmethod
    def set_volume(self, tomogram: np.ndarray, offload_video_to_cpu: bool = False) -> None: ...

    @abstractmethod
    def add_new_mask(self, frame_idx: int, obj_id: int, mask: np.ndarray, inference_state=None) -> Tuple: ...

    @abstractmethod
    def add_new_points_or_box(self, frame_idx: int, obj_id: int, inference_state=None, **kwargs) -> Tuple: ...

    @abstractmethod
    def propagate_in_video(self, start_frame_idx, max_frame_num_to_track=None, reverse=False, inference_state=None) -> Iterator: ...

    @abstractmethod
    def segment_volume(self, start_frame_idx: int, masks=None, vol_shape=None, max_frame_num_to_track=None,
                       min_presence_score: float = 0.5, inference_state=None) -> np.ndarray: ...

    @abstractmethod
    def reset_state(self, inference_state=None) -> None: ...


def get_adapter(config: AdapterConfig, device: str = "cuda") -> BaseAdapter:
    if config.model_type == "sam2":
        from saber_amd.adapters.sam2 import SAM2Adapter
        return SAM2Adapter(config, device)
    raise NotImplementedError("the SAM3 adapter is outside the MI355X hot-path build (SURVEY.md section 2 #19)")
