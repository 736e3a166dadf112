"""cfgAMG and the post-generation mask filters (reference: saber/adapters/sam2/amg.py).

cfgAMG field names / defaults / validation follow amg.py:4-37; MaskFilteringUtils and
FilteredSAM2MaskGenerator follow :39-201 (area, relative-box-size and score filters applied after generate()).
"""
from typing import Any, Dict, List, Optional

import numpy as np
from pydantic import BaseModel, Field, field_validator


class cfgAMG(BaseModel):
    """Configuration of the SAM2 automatic mask generator."""
    npoints: int = Field(gt=0, default=32, description="Number of points to sample")
    points_per_batch: int = Field(gt=0, default=64)
    pred_iou_thresh: float = Field(gt=0, le=1.0, default=0.7)
    stability_score_thresh: float = Field(ge=0, le=1.0, default=0.92)
    stability_score_offset: float = Field(default=0.7)
    crop_n_layers: int = Field(ge=0, default=2)
    box_nms_thresh: float = Field(gt=0, le=1.0, default=0.7)
    crop_n_points_downscale_factor: int = Field(gt=0, default=2)
    use_m2m: bool = Field(default=True)
    multimask_output: bool = Field(default=True)
    sam2_cfg: str = Field(default="small")

    @field_validator("sam2_cfg")
    @classmethod
    def validate_sam2_cfg(cls, v: str) -> str:
        valid = ["tiny", "small", "base", "large"]
        if v not in valid:
            raise ValueError(f"sam2_cfg must be one of {valid}, got {v}")
        return v

    def dict(self, *args: Any, **kwargs: Any) -> Dict[str, Any]:
        return self.model_dump(*args, **kwargs)

    def to_dict(self, *args: Any, **kwargs: Any) -> Dict[str, Any]:
        return self.dict(*args, **kwargs)


class MaskFilteringUtils:
    """Filters over SAM-AMG annotation lists."""

    @staticmethod
    def filter_masks_by_relative_box_size(mask_annotations, max_rel_box_size=None, min_rel_box_size=None,
                                          image_height=None, image_width=None):
        if max_rel_box_size is None and min_rel_box_size is None:
            return mask_annotations
        if image_height is None or image_width is None:
            raise ValueError("image_height and image_width must be provided for relative size filtering")
        out = []
        for ann in mask_annotations:
            bbox = ann.get("bbox", None)
            if bbox is None:
                continue
            _, _, w, h = bbox
            rw, rh = w / image_width, h / image_height
            ok = True
            if max_rel_box_size is not None:
                ok = ok and rw < max_rel_box_size and rh < max_rel_box_size
            if min_rel_box_size is not None:
                ok = ok and rw > min_rel_box_size and rh > min_rel_box_size
            if ok:
                out.append(ann)
        return out

    @staticmethod
    def filter_masks_by_area(mask_annotations, min_area=None, max_area=None):
        if min_area is None and max_area is None:
            return mask_annotations
        out = []
        for ann in mask_annotations:
            area = ann.get("area", 0)
            if min_area is not None and area < min_area:
                continue
            if max_area is not None and area > max_area:
                continue
            out.append(ann)
        return out

    @staticmethod
    def filter_masks_by_score(mask_annotations, min_predicted_iou=None, min_stability_score=None):
        out = []
        for ann in mask_annotations:
            if min_predicted_iou is not None and not ann.get("predicted_iou", 0.0) >= min_predicted_iou:
                continue
            if min_stability_score is not None and not ann.get("stability_score", 0.0) >= min_stability_score:
                continue
            out.append(ann)
        return out


class FilteredSAM2MaskGenerator:
    """Wraps a mask generator (anything with .generate(image)) with the post filters."""

    def __init__(self, base_generator, min_rel_box_size=None, max_rel_box_size=None, min_area_filter=None, max_area_filter=None):
        self.base_generator = base_generator
        self.max_rel_box_size = max_rel_box_size
        self.min_rel_box_size = min_rel_box_size
        self.min_area_filter = min_area_filter
        self.max_area_filter = max_area_filter
        self.filter_utils = MaskFilteringUtils()

    def generate(self, image: np.ndarray) -> List[Dict[str, Any]]:
        anns = self.base_generator.generate(image)
        h, w = image.shape[:2]
        if self.max_rel_box_size is not None or self.min_rel_box_size is not None:
            anns = self.filter_utils.filter_masks_by_relative_box_size(
                anns, max_rel_box_size=self.max_rel_box_size, min_rel_box_size=self.min_rel_box_size, image_height=h, image_width=w)
        if self.min_area_filter is not None or self.max_area_filter is not None:
            anns = self.filter_utils.filter_masks_by_area(anns, min_area=self.min_area_filter, max_area=self.max_area_filter)
        return anns

    def set_filters(self, min_rel_box_size=None, max_rel_box_size=None, min_area_filter=None):
        if min_rel_box_size is not None:
            self.min_rel_box_size = min_rel_box_size
        if max_rel_box_size is not None:
            self.max_rel_box_size = max_rel_box_size
        if min_area_filter is not None:
            self.min_area_filter = min_area_filter

    def __getattr__(self, name):
        return getattr(self.base_generator, name)
