"""Mask-generator construction (reference: saber/adapters/sam2/automask.py).

build_amg (:49-86) returns FilteredSAM2MaskGenerator(base=<generator>, min_area_filter=min_mask_area) where the
base generator is built with SABER's AMG parameters (:66-78).  Here the base generator is EngineMaskGenerator:
`generate(image)` calls saber_amg_generate and unpacks the bit-packed masks into the SAM-AMG dict schema at the
API edge."""
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from saber_amd import pretrained_weights
from saber_amd.adapters.sam2 import amg as fmask

_ENGINES: Dict[Any, Any] = {}


def default_precision() -> str:
    """The adapter's production arithmetic.  Default (round 5): "fp16" - the production kernels on IEEE half operands with fp32 accumulation
    (10 mantissa bits = the TF32 arithmetic the reference enables on its GPUs, saber/utils/io.py:127-130; include/saber_amd.h:
    SABER_PRECISION_FP16): ~1e-3 from the fp32 oracle end to end against bf16's 5-7e-3, at the same speed, with the engine's run-time overflow
    sentinel (SaberRangeError instead of NaN masks if an activation leaves the range of half).  SABER_AMD_PRECISION=bf16 selects the arithmetic
    of BASELINE configs[1] (bench.py's headline); read ONCE per model, when its first handle is built."""
    import os
    precision = os.environ.get("SABER_AMD_PRECISION", "fp16")
    if precision not in ("bf16", "fp16"):
        raise ValueError(f"SABER_AMD_PRECISION must be 'bf16' or 'fp16', got '{precision}'")
    return precision


def get_engine(sam2_cfg: str, device, checkpoint: Optional[str] = None, max_images: int = 21, max_prompts: int = 1024, replica: int = 0,
               precision: Optional[str] = None):
    """One engine per (device, trunk, weights): the reference builds a second SAM2 copy for AMG;
    here adapter and generator share one handle.  replica > 0: further handles of the same model on the same device (the z-loop keeps
    two slices in flight per GPU, one handle per thread)."""
    from saber_amd.engine import Engine
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    if dev.type != "cuda":
        raise RuntimeError(f"the MI355X engine needs a ROCm device, got '{dev}' (there is no CPU fallback)")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    src = pretrained_weights.resolve_weights(sam2_cfg, checkpoint)
    if precision is None:
        precision = default_precision()
    if precision not in ("bf16", "fp16"):
        raise ValueError(f"precision must be 'bf16' or 'fp16', got '{precision}'")
    key = (idx, sam2_cfg, tuple(sorted(src.items())), replica, precision)
    if key not in _ENGINES:
        if src.pop("fitted_decoder", None):               # SABER_AMD_SEEDED_WEIGHTS=fitted (tests / bench): seeded encoder + fitted mask decoder
            from saber_amd.model_config import get_config
            from saber_amd.weights import fitted_decoder_weights
            src = {"weights": fitted_decoder_weights(get_config(sam2_cfg), src.get("seed", 0))}
        eng = Engine(sam2_cfg, device=idx, max_images=max_images, max_prompts=max_prompts, precision=precision, **src)
        eng._build = (sam2_cfg, checkpoint)       # what get_replica() needs to build an identical handle
        _ENGINES[key] = eng
    return _ENGINES[key]


def get_replica(engine, replica: int):
    """A further handle of the SAME model (trunk and weights) as `engine` on the same device.  The z-loop keeps several slices in
    flight per GPU; every handle must segment with the model slice 0 was segmented with, whatever the adapter's `cfg` field says
    (the AMG model is chosen by amg_cfg.sam2_cfg, reference automask.py:61)."""
    sam2_cfg, checkpoint = getattr(engine, "_build", (engine.cfg.name, None))
    # the replica takes the 16-bit operand type OF THE HANDLE IT REPLICATES, not whatever the environment says by now
    return get_engine(sam2_cfg, engine.device, checkpoint, max_images=engine.max_images, max_prompts=engine.max_prompts, replica=replica,
                      precision=engine.operands)


def get_default() -> Dict[str, Any]:
    """Default AMG parameters of the reference (automask.py:29-46)."""
    return {"npoints": 32, "points_per_batch": 64, "pred_iou_thresh": 0.7, "stability_score_thresh": 0.92,
            "stability_score_offset": 0.7, "crop_n_layers": 2, "box_nms_thresh": 0.7, "crop_n_points_downscale_factor": 2,
            "use_m2m": True, "multimask_output": True}


class EngineMaskGenerator:
    """generate(image) with the contract of sam2.SAM2AutomaticMaskGenerator.generate."""

    def __init__(self, engine, amg_params: Dict[str, Any], max_masks: int = 2048):
        from saber_amd.engine import make_amg_params
        self.engine = engine
        self.params = make_amg_params(amg_params)
        self.max_masks = max_masks

    @torch.inference_mode()
    def generate_device(self, image):
        """image: (H,W)/(H,W,3) float32 in [0,1], numpy or device tensor -> (bits, meta) on the device."""
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).to(self.engine.device)
        return self.engine.amg_generate(image.contiguous(), self.params, max_masks=self.max_masks)

    def generate(self, image) -> List[Dict[str, Any]]:
        from saber_amd.engine import unpack_bits
        from saber_amd.segmenters.utils import DEVICE_ROW_KEY, DeviceMaskRows
        H, W = image.shape[:2]
        bits, meta = self.generate_device(image)
        masks = unpack_bits(bits, W) if len(meta) else []
        rows = DeviceMaskRows(self.engine, bits, H, W)
        return [{"segmentation": masks[i], DEVICE_ROW_KEY: (rows, i), "area": int(m.area), "bbox": [float(v) for v in m.bbox_xywh],
                 "predicted_iou": float(m.predicted_iou), "point_coords": [[float(m.point_xy[0]), float(m.point_xy[1])]],
                 "stability_score": float(m.stability_score), "crop_box": [float(v) for v in m.crop_box_xywh]}
                for i, m in enumerate(meta)]


def build_amg(amg_params: Dict[str, Any], min_mask_area: int, device="cuda", checkpoint: Optional[str] = None):
    engine = get_engine(amg_params["sam2_cfg"], device, checkpoint)
    return fmask.FilteredSAM2MaskGenerator(base_generator=EngineMaskGenerator(engine, amg_params), min_area_filter=min_mask_area)


def amg_cli():
    """click options of the reference's amg_cli decorator (automask.py:9-24), same flags and defaults."""
    import click

    def decorator(f):
        f = click.option("-cfg", "--sam2-cfg", required=False, default="small", help="SAM2 Model Config",
                         type=click.Choice(["large", "base", "small", "tiny"], case_sensitive=False))(f)
        f = click.option("-npts", "--npoints", type=int, default=32, help="Number of points per side")(f)
        f = click.option("-nbatch", "--points-per-batch", type=int, default=64, help="Number of points per batch")(f)
        f = click.option("-iou", "--pred-iou-thresh", type=float, default=0.7, help="Prediction IOU threshold")(f)
        f = click.option("-nlayers", "--crop-n-layers", type=int, default=2, help="Number of crop layers")(f)
        f = click.option("-box", "--box-nms-thresh", type=float, default=0.7, help="Box NMS threshold")(f)
        f = click.option("-crop", "--crop-n-points", type=int, default=2, help="Crop N Points Downscale Factor")(f)
        f = click.option("-m2m", "--use-m2m", type=bool, default=True, help="Use M2M")(f)
        f = click.option("-multi", "--multimask", type=bool, default=True, help="Multimask Output")(f)
        return f
    return decorator
