"""Per-GPU model factories for the worker pool (reference: saber/segmenters/loaders.py:9-64): GPUPool(init_fn=...) calls one of these
once per GPU and hands the returned dict to every task (saber/entry_points/run_tomogram_segment.py:251-256, run_micrograph_segment.py)."""
import torch

from saber_amd.adapters.base import SAM2AdapterConfig
from saber_amd.adapters.sam2.amg import cfgAMG
from saber_amd.classifier.models import common
from saber_amd.segmenters.micro import cryoMicroSegmenter
from saber_amd.segmenters.tomo import multiDepthTomoSegmenter, tomoSegmenter


def micrograph_workflow(gpu_id: int, cfg: cfgAMG, model_weights: str, model_config: str, target_class: int):
    """Micrograph segmentation models once per GPU (reference loaders.py:9-23): the domain classifier (None when no weights / config are
    given) rides in the adapter config, the AMG configuration is the caller's."""
    torch.cuda.set_device(gpu_id)
    predictor = common.get_predictor(model_weights, model_config, gpu_id)
    adapter_cfg = SAM2AdapterConfig(classifier=predictor, amg_cfg=cfg)
    segmenter = cryoMicroSegmenter(cfg=adapter_cfg, deviceID=gpu_id)
    return {"segmenter": segmenter, "target_class": target_class}


def tomogram_workflow(gpu_id: int, model_weights: str, model_config: str, target_class: int, num_slabs: int):
    """Tomogram segmentation models once per GPU (reference loaders.py:25-45): several slabs -> multiDepthTomoSegmenter."""
    torch.cuda.set_device(gpu_id)
    predictor = common.get_predictor(model_weights, model_config, gpu_id)
    cfg_obj = SAM2AdapterConfig(classifier=predictor)
    if num_slabs > 1:
        segmenter = multiDepthTomoSegmenter(cfg=cfg_obj, deviceID=gpu_id, target_class=target_class)
    else:
        segmenter = tomoSegmenter(cfg=cfg_obj, deviceID=gpu_id)
    return {"predictor": predictor, "segmenter": segmenter, "target_class": target_class}


def base_microsegmenter(gpu_id: int, cfg: cfgAMG):
    torch.cuda.set_device(gpu_id)
    return {"segmenter": cryoMicroSegmenter(amg_cfg=cfg, deviceID=gpu_id)}


def base_tomosegmenter(gpu_id: int, cfg: cfgAMG):
    torch.cuda.set_device(gpu_id)
    return {"segmenter": tomoSegmenter(amg_cfg=cfg, deviceID=gpu_id)}
