"""Per-GPU model factories for the worker pool (reference: saber/segmenters/loaders.py:47-64)."""
import torch

from saber_amd.adapters.sam2.amg import cfgAMG
from saber_amd.segmenters.micro import cryoMicroSegmenter
from saber_amd.segmenters.tomo import tomoSegmenter


def base_microsegmenter(gpu_id: int, cfg: cfgAMG):
    torch.cuda.set_device(gpu_id)
    return {"segmenter": cryoMicroSegmenter(amg_cfg=cfg, deviceID=gpu_id)}


def base_tomosegmenter(gpu_id: int, cfg: cfgAMG):
    torch.cuda.set_device(gpu_id)
    return {"segmenter": tomoSegmenter(amg_cfg=cfg, deviceID=gpu_id)}
