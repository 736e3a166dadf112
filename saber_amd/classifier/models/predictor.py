"""Predictor (reference: saber/classifier/models/predictor.py:9-249) over the C-ABI's saber_classifier_* entry points.

Same constructor, `config` attribute, predict / batch_predict signatures and return values (an (Nmasks, num_classes) float32 array, rows
of masks below min_area left at zero).  The model behind it is the reference's SAM2Classifier (saber/classifier/models/SAM2.py): the
engine's Hiera encoder as the frozen backbone and the projection / classifier head of the checkpoint; everything from the whole-image
z-score to the softmax runs on the device (csrc/classifier.hip).  There is no CPU path."""
import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch
import yaml

HEAD_PREFIXES = ("projection.", "classifier.")


def load_head_weights(path: str) -> Dict[str, np.ndarray]:
    """state_dict of a trained SAM2Classifier (common.load_model_weights, common.py:73-85: a dict with a "model" entry, or the bare
    state_dict)."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
    return {k: v.detach().float().cpu().numpy() for k, v in sd.items() if k.startswith(HEAD_PREFIXES) and not k.endswith("num_batches_tracked")}


class Predictor:
    def __init__(self, model_config: Optional[str], model_weights: Optional[str], min_area: int = 250, deviceID: int = 0, *,
                 config: Optional[dict] = None, head_weights: Optional[Dict[str, np.ndarray]] = None, engine=None):
        """model_config / model_weights: the reference's yaml + .pth pair.  Keyword-only alternatives for callers that already hold them
        in memory: `config` (the parsed yaml), `head_weights` (name -> array), `engine` (a finalized saber_amd.engine.Engine to use as
        the backbone instead of the shared per-device handle of config['amg_params']['sam2_cfg'])."""
        self.min_area = min_area
        if config is None:
            with open(model_config, "r") as f:
                config = yaml.safe_load(f)
        self.config = config
        self.num_classes = int(config["model"]["num_classes"])
        if engine is None:
            from saber_amd.adapters.sam2.automask import get_engine
            from saber_amd.utils import io
            engine = get_engine(config["amg_params"]["sam2_cfg"], io.get_available_devices(deviceID))
        self.engine = engine
        self.device = engine.device
        if head_weights is None:
            head_weights = load_head_weights(model_weights)
        self.lib = engine.lib
        h = C.c_void_p()
        engine._check(self.lib.saber_classifier_create(engine.h, self.num_classes, C.byref(h)))
        self.h = h
        for name, arr in head_weights.items():
            if name.endswith("num_batches_tracked"):
                continue
            a = np.ascontiguousarray(arr, dtype=np.float32).reshape(arr.shape if np.ndim(arr) else (1,))
            shape = (C.c_int64 * a.ndim)(*a.shape)
            engine._check(self.lib.saber_classifier_set_weight(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        engine._check(self.lib.saber_classifier_finalize(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.saber_classifier_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _to_device(self, image, masks):
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32))
        if isinstance(masks, np.ndarray):
            masks = torch.from_numpy(np.ascontiguousarray(masks.astype(np.uint8, copy=False)))
        image = image.to(self.device, dtype=torch.float32).contiguous()
        masks = (masks.to(self.device) > 0).to(torch.uint8).contiguous() if masks.dtype != torch.uint8 else masks.to(self.device).contiguous()
        if image.ndim != 2 or masks.ndim != 3 or masks.shape[1:] != image.shape:
            raise ValueError(f"Predictor.predict expects an (H,W) image and (N,H,W) masks, got {tuple(image.shape)} and {tuple(masks.shape)}")
        return image, masks

    @torch.inference_mode()
    def predict(self, image, masks) -> np.ndarray:
        image, masks = self._to_device(image, masks)
        n = int(masks.shape[0])
        probs = np.zeros((n, self.num_classes), dtype=np.float32)
        if n == 0:
            return probs
        H, W = image.shape
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self.engine._check(self.lib.saber_classifier_predict(self.h, C.c_void_p(image.data_ptr()), H, W, C.c_void_p(masks.data_ptr()), n,
                                                             int(self.min_area), probs.ctypes.data_as(C.c_void_p), stream))
        return probs

    @torch.inference_mode()
    def batch_predict(self, image, masks, batch_size: int = 32) -> np.ndarray:
        """predictor.py:177-219: the masks in groups of batch_size (each group is one predict call; the whole-image statistics are
        recomputed per group there too)."""
        if isinstance(masks, np.ndarray):
            masks = torch.from_numpy(np.ascontiguousarray(masks.astype(np.uint8, copy=False)))
        total = int(masks.shape[0])
        out = np.zeros((total, self.num_classes), dtype=np.float32)
        for s in range(0, total, batch_size):
            out[s:s + batch_size] = self.predict(image, masks[s:s + batch_size])
        return out

    # ---- test / inspection access
    def head(self, mask_crops: torch.Tensor) -> np.ndarray:
        """SAM2Classifier.forward after the backbone on the embeddings engine slots 0..k-1 hold now; mask_crops (k,320,320) uint8."""
        m = mask_crops.to(self.device, dtype=torch.uint8).contiguous()
        k = int(m.shape[0])
        probs = np.zeros((k, self.num_classes), dtype=np.float32)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self.engine._check(self.lib.saber_classifier_head(self.h, C.c_void_p(m.data_ptr()), k, probs.ctypes.data_as(C.c_void_p), stream))
        return probs

    def last_crops(self, n: int):
        crops = torch.empty((n, 320, 320), dtype=torch.float32, device=self.device)
        cm = torch.empty((n, 320, 320), dtype=torch.uint8, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self.engine._check(self.lib.saber_classifier_get_crops(self.h, n, C.c_void_p(crops.data_ptr()), C.c_void_p(cm.data_ptr()), stream))
        return crops, cm
