"""Python host wrapper of the C-ABI engine: PyTorch-ROCm tensors in, masks out.

PyTorch is used only for device memory, streams and (elsewhere) torch.distributed; every
arithmetic step of the hot path runs in the HIP library.  Errors from the C-ABI surface as
Python exceptions (ValueError for bad configuration, RuntimeError for state / HIP errors) so
that the reference's per-task accounting (saber/utils/parallelization.py:129-135) keeps working.
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .model_config import get_config
from .weights import seeded_weights, load_checkpoint

SABER_U16, SABER_F32 = 0, 1


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    """One engine handle bound to one device (reference threading contract: one caller per device)."""

    def __init__(self, trunk: str = "large", device: int = 0, weights: Optional[Dict[str, np.ndarray]] = None,
                 checkpoint: Optional[str] = None, seed: int = 0, max_images: int = 1, max_prompts: int = 64, weight_format: str = "bf16", precision: str = "bf16",
                 operands: Optional[str] = None):
        self.lib = _lib.load()
        self.cfg = get_config(trunk)  # ValueError for unknown names, like the reference
        if not torch.cuda.is_available():
            raise RuntimeError("saber_amd.Engine needs a ROCm device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device(f"cuda:{device}")
        self.device_index = device
        h = C.c_void_p()
        st = self.lib.saber_engine_create(device, trunk.encode(), max_images, max_prompts, C.byref(h))
        if st != 0:
            msg = self.lib.saber_last_error(None).decode()
            raise (ValueError if st == -1 else RuntimeError)(msg)
        self.h = h
        self.max_images, self.max_prompts = max_images, max_prompts
        if weight_format not in ("bf16", "fp8", "mxfp8"):
            raise ValueError(f"weight_format must be 'bf16', 'fp8' or 'mxfp8', got '{weight_format}'")
        self.weight_format = weight_format
        if weight_format != "bf16":      # fp8: e4m3 storage of the stage-2/3 block weights; mxfp8: MX operands on the fp8 MFMA (include/saber_amd.h: saber_engine_set_weight_format)
            self._check(self.lib.saber_engine_set_weight_format(self.h, {"fp8": 1, "mxfp8": 2}[weight_format]))
        if precision not in ("bf16", "fp16", "exact"):
            raise ValueError(f"precision must be 'bf16', 'fp16' or 'exact', got '{precision}'")
        # operands: the 16-bit MFMA operand type the weights are converted to at finalize ("bf16" | "fp16"; include/saber_amd.h:
        # SABER_PRECISION_FP16).  Implied by `precision`; only a handle created with precision="exact" needs it spelled out, to say which
        # 16-bit arithmetic set_precision can switch back to.
        if operands is None:
            operands = "fp16" if precision == "fp16" else "bf16"
        if operands not in ("bf16", "fp16") or (precision in ("bf16", "fp16") and operands != precision):
            raise ValueError(f"operands must be 'bf16' or 'fp16' and agree with precision, got operands='{operands}' precision='{precision}'")
        self.operands = operands
        self.precision = precision
        self.has_exact = precision == "exact"      # fp32 weight copies are kept only when the handle was created in the exact mode
        if operands == "fp16":
            self._check(self.lib.saber_engine_set_precision(self.h, 2))
        if precision == "exact":         # fp32 operands everywhere (include/saber_amd.h: saber_engine_set_precision); keeps fp32 weight copies
            self._check(self.lib.saber_engine_set_precision(self.h, 1))
        if weights is None:
            weights = load_checkpoint(checkpoint, self.cfg) if checkpoint else seeded_weights(self.cfg, seed)
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.saber_engine_set_weight(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        self._check(self.lib.saber_engine_finalize(self.h))

    @classmethod
    def bare(cls, device: int = 0) -> "Engine":
        """A handle without model weights: enough for the volume post-processing calls (separate_masks, smooth_labels,
        gaussian_smoothing_3d), which only need the device binding and the per-handle error string."""
        self = cls.__new__(cls)
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("saber_amd.Engine needs a ROCm device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device(f"cuda:{device}")
        self.device_index = device
        h = C.c_void_p()
        st = self.lib.saber_engine_create(device, b"tiny", 1, 1, C.byref(h))
        if st != 0:
            raise RuntimeError(self.lib.saber_last_error(None).decode())
        self.h = h
        self.cfg = None
        self.max_images = self.max_prompts = 0
        return self

    def _check(self, st: int):
        if st != 0:
            msg = self.lib.saber_last_error(self.h).decode()
            raise (ValueError if st == -1 else _lib.SaberRangeError if st == _lib.SABER_ERR_RANGE else RuntimeError)(f"saber_amd: {msg}")

    def check_finite(self):
        """Overflow sentinel (include/saber_amd.h: saber_engine_check_finite): waits for the current stream and raises SaberRangeError when an
        encode / decode since the last check produced NaN / inf (fp16 operands: an activation beyond 65 504).  amg_generate checks by itself;
        call this after encode / decode_points where the host synchronises anyway."""
        self._check(self.lib.saber_engine_check_finite(self.h, _stream()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.saber_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ K0
    def prepare(self, img: torch.Tensor) -> torch.Tensor:
        """prep.prepare on device.  img: (H,W) uint16 or float32 device tensor -> (H,W) float32 in [0,1];
        (H,W,3) float32 -> (H,W,3) float32 (the reference's treatment of an RGB array: box filter over all three axes, one min/max)."""
        assert img.is_cuda and img.is_contiguous()
        if img.dim() == 3:
            if img.shape[2] != 3 or img.dtype != torch.float32:
                raise ValueError(f"prepare: a 3-D input must be (H,W,3) float32, got {tuple(img.shape)} {img.dtype}")
            out = torch.empty(img.shape, dtype=torch.float32, device=img.device)
            self._check(self.lib.saber_prepare_rgb(self.h, _ptr(img), img.shape[0], img.shape[1], _ptr(out), _stream()))
            return out
        assert img.dim() == 2
        if img.dtype == torch.uint16:
            dt = SABER_U16
        elif img.dtype == torch.float32:
            dt = SABER_F32
        else:
            raise ValueError(f"prepare: unsupported dtype {img.dtype}")
        out = torch.empty(img.shape, dtype=torch.float32, device=img.device)
        self._check(self.lib.saber_prepare(self.h, _ptr(img), dt, img.shape[0], img.shape[1], _ptr(out), _stream()))
        return out

    # ------------------------------------------------------------------ encoder
    def encode(self, img: torch.Tensor, crop_boxes: Optional[Sequence[Sequence[int]]] = None, slot0: int = 0, normalised_grey: bool = False):
        """img: (H,W) or (H,W,3) float32 in [0,1] on the device; crop_boxes: list of [x0,y0,x1,y1].  normalised_grey: (H,W) plane that
        already carries the model's input normalisation (channels = -1 of saber_encode)."""
        assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
        H, W = img.shape[:2]
        ch = 1 if img.dim() == 2 else img.shape[2]
        if normalised_grey:
            assert img.dim() == 2
            ch = -1
        if crop_boxes is None:
            crop_boxes = [[0, 0, W, H]]
        cb = np.ascontiguousarray(np.asarray(crop_boxes, dtype=np.int32).reshape(-1, 4))
        self._check(self.lib.saber_encode(self.h, _ptr(img), H, W, ch, cb.ctypes.data_as(C.POINTER(C.c_int)), len(cb), slot0, _stream()))

    def get_features(self, slot: int = 0):
        emb = torch.empty((256, 64, 64), dtype=torch.float32, device=self.device)
        s0 = torch.empty((32, 256, 256), dtype=torch.float32, device=self.device)
        s1 = torch.empty((64, 128, 128), dtype=torch.float32, device=self.device)
        self._check(self.lib.saber_get_features(self.h, slot, _ptr(emb), _ptr(s0), _ptr(s1), _stream()))
        return {"image_embed": emb, "feat_s0": s0, "feat_s1": s1}

    # ------------------------------------------------------------------ decoder
    def decode_points(self, pts: torch.Tensor, slot: int = 0, multimask: bool = True, mask_input: Optional[torch.Tensor] = None,
                      labels: Optional[torch.Tensor] = None):
        """pts: (n,2) float32 model-pixel coords on device. Returns (lowres (n,M,256,256), iou (n,M), obj (n,))."""
        assert pts.is_cuda and pts.dtype == torch.float32 and pts.is_contiguous()
        n = pts.shape[0]
        M = 3 if multimask else 1
        low = torch.empty((n, M, 256, 256), dtype=torch.float32, device=self.device)
        iou = torch.empty((n, M), dtype=torch.float32, device=self.device)
        obj = torch.empty((n,), dtype=torch.float32, device=self.device)
        if mask_input is not None:
            assert mask_input.is_cuda and mask_input.dtype == torch.float32 and mask_input.is_contiguous() and mask_input.numel() == n * 65536
        if labels is not None:
            assert labels.is_cuda and labels.dtype == torch.int32 and labels.numel() == n
        self._check(self.lib.saber_decode_points(self.h, slot, _ptr(pts), _ptr(labels), n, int(multimask), _ptr(mask_input),
                                                 _ptr(low), _ptr(iou), _ptr(obj), _stream()))
        return low, iou, obj

    def decode_prompts(self, pts: torch.Tensor, labels: torch.Tensor, slot: int = 0, multimask: bool = False, mask_input: Optional[torch.Tensor] = None):
        """Prompts of several points each (clicks with labels 1 / 0; a box = its two corners with labels 2 / 3 first; -1 = not a point).
        pts: (n,k,2) float32 model-pixel coords, labels: (n,k) int32, both on the device.  With k > 1 the handle must be in the exact
        precision mode (the bf16 decoder kernels carry 8 tokens per prompt).  Returns (lowres (n,M,256,256), iou (n,M), obj (n,))."""
        assert pts.is_cuda and pts.dtype == torch.float32 and pts.is_contiguous() and pts.dim() == 3 and pts.shape[2] == 2
        n, k = pts.shape[:2]
        assert labels.is_cuda and labels.dtype == torch.int32 and labels.is_contiguous() and tuple(labels.shape) == (n, k)
        M = 3 if multimask else 1
        low = torch.empty((n, M, 256, 256), dtype=torch.float32, device=self.device)
        iou = torch.empty((n, M), dtype=torch.float32, device=self.device)
        obj = torch.empty((n,), dtype=torch.float32, device=self.device)
        if mask_input is not None:
            assert mask_input.is_cuda and mask_input.dtype == torch.float32 and mask_input.is_contiguous() and mask_input.numel() == n * 65536
        self._check(self.lib.saber_decode_prompts(self.h, slot, _ptr(pts), _ptr(labels), n, k, int(multimask), _ptr(mask_input),
                                                  _ptr(low), _ptr(iou), _ptr(obj), _stream()))
        return low, iou, obj

    # ------------------------------------------------------------------ AMG
    def amg_generate(self, img: torch.Tensor, params: "_lib.AmgParams", max_masks: int = 1024):
        """img: (H,W) or (H,W,3) float32 in [0,1].  Returns (bits uint32 (n,H,W32) device tensor, list of MaskMeta)."""
        assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
        H, W = img.shape[:2]
        ch = 1 if img.dim() == 2 else img.shape[2]
        W32 = (W + 31) // 32
        bits = torch.empty((max_masks, H, W32), dtype=torch.int32, device=self.device)
        meta = (_lib.MaskMeta * max_masks)()
        cnt = C.c_int(0)
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream == 0:
            # hipGraph capture is not possible on the legacy default stream: run on a stream of this handle, ordered after / before it
            side = getattr(self, "_side_stream", None)
            if side is None:
                side = self._side_stream = torch.cuda.Stream(self.device)
            side.wait_stream(cur)
            st = self.lib.saber_amg_generate(self.h, _ptr(img), H, W, ch, C.byref(params), _ptr(bits), max_masks, meta, C.byref(cnt), C.c_void_p(side.cuda_stream))
            cur.wait_stream(side)
        else:
            st = self.lib.saber_amg_generate(self.h, _ptr(img), H, W, ch, C.byref(params), _ptr(bits), max_masks, meta, C.byref(cnt), _stream())
        if st == _lib.SABER_ERR_CAPACITY and cnt.value > max_masks:
            # the engine reports the count it needed: retry once with that capacity instead of failing the slice
            return self.amg_generate(img, params, max_masks=cnt.value)
        self._check(st)
        n = cnt.value
        return bits[:n], [meta[i] for i in range(n)]

    def label_plane(self, bits: torch.Tensor, order: Sequence[int], H: int, W: int) -> torch.Tensor:
        plane = torch.empty((H, W), dtype=torch.uint16, device=self.device)
        n = len(order)
        arr = (C.c_int * max(n, 1))(*order)
        self._check(self.lib.saber_label_plane(self.h, _ptr(bits) if n else None, arr, n, H, W, _ptr(plane), _stream()))
        return plane

    def pair_intersections(self, bits: torch.Tensor, H: int, W: int) -> torch.Tensor:
        n = bits.shape[0]
        inter = torch.empty((n, n), dtype=torch.int32, device=self.device)
        self._check(self.lib.saber_mask_pair_intersections(self.h, _ptr(bits) if n else None, n, H, W, _ptr(inter) if n else None, _stream()))
        return inter

    def separate_masks(self, planes: torch.Tensor, min_mask_area: int = 100):
        """utils.separate_masks on the device.  planes: (Z,H,W) uint16 / int16 device tensor of per-slice label planes.
        Returns ((Z,H,W) int32 device tensor holding the uint32 labels, number of labels)."""
        assert planes.is_cuda and planes.dim() == 3 and planes.is_contiguous() and planes.dtype in (torch.uint16, torch.int16)
        Z, H, W = planes.shape
        out = torch.empty((Z, H, W), dtype=torch.int32, device=planes.device)
        n = C.c_int(0)
        self._check(self.lib.saber_separate_masks(self.h, _ptr(planes), Z, H, W, int(min_mask_area), _ptr(out), C.byref(n), _stream()))
        return out, n.value

    def smooth_labels(self, labels: torch.Tensor, scale: float = 0.075):
        """filters.masks.fast_3d_gaussian_smoothing on the device.  labels: (Z,H,W) device tensor of uint8 / (u)int16 / (u)int32 label
        values (non-negative).  Returns ((Z,H,W) uint8 device tensor, number of labels found)."""
        assert labels.is_cuda and labels.dim() == 3 and labels.is_contiguous()
        if labels.dtype not in (torch.uint8, torch.int16, torch.uint16, torch.int32, torch.uint32):
            raise ValueError(f"smooth_labels: unsupported dtype {labels.dtype}")
        Z, H, W = labels.shape
        out = torch.empty((Z, H, W), dtype=torch.uint8, device=labels.device)
        n = C.c_int(0)
        self._check(self.lib.saber_smooth_labels(self.h, _ptr(labels), labels.element_size(), Z, H, W, float(scale), _ptr(out), C.byref(n), _stream()))
        return out, n.value

    def gaussian_smoothing_3d(self, mask: torch.Tensor, sigma: float) -> torch.Tensor:
        """filters.gaussian.gaussian_smoothing_3d on the device.  mask: (Z,H,W) bool / uint8 0-1 device tensor -> float32 field."""
        assert mask.is_cuda and mask.dim() == 3 and mask.is_contiguous() and mask.dtype in (torch.bool, torch.uint8)
        Z, H, W = mask.shape
        out = torch.empty((Z, H, W), dtype=torch.float32, device=mask.device)
        self._check(self.lib.saber_gaussian_smoothing_3d(self.h, _ptr(mask), Z, H, W, float(sigma), _ptr(out), _stream()))
        return out

    def set_precision(self, precision: str):
        """Switch between the handle's 16-bit production arithmetic ("bf16" or "fp16": whichever its weights were converted to) and the
        fp32 exact mode (only on a handle created with precision="exact", which keeps the fp32 weight copies)."""
        if precision not in ("bf16", "fp16", "exact"):
            raise ValueError(f"precision must be 'bf16', 'fp16' or 'exact', got '{precision}'")
        self._check(self.lib.saber_engine_set_precision(self.h, {"bf16": 0, "exact": 1, "fp16": 2}[precision]))
        self.precision = precision

    def set_encoder_stream(self, stream_ptr):
        """The mask generator's encoder passes on another HIP stream (include/saber_amd.h: saber_engine_set_encoder_stream); None = default."""
        self._check(self.lib.saber_engine_set_encoder_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def set_device_amg(self, enable: bool):
        """Filters / NMS / compaction of the mask generator on the device (default) or on the host (include/saber_amd.h: saber_engine_set_device_amg)."""
        self._check(self.lib.saber_engine_set_device_amg(self.h, int(bool(enable))))

    def set_iou_pruning(self, enable: bool):
        """IoU pruning of the AMG m2m pass (include/saber_amd.h: saber_engine_set_iou_pruning); on by default, results identical."""
        self._check(self.lib.saber_engine_set_iou_pruning(self.h, int(bool(enable))))

    def last_pruning(self):
        """(m2m candidates skipped, m2m candidates) of the last amg_generate call"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.saber_amg_last_pruning(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_graphs(self, enable: bool):
        """hipGraph replay of the AMG launch sequences (include/saber_amd.h: saber_engine_set_graphs)."""
        self._check(self.lib.saber_engine_set_graphs(self.h, int(bool(enable))))

    def graph_stats(self):
        """(sequences captured, sequences replayed) on this handle."""
        import ctypes as C
        c, r = C.c_int(0), C.c_int(0)
        self._check(self.lib.saber_engine_graph_stats(self.h, C.byref(c), C.byref(r)))
        return c.value, r.value

    def profile_begin(self):
        self._check(self.lib.saber_profile_begin(self.h))

    def profile_end(self) -> dict:
        n = len(_lib.PROFILE_CLASSES)
        arr = (_lib.ProfileClass * n)()
        self._check(self.lib.saber_profile_end(self.h, arr, n))
        return {name: {"launches": int(arr[i].launches), "ms": float(arr[i].ms), "flops": float(arr[i].flops), "bytes": float(arr[i].bytes)}
                for i, name in enumerate(_lib.PROFILE_CLASSES)}

    def encoder_flops(self) -> float:
        return float(self.lib.saber_encoder_flops(self.h))


def unpack_bits(bits: torch.Tensor, W: int) -> np.ndarray:
    """(n,H,W32) int32 device tensor -> (n,H,W) bool numpy (bit b of word w = pixel 32w+b).  Unpacked on the device
    (saber_k_unpack_masks) and copied through a pinned buffer; the caller owns the returned array."""
    n, H = bits.shape[:2]
    if n == 0:
        return np.zeros((0, H, W), dtype=bool)
    assert bits.is_cuda and bits.is_contiguous() and bits.shape[2] == (W + 31) // 32
    lib = _lib.load()
    out = np.empty((n, H, W), dtype=np.uint8)
    step = max(1, (64 << 20) // (H * W))                      # masks per pass: 64 MiB of device + pinned staging
    dev = torch.empty((min(step, n), H, W), dtype=torch.uint8, device=bits.device)
    pin = torch.empty((min(step, n), H, W), dtype=torch.uint8, pin_memory=True)
    for i0 in range(0, n, step):
        k = min(step, n - i0)
        if lib.saber_k_unpack_masks(_ptr(bits[i0:i0 + k]), k, H, W, _ptr(dev), _stream()) != 0:
            raise _lib.SaberAmdError(lib.saber_k_last_error().decode())
        pin[:k].copy_(dev[:k], non_blocking=True)
        torch.cuda.current_stream(bits.device).synchronize()
        out[i0:i0 + k] = pin[:k].numpy()
    return out.view(np.bool_)


def make_amg_params(amg: Optional[dict] = None) -> "_lib.AmgParams":
    """cfgAMG dict (saber/adapters/sam2/amg.py:7-17) -> C struct, with the upstream defaults SABER does not pass."""
    a = dict(npoints=32, points_per_batch=64, pred_iou_thresh=0.7, stability_score_thresh=0.92, stability_score_offset=0.7,
             crop_n_layers=2, box_nms_thresh=0.7, crop_n_points_downscale_factor=2, use_m2m=True, multimask_output=True)
    a["crop_nms_thresh"] = 0.7          # upstream default; not a cfgAMG field (bench.py's tail mode and tests may override it)
    a.update({k: v for k, v in (amg or {}).items() if k in a})
    return _lib.AmgParams(points_per_side=a["npoints"], points_per_batch=a["points_per_batch"], pred_iou_thresh=a["pred_iou_thresh"],
                          stability_score_thresh=a["stability_score_thresh"], stability_score_offset=a["stability_score_offset"],
                          mask_threshold=0.0, box_nms_thresh=a["box_nms_thresh"], crop_n_layers=a["crop_n_layers"], crop_nms_thresh=a["crop_nms_thresh"],
                          crop_overlap_ratio=512 / 1500, crop_n_points_downscale_factor=a["crop_n_points_downscale_factor"],
                          use_m2m=int(a["use_m2m"]), multimask_output=int(a["multimask_output"]))
