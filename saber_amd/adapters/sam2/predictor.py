"""SAM2Adapter on the MI355X engine (reference: saber/adapters/sam2/predictor.py).

Hot-path method: segment_image_2d (:48-70) = prep.prepare -> lazily built mask generator -> generate.
The video-propagation methods of the ABC (set_volume / add_new_mask / propagate_in_video / segment_volume, :76-348) run the memory path
of saber_amd.adapters.sam2.video on a second engine handle built from config.cfg, like the reference's second model."""
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from saber_amd.adapters.base import BaseAdapter, SAM2AdapterConfig
from saber_amd.adapters.sam2.amg import cfgAMG
from saber_amd.adapters.sam2.automask import build_amg
from saber_amd.utils import preprocessing as prep



class SAM2Adapter(BaseAdapter):
    def __init__(self, config: SAM2AdapterConfig, device="cuda"):
        if config.num_maskmem > 7:
            raise ValueError("num_maskmem must be at most 7")
        self._config = config
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        # the reference builds its video predictor from config.cfg here; that model belongs to the propagation path
        # (next row).  The AMG model is chosen by amg_cfg.sam2_cfg (automask.py:61) and is built lazily below.
        self.frame_metrics: Dict[int, Dict[int, Dict[str, Any]]] = {}
        self._vol_shape: Optional[Tuple[int, int, int]] = None
        self.inference_state = None
        self._mask_generator = None
        self._video_predictor = None

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

    # ------------------------------------------------------------------ video path
    def _video(self):
        """The video predictor shares the trunk named by config.cfg (reference predictor.py:21-26 builds it from config.cfg, the AMG model
        from amg_cfg.sam2_cfg - two models, as in the reference)."""
        if self._video_predictor is None:
            from saber_amd import pretrained_weights
            from saber_amd.adapters.sam2.automask import get_engine
            from saber_amd.adapters.sam2.video import VideoPredictor
            # max_images = frames per batched encoder pass of the tracking loop (VideoPredictor._frame)
            eng = get_engine(self._config.cfg, self.device, self._config.checkpoint, max_images=16, max_prompts=8, replica=1000)
            W = pretrained_weights.load_weights(self._config.cfg, self._config.checkpoint, video=True)
            self._video_predictor = VideoPredictor(eng, W, num_maskmem=self._config.num_maskmem)
        return self._video_predictor

    @torch.inference_mode()
    def set_volume(self, tomogram: np.ndarray, offload_video_to_cpu: bool = False) -> None:
        """predictor.py:76-86: normalise the tomogram, resize every slice to the model's 1024^2, start an empty inference state"""
        from saber_amd.adapters.sam2.video import load_tomogram_frames_device
        self._vol_shape = tomogram.shape
        self.frame_metrics = {}
        vp = self._video()
        frames = load_tomogram_frames_device(tomogram, vp.lib, vp.dev, 1024, self._config.light_modality)
        vp.init_state(frames, video_hw=(1024, 1024))            # the reference reports the RESIZED size as the video size (preprocessing.py:24)
        self.inference_state = vp

    def add_new_mask(self, frame_idx: int, obj_id: int, mask: np.ndarray, inference_state=None) -> Tuple:
        state = inference_state or self.inference_state
        if state is None:
            raise RuntimeError("Call set_volume() before add_new_mask().")
        return state.add_new_mask(frame_idx, obj_id, mask)

    def add_new_points_or_box(self, frame_idx: int, obj_id: int, inference_state=None, **kwargs) -> Tuple:
        """predictor.py:171-180: delegates to the video predictor (points=, labels=, clear_old_points=, normalize_coords=, box=).  Clicks,
        several clicks per call, boxes and corrections of tracked frames follow upstream (adapters/sam2/video.py); prompts of more than one
        point are decoded in the engine's exact precision mode, which the handle must have been created with (NotImplementedError says so
        otherwise).  No SABER caller uses them: segmenters/base.py:265-280 seeds propagation with masks."""
        state = inference_state or self.inference_state
        if state is None:
            raise RuntimeError("Call set_volume() before add_new_points_or_box().")
        return state.add_new_points_or_box(frame_idx, obj_id, **kwargs)

    @torch.inference_mode()
    def propagate_in_video(self, start_frame_idx, max_frame_num_to_track=None, reverse=False, inference_state=None) -> Iterator:
        """yields (frame_idx, obj_ids, mask_logits, mask_logits, None) like the reference (predictor.py:181-202)"""
        state = inference_state or self.inference_state
        if state is None:
            raise RuntimeError("Call set_volume() before propagate_in_video().")
        for t, ids, logits in state.propagate_in_video(start_frame_idx, max_frame_num_to_track, reverse):
            yield t, ids, logits, logits, None

    @staticmethod
    def _normalize_masks(masks) -> List[np.ndarray]:
        if masks is None:
            return []
        if isinstance(masks, torch.Tensor):
            masks = masks.cpu().numpy()
        if isinstance(masks, np.ndarray) and masks.ndim >= 3:
            return [np.squeeze(masks[i]).astype(np.float32) for i in range(masks.shape[0])]
        out = []
        for m in masks:
            if isinstance(m, torch.Tensor):
                m = m.cpu().numpy()
            if isinstance(m, dict):
                m = m["segmentation"]
            out.append(np.squeeze(m).astype(np.float32))
        return out

    @torch.inference_mode()
    def segment_volume(self, start_frame_idx: int, masks=None, vol_shape=None, max_frame_num_to_track=None,
                       min_presence_score: float = 0.5, inference_state=None) -> np.ndarray:
        """Bidirectional propagation + presence-score filter, predictor.py:232-348 step by step.  The reference captures the mask decoder's
        object-score logits with a forward hook and files them under `_current_frame`, which it updates only AFTER the generator has
        yielded a frame: a frame's scores therefore land on the frame yielded before it.  Reproduced as is (it feeds the boundary fit)."""
        from saber_amd.filters.estimate_thickness import fit_organelle_boundaries
        state = inference_state or self.inference_state
        if state is None:
            raise RuntimeError("Call set_volume() before segment_volume().")
        if vol_shape is None:
            vol_shape = self._vol_shape
        if vol_shape is None:
            raise RuntimeError("vol_shape required when inference_state is passed explicitly.")
        Z, H, W = vol_shape
        mask_list = self._normalize_masks(masks)
        current = {"frame": None}
        captured: Dict[Any, list] = {}
        state.hook = lambda score: captured.setdefault(current["frame"], []).append(np.array([score], dtype=np.float32))
        try:
            for obj_id, mask in enumerate(mask_list, start=1):
                if np.max(mask) == 0:
                    continue
                self.add_new_mask(frame_idx=start_frame_idx, obj_id=obj_id, mask=mask, inference_state=state)
            self.frame_metrics = {}
            # the label volume is painted on the device (threshold + nearest resize to the tomogram's size + label, one launch per object
            # and frame) and comes back to the host once, after both passes
            import ctypes as C
            vol_dev = torch.zeros((Z, H, W), dtype=torch.int16, device=state.dev)
            flag = torch.zeros(1, dtype=torch.int32, device=state.dev)
            painted = set()

            def _apply(frame_idx, obj_ids, mask_logits, want_flag=False):
                Hv, Wv = mask_logits.shape[-2:]
                stream = C.c_void_p(torch.cuda.current_stream(state.dev).cuda_stream)
                for i, obj_id in enumerate(obj_ids):
                    lg = mask_logits[i, 0]
                    if state.lib.saber_k_paint_nearest(C.c_void_p(lg.data_ptr()), Hv, Wv, 0.0, int(obj_id), C.c_void_p(vol_dev[frame_idx].data_ptr()), H, W,
                                                       C.c_void_p(flag.data_ptr()) if want_flag else None, stream) != 0:
                        raise RuntimeError(state.lib.saber_k_last_error().decode())
                painted.add(frame_idx)

            for frame_idx, obj_ids, mask_logits, _, _ in self.propagate_in_video(start_frame_idx, max_frame_num_to_track, False, state):
                current["frame"] = frame_idx
                _apply(frame_idx, obj_ids, mask_logits, want_flag=(frame_idx == start_frame_idx))
            for frame_idx, obj_ids, mask_logits, _, _ in self.propagate_in_video(start_frame_idx, max_frame_num_to_track, True, state):
                current["frame"] = frame_idx
                # predictor.py:318-319: only frames the forward pass left empty.  Both passes share the start frame alone; whether its
                # forward masks painted a pixel is the one thing read back here
                if frame_idx not in painted or (frame_idx == start_frame_idx and int(flag.item()) == 0):
                    _apply(frame_idx, obj_ids, mask_logits)
            vol_masks = vol_dev.cpu().numpy().view(np.uint16)
        finally:
            state.hook = None
        n_masks = len(mask_list)
        if n_masks > 0:
            frame_scores = np.zeros([Z, n_masks])
            for fidx, scores in captured.items():
                if fidx is None:
                    continue
                v = np.concatenate([s.flatten() for s in scores])
                n = min(len(v), n_masks)
                frame_scores[fidx, :n] = v[:n]
            self.frame_scores = frame_scores
            bounds = fit_organelle_boundaries(frame_scores, plot=False)
            for fidx in range(Z):
                self.frame_metrics[fidx] = {}
                for mi in range(n_masks):
                    obj_id = mi + 1
                    ps = float(bounds[fidx, mi])
                    self.frame_metrics[fidx][obj_id] = {"presence_score": ps}
                    if ps < min_presence_score:
                        vol_masks[fidx][vol_masks[fidx] == obj_id] = 0
        return vol_masks.astype(np.uint16)

    def clear_all_prompts_in_frame(self, *args, inference_state=None, **kwargs):
        """predictor.py:360-362: delegates to the video predictor (frame_idx, obj_id, need_output=True).  The reference passes upstream's
        inference_state as the first positional argument; here the state IS the predictor, so a leading state object is dropped."""
        state = inference_state or self.inference_state
        if args and hasattr(args[0], "clear_all_prompts_in_frame"):
            state, args = args[0], args[1:]
        if state is None:
            raise RuntimeError("Call set_volume() before clear_all_prompts_in_frame().")
        return state.clear_all_prompts_in_frame(*args, **kwargs)

    def remove_object(self, *args, inference_state=None, **kwargs):
        """predictor.py:364-366: delegates to the video predictor (obj_id, strict=False, need_output=True)."""
        state = inference_state or self.inference_state
        if args and hasattr(args[0], "remove_object"):
            state, args = args[0], args[1:]
        if state is None:
            raise RuntimeError("Call set_volume() before remove_object().")
        return state.remove_object(*args, **kwargs)

    def reset_state(self, inference_state=None) -> None:
        state = inference_state or self.inference_state
        if state is not None:
            state.reset_state()          # like upstream's reset_state: prompts and tracking results go, the loaded frames stay
