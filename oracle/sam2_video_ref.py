"""ORACLE (test infrastructure, never shipped) for the NEXT row of SURVEY.md section 8(f): SAM2 video propagation
(`SAM2Adapter.segment_volume`, saber/adapters/sam2/predictor.py:232-348, which drives the third-party `sam2` video predictor).

Restated here, from the published SAM 2.1 architecture, are the two model blocks the image path does not have:

  memory_attention   4 layers of [RoPE self-attention over the 64x64 frame tokens, RoPE cross-attention to the memory tokens
                     (64-channel spatial memories + object-pointer tokens, the pointers excluded from the rotation), ReLU MLP],
                     pre-norm residuals, final LayerNorm                      (upstream sam2/modeling/memory_attention.py)
  memory_encoder     mask -> 4 x [3x3 stride-2 conv, LayerNorm2d, GELU] -> 1x1 conv, added to the 1x1-projected frame
                     features, 2 ConvNeXt blocks (7x7 depthwise, LayerNorm2d, 4x MLP, layer scale), 1x1 to 64 channels,
                     plus the normalised sine position encoding of the result   (upstream sam2/modeling/memory_encoder.py)

Weights are addressed by their upstream checkpoint keys (`memory_attention.layers.0.self_attn.q_proj.weight`, ...), like
oracle/sam2_ref.py.  Parity unpinned (the reference pins nothing here and `sam2` is absent); cross-checked in this container
against the independent restatement in `transformers` (`Sam2VideoMemoryAttention`, `Sam2VideoMemoryEncoder`) with shared random
weights: oracle/hf_crosscheck_video.py, tests/test_oracle_video.py.  The tracking loop around them (memory bank, object pointers,
temporal encodings, occlusion logic, the adapter's bidirectional pass with its presence-score filter) follows further down:
VideoPredictorRef and segment_volume_ref.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

T = torch.Tensor


def _lin(W: Dict[str, T], key: str, x: T) -> T:
    return F.linear(x, W[key + ".weight"], W[key + ".bias"])


def _ln(W: Dict[str, T], key: str, x: T, eps: float = 1e-5) -> T:
    return F.layer_norm(x, (x.shape[-1],), W[key + ".weight"], W[key + ".bias"], eps)


def _ln2d(W: Dict[str, T], key: str, x: T, eps: float = 1e-6) -> T:
    """LayerNorm over the channel axis of an NCHW map (upstream LayerNorm2d)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return W[key + ".weight"][None, :, None, None] * ((x - u) / torch.sqrt(s + eps)) + W[key + ".bias"][None, :, None, None]


# ------------------------------------------------------------------------------------------------ axial RoPE
def rope_table(end_x: int = 64, end_y: int = 64, dim: int = 256, theta: float = 10000.0):
    """(cos, sin) of shape (end_x * end_y, dim): the first dim/2 channels rotate with the x coordinate, the rest with y;
    consecutive channel PAIRS share a frequency (complex multiplication of (even, odd) pairs upstream)."""
    f = 1.0 / (theta ** (torch.arange(0, dim, 4)[: dim // 4].float() / dim))
    idx = torch.arange(end_x * end_y)
    ang = torch.cat([torch.outer((idx % end_x).float(), f), torch.outer(torch.div(idx, end_x, rounding_mode="floor").float(), f)], -1)
    ang = ang.repeat_interleave(2, -1)
    return ang.cos(), ang.sin()


def rope_rotate(x: T, cos: T, sin: T) -> T:
    """x: (..., n, dim) with n == cos.shape[0]; (a, b) -> (a cos - b sin, b cos + a sin) per channel pair."""
    a, b = x[..., 0::2], x[..., 1::2]
    rot = torch.stack((-b, a), -1).flatten(-2)
    return x * cos + rot * sin


def rope_attention(W: Dict[str, T], key: str, q_in: T, k_in: T, v_in: T, cos: T, sin: T, heads: int = 1, n_k_exclude: int = 0,
                   repeat_k: bool = False) -> T:
    """Attention with rotary position encoding on q and on the first (n_k - n_k_exclude) keys.  Inputs (B, n, C)."""
    q, k, v = _lin(W, key + ".q_proj", q_in), _lin(W, key + ".k_proj", k_in), _lin(W, key + ".v_proj", v_in)
    B, nq, D = q.shape
    hd = D // heads
    sp = lambda t: t.view(B, -1, heads, hd).transpose(1, 2)
    q, k, v = sp(q), sp(k), sp(v)
    q = rope_rotate(q, cos, sin)
    n_rot = k.shape[2] - n_k_exclude
    if n_rot > 0:
        ck, sk = cos, sin
        if repeat_k and n_rot != nq:        # memories of several frames: the 64x64 table repeats per frame
            r = n_rot // nq
            ck, sk = cos.repeat(r, 1), sin.repeat(r, 1)
        k = torch.cat([rope_rotate(k[:, :, :n_rot], ck, sk), k[:, :, n_rot:]], 2)
    a = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, -1) @ v
    return _lin(W, key + ".out_proj", a.transpose(1, 2).reshape(B, nq, D))


def memory_attention(W: Dict[str, T], curr: T, memory: T, curr_pos: T, memory_pos: T, num_obj_ptr_tokens: int = 0, n_layers: int = 4,
                     rope_feat=(64, 64), theta: float = 10000.0, prefix: str = "memory_attention") -> T:
    """curr (B, HW, 256) frame tokens, curr_pos their position encoding (added once, scaled by 0.1);
    memory (B, N, 64) memory tokens [spatial memories ..., object pointers], memory_pos (B, N, 64).  Returns (B, HW, 256)."""
    cos, sin = rope_table(rope_feat[0], rope_feat[1], curr.shape[-1], theta)
    x = curr + 0.1 * curr_pos
    for i in range(n_layers):
        p = f"{prefix}.layers.{i}"
        t = _ln(W, p + ".norm1", x)
        x = x + rope_attention(W, p + ".self_attn", t, t, t, cos, sin)
        t = _ln(W, p + ".norm2", x)
        x = x + rope_attention(W, p + ".cross_attn_image", t, memory + memory_pos, memory, cos, sin, n_k_exclude=num_obj_ptr_tokens,
                               repeat_k=True)
        t = _ln(W, p + ".norm3", x)
        x = x + _lin(W, p + ".linear2", F.relu(_lin(W, p + ".linear1", t)))
    return _ln(W, prefix + ".norm", x)


# ------------------------------------------------------------------------------------------------ memory encoder
def sine_position_encoding(shape, num_pos_feats: int = 32, temperature: float = 10000.0) -> T:
    """Normalised 2-D sine encoding (B, 2 * num_pos_feats, H, W): y half first, then x (upstream PositionEmbeddingSine,
    normalize=True, scale 2 pi)."""
    B, _, H, Wd = shape
    eps, scale = 1e-6, 2 * math.pi
    y = torch.arange(1, H + 1, dtype=torch.float32)[:, None].expand(H, Wd)
    x = torch.arange(1, Wd + 1, dtype=torch.float32)[None, :].expand(H, Wd)
    y = y / (H + eps) * scale
    x = x / (Wd + eps) * scale
    d = torch.arange(num_pos_feats, dtype=torch.float32)
    d = temperature ** (2 * torch.div(d, 2, rounding_mode="floor") / num_pos_feats)
    px, py = x[:, :, None] / d, y[:, :, None] / d
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), 3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), 3).flatten(2)
    return torch.cat((py, px), 2).permute(2, 0, 1)[None].expand(B, -1, -1, -1)


def mask_downsampler(W: Dict[str, T], masks: T, prefix: str = "memory_encoder.mask_downsampler") -> T:
    """(B,1,H,W) -> (B,256,H/16,W/16): conv/LayerNorm2d/GELU at encoder.{0,1},{3,4},{6,7},{9,10}, final 1x1 at encoder.12."""
    x = masks
    for j in range(4):
        x = F.conv2d(x, W[f"{prefix}.encoder.{3 * j}.weight"], W[f"{prefix}.encoder.{3 * j}.bias"], stride=2, padding=1)
        x = F.gelu(_ln2d(W, f"{prefix}.encoder.{3 * j + 1}", x))
    return F.conv2d(x, W[f"{prefix}.encoder.12.weight"], W[f"{prefix}.encoder.12.bias"])


def memory_encoder(W: Dict[str, T], pix_feat: T, masks: T, skip_mask_sigmoid: bool = False, n_fuser_layers: int = 2,
                   prefix: str = "memory_encoder"):
    """pix_feat (B,256,64,64) frame features, masks (B,1,1024,1024) high-res mask logits.  Returns (memory features (B,64,64,64),
    their position encoding (B,64,64,64))."""
    if not skip_mask_sigmoid:
        masks = torch.sigmoid(masks)
    m = mask_downsampler(W, masks, prefix + ".mask_downsampler")
    x = F.conv2d(pix_feat, W[prefix + ".pix_feat_proj.weight"], W[prefix + ".pix_feat_proj.bias"]) + m
    for i in range(n_fuser_layers):
        p = f"{prefix}.fuser.layers.{i}"
        C = x.shape[1]
        h = F.conv2d(x, W[p + ".dwconv.weight"], W[p + ".dwconv.bias"], padding=W[p + ".dwconv.weight"].shape[-1] // 2, groups=C)
        h = _ln2d(W, p + ".norm", h).permute(0, 2, 3, 1)
        h = _lin(W, p + ".pwconv2", F.gelu(_lin(W, p + ".pwconv1", h)))
        x = x + (W[p + ".gamma"] * h).permute(0, 3, 1, 2)
    x = F.conv2d(x, W[prefix + ".out_proj.weight"], W[prefix + ".out_proj.bias"])
    return x, sine_position_encoding(x.shape, x.shape[1] // 2)


# ------------------------------------------------------------------------------------------------ HF key map (cross-check only)
def from_hf_memory_attention(sd: Dict[str, T], prefix: str = "memory_attention") -> Dict[str, T]:
    out = {}
    for k, v in sd.items():
        k2 = k.replace(".o_proj.", ".out_proj.").replace(".layer_norm1.", ".norm1.").replace(".layer_norm2.", ".norm2.").replace(".layer_norm3.", ".norm3.")
        if k2.startswith("layer_norm."):
            k2 = "norm." + k2[len("layer_norm."):]
        if k2.startswith("rotary_emb"):
            continue
        out[f"{prefix}.{k2}"] = v.detach().float()
    return out


def from_hf_memory_encoder(sd: Dict[str, T], prefix: str = "memory_encoder") -> Dict[str, T]:
    out = {}
    for k, v in sd.items():
        v = v.detach().float()
        if k.startswith("mask_downsampler.layers."):
            _, _, j, kind, leaf = k.split(".")
            out[f"{prefix}.mask_downsampler.encoder.{3 * int(j) + (0 if kind == 'conv' else 1)}.{leaf}"] = v
        elif k.startswith("mask_downsampler.final_conv."):
            out[f"{prefix}.mask_downsampler.encoder.12.{k.split('.')[-1]}"] = v
        elif k.startswith("feature_projection."):
            out[f"{prefix}.pix_feat_proj.{k.split('.')[-1]}"] = v
        elif k.startswith("projection."):
            out[f"{prefix}.out_proj.{k.split('.')[-1]}"] = v
        elif k.startswith("memory_fuser.layers."):
            parts = k.split(".")
            i, name = parts[2], parts[3]
            name = {"depthwise_conv": "dwconv", "layer_norm": "norm", "pointwise_conv1": "pwconv1", "pointwise_conv2": "pwconv2", "scale": "gamma"}[name]
            out[f"{prefix}.fuser.layers.{i}.{name}" + ("" if name == "gamma" else "." + parts[4])] = v
    return out


# ------------------------------------------------------------------------------------------------ tracking loop
# Restated from the published SAM 2.1 video predictor (upstream sam2/sam2_video_predictor.py and sam2/modeling/sam2_base.py as the
# reference drives them: saber/adapters/sam2/predictor.py:24-34 builds it with vos_optimized=False and truncates maskmem_tpos_enc to
# num_maskmem rows; :163-202 add_new_mask / propagate_in_video; :232-348 segment_volume).  Settings of the sam2.1 configs + the video
# predictor's overrides: per-object tracking (batch 1), use_obj_ptrs_in_encoder, max_obj_ptrs_in_encoder 16, signed + projected temporal
# encoding of pointers, only_obj_ptrs_in_the_past_for_eval, pred_obj_scores (+ MLP), fixed_no_obj_ptr, no_obj_embed_spatial,
# use_mlp_for_obj_ptr_proj, multimask_output_for_tracking with 0..1 points, use_multimask_token_for_obj_ptr, directly_add_no_mem_embed,
# sigmoid scale / bias 20 / -10, binarize_mask_from_pts_for_mem_enc, max_cond_frames_in_attn -1, memory_temporal_stride 1,
# non_overlap_masks off, fill_hole_area: the optional CUDA extension is treated as absent (upstream then skips the step with a warning).
# Cross-checked against the independent `transformers` Sam2VideoModel where the two implement the same published steps
# (oracle/hf_crosscheck_video.py); where they differ (mask-prompted frames: upstream encodes their memory in the propagation preflight
# from the UP-SAMPLED low-res mask, binarised), upstream is followed because it is what the reference executes.
NO_OBJ_SCORE = -1024.0


def get_1d_sine_pe(pos: T, dim: int, temperature: float = 10000.0) -> T:
    pe_dim = dim // 2
    dim_t = torch.arange(pe_dim, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / pe_dim)
    e = pos.unsqueeze(-1) / dim_t
    return torch.cat([e.sin(), e.cos()], dim=-1)


def _mlp(W: Dict[str, T], key: str, x: T, n: int) -> T:
    for i in range(n):
        x = _lin(W, f"{key}.layers.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


class VideoPredictorRef:
    """One model, any number of objects tracked independently (as upstream's per-object inference does)."""

    def __init__(self, weights, cfg, num_maskmem: int = 2, cond_memory_from_full_res_mask: bool = False):
        """cond_memory_from_full_res_mask: the one step where the independent `transformers` restatement departs from upstream - it encodes
        the memory of a mask-prompted frame from the prompt mask at full resolution, upstream (and therefore the reference and this
        oracle's default) from the low-res output up-sampled again.  Only oracle/hf_crosscheck_tracking.py sets it, to pin everything else."""
        from oracle import sam2_ref
        self.cond_memory_from_full_res_mask = cond_memory_from_full_res_mask
        if num_maskmem > 7:
            raise ValueError("num_maskmem must be less than 7")                 # reference predictor.py:28-29
        self.S = sam2_ref
        self.W = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v).float()) for k, v in weights.items()}
        self.cfg = cfg
        self.num_maskmem = num_maskmem
        self.tpos = self.W["maskmem_tpos_enc"][:num_maskmem]                    # (num_maskmem,1,1,64): predictor.py:31-32
        self.image_size = cfg.image_size
        self._dense_pe = sam2_ref.dense_pe(self.W, self.image_size // 16)
        self.hook = None                                                         # called with the object-score logits of every decoder call
        self.feat_memo = None     # tests may set a dict shared by several predictors over the SAME weights: frame content -> features (saves re-encoding; the arithmetic is untouched)

    # ---- state
    def init_state(self, images: T, video_hw=None):
        """images (Z,3,S,S) float tensor exactly as the adapter feeds them (adapters/preprocessing.py:27-70: values in [-3,1] for
        a tomogram normalised to [-1,1], no ImageNet statistics)."""
        self.images = images
        self.num_frames = images.shape[0]
        self.video_hw = video_hw or tuple(images.shape[-2:])
        self.feat_cache = {}
        self.obj_ids = []
        self.out = {}            # obj_id -> {"cond": {t: out}, "non_cond": {t: out}}
        self.temp = {}           # obj_id -> {t: out} (prompted frames awaiting the preflight)
        self._points = {}        # (obj_id, t) -> (points in model pixels (1,k,2), labels (1,k)): upstream's point_inputs_per_obj
        self._tracked = {}       # (obj_id, t) -> reverse flag of the propagation that last went over the frame: upstream's frames_tracked_per_obj

    @torch.no_grad()
    def _feats(self, t: int):
        if t not in self.feat_cache:
            key = None
            if self.feat_memo is not None:
                import hashlib
                key = hashlib.blake2b(self.images[t].contiguous().numpy().tobytes(), digest_size=16).digest()
                if key in self.feat_memo:
                    self.feat_cache = {t: self.feat_memo[key]}
                    return self.feat_cache[t]
            S, W = self.S, self.W
            fpn = S.fpn_neck(W, self.cfg, S.hiera_trunk(W, self.cfg, self.images[t:t + 1]))
            d = "sam_mask_decoder."
            s0 = F.conv2d(fpn[0], W[d + "conv_s0.weight"], W[d + "conv_s0.bias"])
            s1 = F.conv2d(fpn[1], W[d + "conv_s1.weight"], W[d + "conv_s1.bias"])
            pos = sine_position_encoding(fpn[2].shape, 128)
            self.feat_cache = {t: (fpn[2], pos, s0, s1)}                         # upstream caches the latest frame only
            if key is not None:
                self.feat_memo[key] = self.feat_cache[t]
        return self.feat_cache[t]

    # ---- SAM heads on a (possibly memory-conditioned) feature map
    @torch.no_grad()
    def _sam_heads(self, pix_feat: T, s0: T, s1: T, mask_inputs: Optional[T], multimask_output: bool, pts: Optional[T] = None, lab: Optional[T] = None):
        S, W = self.S, self.W
        if pts is None:         # no point prompt: one padding point with label -1 (SAM2Base._forward_sam_heads)
            pts = torch.zeros(1, 1, 2)
            lab = -torch.ones(1, 1, dtype=torch.int64)
        sparse, dense = S.prompt_encoder(W, pts, lab, mask_inputs, self.image_size)
        feats = {"image_embed": pix_feat, "feat_s0": s0, "feat_s1": s1}
        low_multi, ious, obj, all_masks, all_iou, toks = S.mask_decoder(W, feats, sparse, dense, multimask_output, self._dense_pe, return_tokens=True)
        if self.hook is not None:
            self.hook(obj.clone())
        appearing = obj > 0                                                       # (1,1)
        low_multi = torch.where(appearing[:, None, None], low_multi, torch.full_like(low_multi, NO_OBJ_SCORE))
        if multimask_output:
            best = int(torch.argmax(ious, dim=-1))
            low = low_multi[:, best:best + 1]
            tok = toks[:, 1 + best]
        else:
            low = low_multi
            tok = toks[:, 0]
        high = F.interpolate(low, size=(self.image_size, self.image_size), mode="bilinear", align_corners=False)
        ptr = _mlp(W, "obj_ptr_proj", tok, 3)
        lam = appearing.float()
        ptr = lam * ptr + (1 - lam) * W["no_obj_ptr"]
        return low, high, ptr, obj

    @torch.no_grad()
    def _use_mask_as_output(self, pix_feat: T, s0: T, s1: T, mask_inputs: T):
        """mask_inputs (1,1,S,S) float 0/1"""
        W = self.W
        high = mask_inputs * 20.0 - 10.0
        low = F.interpolate(high, size=(high.shape[-2] // 4, high.shape[-1] // 4), mode="bilinear", align_corners=False, antialias=True)
        md = F.conv2d(mask_inputs, W["mask_downsample.weight"], W["mask_downsample.bias"], stride=4)
        _, _, ptr, _ = self._sam_heads(pix_feat, s0, s1, md, multimask_output=False)
        appearing = torch.any(mask_inputs.flatten(1) > 0.0, dim=1)[:, None]
        lam = appearing.float()
        obj = 20.0 * lam - 10.0
        ptr = lam * ptr + (1 - lam) * W["no_obj_ptr"]
        return low, high, ptr, obj

    @torch.no_grad()
    def _encode_memory(self, pix_feat_raw: T, high_res_masks: T, obj: T, is_mask_from_pts: bool):
        W = self.W
        m = (high_res_masks > 0).float() if is_mask_from_pts else torch.sigmoid(high_res_masks)
        m = m * 20.0 - 10.0
        feat, pos = memory_encoder(W, pix_feat_raw, m, skip_mask_sigmoid=True)
        feat = feat + (1 - (obj > 0).float())[..., None, None] * W["no_obj_embed_spatial"][..., None, None]
        return feat.to(torch.bfloat16).float(), pos                                # upstream stores the memory in bfloat16

    # ---- prompts
    @torch.no_grad()
    def add_new_mask(self, frame_idx: int, obj_id: int, mask):
        """mask: (H,W) array, non-zero = object.  Returns the frame's masks at video resolution like upstream's add_new_mask."""
        if obj_id not in self.obj_ids:
            self.obj_ids.append(obj_id)
            self.out[obj_id] = {"cond": {}, "non_cond": {}}
            self.temp[obj_id] = {}
        m = torch.as_tensor(np.asarray(mask)).float()[None, None]
        if m.shape[-2:] != (self.image_size, self.image_size):
            m = F.interpolate(m, size=(self.image_size, self.image_size), mode="bilinear", align_corners=False, antialias=True)
            m = (m >= 0.5).float()
        else:
            m = (m > 0).float() if m.dtype != torch.float32 else m
        pix, pos, s0, s1 = self._feats(frame_idx)
        low, high, ptr, obj = self._use_mask_as_output(pix, s0, s1, m)
        self.temp[obj_id][frame_idx] = {"pred_masks": low, "obj_ptr": ptr, "object_score_logits": obj, "maskmem_features": None, "maskmem_pos_enc": None}
        if self.cond_memory_from_full_res_mask:
            self.temp[obj_id][frame_idx]["high_res_for_memory"] = high
        return frame_idx, list(self.obj_ids), F.interpolate(low, size=self.video_hw, mode="bilinear", align_corners=False)

    @torch.no_grad()
    def add_new_points_or_box(self, frame_idx: int, obj_id: int, points=None, labels=None, clear_old_points: bool = True, normalize_coords: bool = True, box=None):
        """upstream SAM2VideoPredictor.add_new_points_or_box on a frame that has not been tracked yet (an initial conditioning frame: the SAM heads
        run on the frame's own features + no_mem_embed; a previous output on the frame enters as the mask prompt, clamped to +-32).  A box is
        its two corners with labels 2 / 3 in front of the clicks (only with clear_old_points, as upstream); clicks accumulate over calls unless
        clear_old_points; multimask output (best mask by predicted IoU) only while there is at most ONE point (SAM2Base._use_multimask with
        multimask_min_pt_num 0 / multimask_max_pt_num 1), otherwise the single-mask output with dynamic_multimask_via_stability."""
        if (points is not None) != (labels is not None):
            raise ValueError("points and labels must be provided together")
        if points is None and box is None:
            raise ValueError("at least one of points or box must be provided as input")
        pts = torch.zeros(1, 0, 2) if points is None else torch.as_tensor(np.asarray(points), dtype=torch.float32).reshape(1, -1, 2)
        lab = torch.zeros(1, 0, dtype=torch.int64) if labels is None else torch.as_tensor(np.asarray(labels), dtype=torch.int64).reshape(1, -1)
        if box is not None:
            if not clear_old_points:
                raise ValueError("cannot add box without clearing old points, since box prompt must be provided before any point prompt "
                                 "(please use clear_old_points=True instead)")
            pts = torch.cat([torch.as_tensor(np.asarray(box), dtype=torch.float32).reshape(1, 2, 2), pts], 1)
            lab = torch.cat([torch.tensor([[2, 3]], dtype=torch.int64), lab], 1)
        if obj_id not in self.obj_ids:
            self.obj_ids.append(obj_id)
            self.out[obj_id] = {"cond": {}, "non_cond": {}}
            self.temp[obj_id] = {}
        if normalize_coords:
            pts = pts / torch.tensor([self.video_hw[1], self.video_hw[0]], dtype=torch.float32)
        pts = pts * self.image_size
        held = self._points
        old = None if clear_old_points else held.get((obj_id, frame_idx))
        if old is not None:
            pts, lab = torch.cat([old[0], pts], 1), torch.cat([old[1], lab], 1)
        held[(obj_id, frame_idx)] = (pts, lab)
        # A frame a propagation has already gone over is CORRECTED (upstream: is_init_cond_frame False): its features are conditioned on the
        # memory bank in the direction it was tracked, and its new output stays a non-conditioning one (add_all_frames_to_correct_as_cond False)
        is_init = (obj_id, frame_idx) not in self._tracked
        prev = self.temp[obj_id].get(frame_idx) or self.out[obj_id]["cond"].get(frame_idx) or self.out[obj_id]["non_cond"].get(frame_idx)
        mask_in = torch.clamp(prev["pred_masks"], -32.0, 32.0) if prev is not None else None
        pix, pos, s0, s1 = self._feats(frame_idx)
        if is_init:
            feat = pix + self.W["no_mem_embed"].view(1, -1, 1, 1)
        else:
            feat = self._memory_conditioned(obj_id, frame_idx, pix, pos, self._tracked[(obj_id, frame_idx)])
        low, high, ptr, obj = self._sam_heads(feat, s0, s1, mask_in, multimask_output=pts.shape[1] <= 1, pts=pts, lab=lab)
        self.temp[obj_id][frame_idx] = {"pred_masks": low, "obj_ptr": ptr, "object_score_logits": obj, "maskmem_features": None, "maskmem_pos_enc": None,
                                        "is_cond": is_init}
        return frame_idx, list(self.obj_ids), F.interpolate(low, size=self.video_hw, mode="bilinear", align_corners=False)

    @torch.no_grad()
    def _preflight(self):
        for oid in self.obj_ids:
            for t, out in self.temp[oid].items():
                if out["maskmem_features"] is None:
                    high = F.interpolate(out["pred_masks"], size=(self.image_size, self.image_size), mode="bilinear", align_corners=False)
                    high = out.pop("high_res_for_memory", high)
                    pix = self._feats(t)[0]
                    out["maskmem_features"], out["maskmem_pos_enc"] = self._encode_memory(pix, high, out["object_score_logits"], is_mask_from_pts=True)
                if out.pop("is_cond", True):
                    self.out[oid]["cond"][t] = out
                else:
                    self.out[oid]["non_cond"][t] = out
            for t in self.out[oid]["cond"]:              # upstream keeps the two dictionaries disjoint (a corrected conditioning frame keeps its old output)
                self.out[oid]["non_cond"].pop(t, None)
            self.temp[oid] = {}
            if not self.out[oid]["cond"]:
                raise RuntimeError("No input points or masks are provided for any object; please add inputs first.")

    # ---- one tracked frame of one object
    @torch.no_grad()
    def _memory_conditioned(self, oid, t: int, pix: T, pos: T, reverse: bool) -> T:
        W = self.W
        store = self.out[oid]
        mems, mem_pos = [], []
        for tc, o in store["cond"].items():                                       # every conditioning frame (max_cond_frames_in_attn = -1)
            mems.append(o["maskmem_features"]); mem_pos.append((o["maskmem_pos_enc"], self.num_maskmem - 1))
        for t_pos in range(1, self.num_maskmem):
            t_rel = self.num_maskmem - t_pos
            prev = t + t_rel if reverse else t - t_rel
            o = store["non_cond"].get(prev)
            if o is None:
                continue
            mems.append(o["maskmem_features"]); mem_pos.append((o["maskmem_pos_enc"], self.num_maskmem - t_pos - 1))
        flat = lambda x: x.flatten(2).permute(0, 2, 1)                            # (1,C,H,W) -> (1,HW,C)
        memory = [flat(m) for m in mems]
        memory_pos = [flat(p) + self.tpos[ti].view(1, 1, -1) for p, ti in mem_pos]      # temporal encoding added per channel
        # object pointers: conditioning frames in the (temporal) past, then up to max_obj_ptrs - 1 tracked frames before this one
        max_ptrs = min(self.num_frames, 16)
        sign = -1 if reverse else 1
        offs, ptrs = [], []
        for tc, o in store["cond"].items():
            if (tc >= t) if reverse else (tc <= t):
                offs.append((t - tc) * sign); ptrs.append(o["obj_ptr"])
        for d in range(1, max_ptrs):
            tt = t + d if reverse else t - d
            if tt < 0 or tt >= self.num_frames:
                break
            o = store["non_cond"].get(tt)
            if o is not None:
                offs.append(d); ptrs.append(o["obj_ptr"])
        n_ptr_tokens = 0
        if ptrs:
            P = torch.stack(ptrs, 0)[:, 0]                                        # (n,256)
            pe = get_1d_sine_pe(torch.tensor(offs, dtype=torch.float32) / float(max_ptrs - 1), 256)
            pe = _lin(W, "obj_ptr_tpos_proj", pe)                                 # (n,64)
            memory.append(P.reshape(-1, 4, 64).reshape(1, -1, 64))
            memory_pos.append(pe.repeat_interleave(4, dim=0)[None])
            n_ptr_tokens = P.shape[0] * 4
        out = memory_attention(W, flat(pix), torch.cat(memory, 1), flat(pos), torch.cat(memory_pos, 1), n_ptr_tokens)
        return out.permute(0, 2, 1).reshape(pix.shape)

    @torch.no_grad()
    def _track(self, oid, t: int, reverse: bool):
        pix, pos, s0, s1 = self._feats(t)
        cond = self._memory_conditioned(oid, t, pix, pos, reverse)
        low, high, ptr, obj = self._sam_heads(cond, s0, s1, None, multimask_output=True)
        feat, mpos = self._encode_memory(pix, high, obj, is_mask_from_pts=False)
        return {"pred_masks": low, "obj_ptr": ptr, "object_score_logits": obj, "maskmem_features": feat, "maskmem_pos_enc": mpos}

    @torch.no_grad()
    def propagate_in_video(self, start_frame_idx: int, max_frame_num_to_track=None, reverse: bool = False):
        """yields (frame_idx, obj_ids, video_res_masks (n_obj,1,Hv,Wv)) like upstream"""
        self._preflight()
        n = self.num_frames
        if max_frame_num_to_track is None:
            max_frame_num_to_track = n
        if reverse:
            end = max(start_frame_idx - max_frame_num_to_track, 0)
            order = range(start_frame_idx, end - 1, -1) if start_frame_idx > 0 else []
        else:
            end = min(start_frame_idx + max_frame_num_to_track, n - 1)
            order = range(start_frame_idx, end + 1)
        for t in order:
            lows = []
            for oid in self.obj_ids:
                self._tracked[(oid, t)] = reverse
                if t in self.out[oid]["cond"]:
                    lows.append(self.out[oid]["cond"][t]["pred_masks"])
                else:
                    o = self._track(oid, t, reverse)
                    self.out[oid]["non_cond"][t] = o
                    lows.append(o["pred_masks"])
            allm = torch.cat(lows, 0)
            yield t, list(self.obj_ids), F.interpolate(allm, size=self.video_hw, mode="bilinear", align_corners=False)


def load_tomogram_frames(tomogram: "np.ndarray", image_size: int = 1024, light_modality: bool = False) -> T:
    """adapters/preprocessing.py:27-76 as SAM2Adapter.create_inference_state_from_tomogram applies it: min-max to [-1,1], per-slice resize
    to image_size (skimage.transform.resize, anti_aliasing=True: the identity at image_size, order-1 interpolation at pixel centres,
    preceded by a Gaussian of sigma (factor - 1) / 2 on down-sampled axes),
    3 x channel repeat, then 2x - 1 again (img_mean / img_std are None on this path)."""
    t = np.asarray(tomogram, dtype=np.float64)
    t = (t - t.min()) / (t.max() - t.min())
    t = (t * 2 - 1)
    x = torch.from_numpy(t).float()[:, None]
    if x.shape[-2:] != (image_size, image_size):
        if x.shape[-2] > image_size or x.shape[-1] > image_size:
            # skimage.transform.resize(anti_aliasing=True), restated from its published source (skimage is absent here): on every axis
            # that is down-sampled, scipy.ndimage.gaussian_filter with sigma = (input / output - 1) / 2, cval 0, boundary mode 'mirror'
            # (skimage's default mode 'reflect' maps to ndimage 'mirror'), truncate 4; then order-1 interpolation at pixel centres
            from scipy import ndimage as ndi
            H, Wd = x.shape[-2:]
            sig = (max(0.0, (H / image_size - 1) / 2), max(0.0, (Wd / image_size - 1) / 2))
            x = torch.from_numpy(np.stack([ndi.gaussian_filter(pl[0].numpy(), sig, mode="mirror") for pl in x]))[:, None]
        x = F.interpolate(x, size=(image_size, image_size), mode="bilinear", align_corners=False)
    x = x.repeat(1, 3, 1, 1)
    x = 2 * x - 1
    if light_modality:
        x = (x - x.min()) / (x.max() - x.min()) * 255
    return x


def segment_volume_ref(pred: VideoPredictorRef, start_frame_idx: int, masks, vol_shape, max_frame_num_to_track=None, min_presence_score: float = 0.5):
    """SAM2Adapter.segment_volume (predictor.py:232-348) on the oracle predictor, INCLUDING its hook bookkeeping: the forward hook on the
    mask decoder files each call's object-score logits under `_current_frame`, which the adapter updates only AFTER the generator has
    yielded a frame - so a frame's scores land on the previously yielded frame index (and the calls made inside add_new_mask on None).
    Returns (vol_masks uint16 (Z,H,W), frame_metrics dict, frame_scores (Z,nMasks))."""
    import numpy as np
    from oracle import saber_ref
    Z, H, Wd = vol_shape
    mask_list = [np.squeeze(np.asarray(m)).astype(np.float32) for m in masks]
    state = {"cur": None}
    captured = {}

    def hook(obj):
        captured.setdefault(state["cur"], []).append(obj.detach().cpu().float().numpy())
    pred.hook = hook
    for obj_id, m in enumerate(mask_list, start=1):
        if np.max(m) == 0:
            continue
        pred.add_new_mask(start_frame_idx, obj_id, m)
    vol = np.zeros((Z, H, Wd), dtype=np.uint16)

    def apply(t, obj_ids, logits):
        for i, oid in enumerate(obj_ids):
            m = np.squeeze((logits[i] > 0.0).numpy()).astype(bool)
            if m.shape != (H, Wd):
                m = saber_ref.resize_nearest(m, (H, Wd))
            vol[t] = np.where(m, int(oid), vol[t])
    for t, ids, logits in pred.propagate_in_video(start_frame_idx, max_frame_num_to_track, reverse=False):
        state["cur"] = t
        apply(t, ids, logits)
    for t, ids, logits in pred.propagate_in_video(start_frame_idx, max_frame_num_to_track, reverse=True):
        state["cur"] = t
        if not vol[t].any():
            apply(t, ids, logits)
    pred.hook = None
    n_masks = len(mask_list)
    metrics, frame_scores = {}, np.zeros([Z, max(n_masks, 1)])[:, :n_masks]
    if n_masks > 0:
        for fidx, scores in captured.items():
            if fidx is None:
                continue
            v = np.concatenate([s.flatten() for s in scores])
            k = min(len(v), n_masks)
            frame_scores[fidx, :k] = v[:k]
        bounds = saber_ref.fit_organelle_boundaries(frame_scores)
        for fidx in range(Z):
            metrics[fidx] = {}
            for mi in range(n_masks):
                ps = float(bounds[fidx, mi])
                metrics[fidx][mi + 1] = {"presence_score": ps}
                if ps < min_presence_score:
                    vol[fidx][vol[fidx] == mi + 1] = 0
    return vol.astype(np.uint16), metrics, frame_scores
