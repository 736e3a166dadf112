"""ORACLE (test infrastructure, never shipped): plain-PyTorch fp32 CPU restatement of the
SAM2.1 *image* model executed by the reference's hot path.

Parity status: **parity unpinned at the `sam2` boundary** - the reference's arithmetic
lives in the third-party package `sam2` (facebookresearch/sam2, declared unpinned at
/root/reference/pyproject.toml:26), which is absent from /root/reference and not
installed; the reference's tests pin no numeric output of this path.
This file restates the published SAM2.1 algorithm and is cross-checked against the
independent `transformers==5.15.0` Sam2Model restatement by oracle/hf_crosscheck.py
(identical seeded weights through a key map; max |diff| ~1e-5).

Call sites of the reference that reach this arithmetic:
  saber/adapters/sam2/predictor.py:48-70  (segment_image_2d -> generate)
  saber/adapters/sam2/automask.py:55-78   (build_sam2 + SAM2AutomaticMaskGenerator)

Weights are a flat dict under the UPSTREAM checkpoint key names (saber_amd/weights.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from saber_amd.model_config import (HieraConfig, DEC_DIM, DEC_HEADS, DEC_DEPTH, NUM_MASK_TOKENS,
                                    IMAGE_MEAN, IMAGE_STD, DYN_MULTIMASK_DELTA, DYN_MULTIMASK_THRESH)


def to_torch(weights: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.asarray(v)).float() for k, v in weights.items()}


# ----------------------------------------------------------------------------- transforms (b1)
def sam2_transforms(image: np.ndarray, resolution: int = 1024) -> torch.Tensor:
    """SAM2Transforms: ToTensor (float input: no /255) -> bilinear Resize(res,res)
    -> Normalize(ImageNet mean/std).  image: (h,w,3) float32 -> (1,3,res,res)."""
    x = torch.from_numpy(np.ascontiguousarray(image)).float().permute(2, 0, 1)[None]
    if x.shape[-2:] != (resolution, resolution):
        x = F.interpolate(x, size=(resolution, resolution), mode="bilinear", align_corners=False, antialias=True)
    mean = torch.tensor(IMAGE_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGE_STD).view(1, 3, 1, 1)
    return (x - mean) / std


# ----------------------------------------------------------------------------- Hiera trunk (b2-b5)
def _window_partition(x, ws):
    B, H, W, C = x.shape
    ph, pw = (-H) % ws, (-W) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)
    return x, (Hp, Wp)


def _window_unpartition(w, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // ((Hp // ws) * (Wp // ws))
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W].contiguous()


def _pool(x, s):
    return F.max_pool2d(x.permute(0, 3, 1, 2), kernel_size=s, stride=s).permute(0, 2, 3, 1)


def hiera_pos_embed(W, cfg: HieraConfig, hw) -> torch.Tensor:
    """b3: bicubic-interpolated background + tiled window embedding -> (1,h,w,C)."""
    t = "image_encoder.trunk."
    pe = F.interpolate(W[t + "pos_embed"], size=hw, mode="bicubic")
    win = W[t + "pos_embed_window"]
    pe = pe + win.tile([x // y for x, y in zip(pe.shape, win.shape)])
    return pe.permute(0, 2, 3, 1)


def hiera_block(W, i, spec, x, eps, taps=None):
    din, dout, heads, win, qs = spec
    p = f"image_encoder.trunk.blocks.{i}."
    shortcut = x
    x = F.layer_norm(x, (din,), W[p + "norm1.weight"], W[p + "norm1.bias"], eps)
    if din != dout:
        shortcut = _pool(F.linear(x, W[p + "proj.weight"], W[p + "proj.bias"]), qs)
    H, Wd = x.shape[1:3]
    if win > 0:
        x, pad_hw = _window_partition(x, win)
    B, h, w, _ = x.shape
    qkv = F.linear(x, W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"]).reshape(B, h * w, 3, heads, -1)
    q, k, v = qkv.unbind(2)
    if qs > 1:
        q = _pool(q.reshape(B, h, w, -1), qs)
        h, w = q.shape[1:3]
        q = q.reshape(B, h * w, heads, -1)
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    a = torch.softmax((q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5), dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, h, w, -1)
    a = F.linear(a, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"])
    if qs > 1:
        wsz = win // qs
        H, Wd = shortcut.shape[1:3]
        pad_hw = (H + (-H) % wsz, Wd + (-Wd) % wsz) if win > 0 else None
    else:
        wsz = win
    if win > 0:
        a = _window_unpartition(a, wsz, pad_hw, (H, Wd))
    x = shortcut + a
    y = F.layer_norm(x, (dout,), W[p + "norm2.weight"], W[p + "norm2.bias"], eps)
    y = F.linear(y, W[p + "mlp.layers.0.weight"], W[p + "mlp.layers.0.bias"])
    y = F.gelu(y)
    y = F.linear(y, W[p + "mlp.layers.1.weight"], W[p + "mlp.layers.1.bias"])
    return x + y


def hiera_trunk(W, cfg: HieraConfig, pixels: torch.Tensor, taps: Optional[dict] = None) -> List[torch.Tensor]:
    """pixels (B,3,1024,1024) -> list of 4 stage outputs, NHWC."""
    t = "image_encoder.trunk."
    x = F.conv2d(pixels, W[t + "patch_embed.proj.weight"], W[t + "patch_embed.proj.bias"], stride=4, padding=3)
    x = x.permute(0, 2, 3, 1)
    x = x + hiera_pos_embed(W, cfg, x.shape[1:3])
    if taps is not None:
        taps["patch_embed"] = x
    outs = []
    ends = cfg.stage_ends
    for i, spec in enumerate(cfg.block_specs()):
        x = hiera_block(W, i, spec, x, cfg.ln_eps)
        if taps is not None and ("blocks" in taps):
            taps["blocks"].append(x)
        if i in ends:
            outs.append(x)
    return outs


def fpn_neck(W, cfg: HieraConfig, stage_outs: List[torch.Tensor]) -> List[torch.Tensor]:
    """b6: lateral 1x1 convs; nearest-2x top-down add on levels in fpn_top_down_levels;
    returns [level0 (256^2), level1 (128^2), level2 (64^2)] NCHW (scalp=1 drops 32^2)."""
    n = len(stage_outs) - 1
    prev = None
    out = [None] * (n + 1)
    for i in range(n, -1, -1):
        x = stage_outs[i].permute(0, 3, 1, 2)
        lat = F.conv2d(x, W[f"image_encoder.neck.convs.{n - i}.conv.weight"], W[f"image_encoder.neck.convs.{n - i}.conv.bias"])
        if i in cfg.fpn_top_down_levels and prev is not None:
            prev = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        else:
            prev = lat
        out[i] = prev
    return out[:-1]


def encode_image(W, cfg: HieraConfig, pixels: torch.Tensor, taps: Optional[dict] = None):
    """forward_image + the image-predictor's feature prep (b2-b7).
    Returns dict: image_embed (B,256,64,64), feat_s0 (B,32,256,256), feat_s1 (B,64,128,128)."""
    fpn = fpn_neck(W, cfg, hiera_trunk(W, cfg, pixels, taps))
    d = "sam_mask_decoder."
    s0 = F.conv2d(fpn[0], W[d + "conv_s0.weight"], W[d + "conv_s0.bias"])
    s1 = F.conv2d(fpn[1], W[d + "conv_s1.weight"], W[d + "conv_s1.bias"])
    emb = fpn[2] + W["no_mem_embed"].view(1, -1, 1, 1)
    return {"image_embed": emb, "feat_s0": s0, "feat_s1": s1}


# ----------------------------------------------------------------------------- prompt encoder (b8)
def _pe_encoding(W, coords01: torch.Tensor) -> torch.Tensor:
    G = W["sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = (2 * coords01 - 1) @ G
    c = 2 * math.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(W, size: int = 64) -> torch.Tensor:
    """get_dense_pe(): (1,256,size,size)."""
    g = (torch.arange(size, dtype=torch.float32) + 0.5) / size
    yy, xx = torch.meshgrid(g, g, indexing="ij")
    pe = _pe_encoding(W, torch.stack([xx, yy], dim=-1))
    return pe.permute(2, 0, 1)[None]


def _ln2d(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[None, :, None, None] * x + b[None, :, None, None]


def prompt_encoder(W, points: torch.Tensor, labels: torch.Tensor, mask_input: Optional[torch.Tensor], res: int = 1024):
    """points (B,N,2) in model pixels, labels (B,N) -> sparse (B,N+1,256), dense (B,256,64,64)."""
    p = "sam_prompt_encoder."
    B = points.shape[0]
    pts = points + 0.5
    pts = torch.cat([pts, torch.zeros(B, 1, 2)], dim=1)
    lab = torch.cat([labels, -torch.ones(B, 1, dtype=labels.dtype)], dim=1)
    emb = _pe_encoding(W, pts / res)
    emb = torch.where((lab == -1)[..., None], torch.zeros_like(emb), emb)
    emb = emb + torch.where((lab == -1)[..., None], W[p + "not_a_point_embed.weight"], torch.zeros_like(emb))
    for k in range(4):
        emb = emb + torch.where((lab == k)[..., None], W[p + f"point_embeddings.{k}.weight"], torch.zeros_like(emb))
    if mask_input is None:
        dense = W[p + "no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(B, -1, res // 16, res // 16)
    else:
        m = p + "mask_downscaling."
        x = F.conv2d(mask_input, W[m + "0.weight"], W[m + "0.bias"], stride=2)
        x = F.gelu(_ln2d(x, W[m + "1.weight"], W[m + "1.bias"]))
        x = F.conv2d(x, W[m + "3.weight"], W[m + "3.bias"], stride=2)
        x = F.gelu(_ln2d(x, W[m + "4.weight"], W[m + "4.bias"]))
        dense = F.conv2d(x, W[m + "6.weight"], W[m + "6.bias"])
    return emb, dense


# ----------------------------------------------------------------------------- mask decoder (b9, b10)
def _attn(W, prefix, q, k, v, heads=DEC_HEADS):
    q = F.linear(q, W[prefix + ".q_proj.weight"], W[prefix + ".q_proj.bias"])
    k = F.linear(k, W[prefix + ".k_proj.weight"], W[prefix + ".k_proj.bias"])
    v = F.linear(v, W[prefix + ".v_proj.weight"], W[prefix + ".v_proj.bias"])
    B, Nq, C = q.shape
    hd = C // heads
    q = q.view(B, Nq, heads, hd).transpose(1, 2)
    k = k.view(B, -1, heads, hd).transpose(1, 2)
    v = v.view(B, -1, heads, hd).transpose(1, 2)
    a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, Nq, C)
    return F.linear(a, W[prefix + ".out_proj.weight"], W[prefix + ".out_proj.bias"])


def _ln(W, prefix, x):
    return F.layer_norm(x, (x.shape[-1],), W[prefix + ".weight"], W[prefix + ".bias"], 1e-5)


def _mlp(W, prefix, x, n, sigmoid=False):
    for i in range(n):
        x = F.linear(x, W[f"{prefix}.layers.{i}.weight"], W[f"{prefix}.layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return torch.sigmoid(x) if sigmoid else x


def two_way_transformer(W, src, pos, tokens):
    """src,pos (B,4096,256); tokens (B,T,256) -> (queries, keys)."""
    t = "sam_mask_decoder.transformer."
    queries, keys = tokens, src
    for l in range(DEC_DEPTH):
        L = f"{t}layers.{l}."
        if l == 0:
            queries = _attn(W, L + "self_attn", queries, queries, queries)
        else:
            q = queries + tokens
            queries = queries + _attn(W, L + "self_attn", q, q, queries)
        queries = _ln(W, L + "norm1", queries)
        q = queries + tokens
        k = keys + pos
        queries = queries + _attn(W, L + "cross_attn_token_to_image", q, k, keys)
        queries = _ln(W, L + "norm2", queries)
        queries = queries + _mlp(W, L + "mlp", queries, 2)
        queries = _ln(W, L + "norm3", queries)
        q = queries + tokens
        k = keys + pos
        keys = keys + _attn(W, L + "cross_attn_image_to_token", k, q, queries)
        keys = _ln(W, L + "norm4", keys)
    q = queries + tokens
    k = keys + pos
    queries = queries + _attn(W, t + "final_attn_token_to_image", q, k, keys)
    queries = _ln(W, t + "norm_final_attn", queries)
    return queries, keys


def _stability(logits, delta):
    f = logits.flatten(-2)
    ai = (f > delta).sum(-1).float()
    au = (f > -delta).sum(-1).float()
    return torch.where(au > 0, ai / au, torch.ones_like(au))


def mask_decoder(W, feats, sparse, dense, multimask_output: bool, pos: Optional[torch.Tensor] = None, return_tokens: bool = False):
    """feats: encode_image() dict for ONE image (B=1). sparse (P,T,256), dense (P|1,256,64,64).
    Returns low_res (P,M,256,256), iou (P,M), obj (P,1), all_masks (P,4,256,256), all_iou (P,4)
    [+ the 4 mask tokens after the transformer (P,4,256) with return_tokens: the video path projects one of them to the object pointer]."""
    d = "sam_mask_decoder."
    P = sparse.shape[0]
    out_tok = torch.cat([W[d + "obj_score_token.weight"], W[d + "iou_token.weight"], W[d + "mask_tokens.weight"]], 0)
    tokens = torch.cat([out_tok[None].expand(P, -1, -1), sparse], dim=1)
    src = feats["image_embed"].expand(P, -1, -1, -1) + dense
    if pos is None:
        pos = dense_pe(W, src.shape[-1])
    b, c, h, w = src.shape
    hs, keys = two_way_transformer(W, src.flatten(2).transpose(1, 2), pos.flatten(2).transpose(1, 2).expand(P, -1, -1), tokens)
    iou_tok = hs[:, 1]
    mask_toks = hs[:, 2:2 + NUM_MASK_TOKENS]
    src = keys.transpose(1, 2).reshape(b, c, h, w)
    up = F.conv_transpose2d(src, W[d + "output_upscaling.0.weight"], W[d + "output_upscaling.0.bias"], stride=2)
    up = F.gelu(_ln2d(up + feats["feat_s1"], W[d + "output_upscaling.1.weight"], W[d + "output_upscaling.1.bias"]))
    up = F.conv_transpose2d(up, W[d + "output_upscaling.3.weight"], W[d + "output_upscaling.3.bias"], stride=2)
    up = F.gelu(up + feats["feat_s0"])
    hyper = torch.stack([_mlp(W, f"{d}output_hypernetworks_mlps.{i}", mask_toks[:, i], 3) for i in range(NUM_MASK_TOKENS)], 1)
    bb, cc, hh, ww = up.shape
    masks = (hyper @ up.view(bb, cc, hh * ww)).view(bb, -1, hh, ww)
    iou = _mlp(W, d + "iou_prediction_head", iou_tok, 3, sigmoid=True)
    obj = _mlp(W, d + "pred_obj_score_head", hs[:, 0], 3)
    all_masks, all_iou = masks, iou
    if multimask_output:
        masks, iou = masks[:, 1:], iou[:, 1:]
    else:
        # dynamic_multimask_via_stability (apply_postprocessing=True, automask.py:62)
        best = torch.argmax(all_iou[:, 1:], dim=-1)
        ar = torch.arange(P)
        best_masks = all_masks[:, 1:][ar, best][:, None]
        best_iou = all_iou[:, 1:][ar, best][:, None]
        single, single_iou = all_masks[:, 0:1], all_iou[:, 0:1]
        stable = _stability(single, DYN_MULTIMASK_DELTA) >= DYN_MULTIMASK_THRESH
        masks = torch.where(stable[..., None, None], single, best_masks)
        iou = torch.where(stable, single_iou, best_iou)
    if return_tokens:
        return masks, iou, obj, all_masks, all_iou, mask_toks
    return masks, iou, obj, all_masks, all_iou


# ----------------------------------------------------------------------------- image predictor
class ImagePredictorRef:
    """Restatement of SAM2ImagePredictor.set_image/_predict for one image (b1, b11)."""

    def __init__(self, weights: Dict[str, np.ndarray], cfg: HieraConfig):
        self.W = to_torch(weights)
        self.cfg = cfg
        self.res = cfg.image_size
        self._pos = dense_pe(self.W, self.res // 16)
        self.feats = None
        self.orig_hw = None

    @torch.no_grad()
    def set_image(self, image: np.ndarray):
        self.orig_hw = image.shape[:2]
        self.feats = encode_image(self.W, self.cfg, sam2_transforms(image, self.res))

    def reset_predictor(self):
        self.feats, self.orig_hw = None, None

    def transform_coords(self, coords: torch.Tensor, normalize: bool, orig_hw) -> torch.Tensor:
        if normalize:
            h, w = orig_hw
            coords = coords.clone()
            coords[..., 0] = coords[..., 0] / w
            coords[..., 1] = coords[..., 1] / h
        return coords * self.res

    @torch.no_grad()
    def predict_lowres(self, pts, labels, mask_input=None, multimask_output=True):
        sparse, dense = prompt_encoder(self.W, pts, labels, mask_input, self.res)
        return mask_decoder(self.W, self.feats, sparse, dense, multimask_output, self._pos)

    @torch.no_grad()
    def _predict(self, pts, labels, mask_input=None, multimask_output=True):
        low, iou, obj, _, _ = self.predict_lowres(pts, labels, mask_input, multimask_output)
        masks = F.interpolate(low, self.orig_hw, mode="bilinear", align_corners=False)
        return masks, iou, torch.clamp(low, -32.0, 32.0)
