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
temporal encodings, occlusion logic) is not restated yet.
"""
import math
from typing import Dict

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
