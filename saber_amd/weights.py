"""Weights for the SAM2.1 image model under the UPSTREAM checkpoint key names.

Two sources:

* ``load_checkpoint(path)`` - a real ``sam2.1_hiera_*.pt`` (``torch.load(path)["model"]``,
  the file the reference downloads in saber/pretrained_weights.py:174-202).  Not
  available offline, so this path is exercised only by a round-trip test.
* ``seeded_weights(cfg, seed)`` - deterministic synthetic weights generated per tensor
  from a counter-based RNG keyed on (seed, crc32(name)).  A Hiera-L state dict is
  ~850 MB and cannot be a fixture; the GPU box regenerates the identical tensors from
  the seed.  Gains are chosen so activations stay O(1)-O(10) through 48 blocks and the
  mask logits are large enough that the AMG filters (pred_iou 0.7 / stability 0.92,
  reference: saber/adapters/sam2/amg.py:7-17) keep a non-trivial set of masks.
"""
import zlib
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np

from .model_config import HieraConfig, DEC_DIM, DEC_MLP, DEC_DEPTH, NUM_MASK_TOKENS


MEM_DIM = 64          # channels of a spatial memory / of an object-pointer token
NUM_MASKMEM_CKPT = 7  # rows of maskmem_tpos_enc in the checkpoints (the reference keeps the first `num_maskmem`, default 2)


def video_param_specs() -> "OrderedDict[str, Tuple[tuple, str, float]]":
    """The tensors the video (memory) path adds to the image model, under their upstream checkpoint keys: memory attention,
    memory encoder, object-pointer heads and the learned 'no memory / no object' embeddings (sam2.1 configs: 4 memory-attention
    layers, d_model 256, memory dim 64, 2 fuser blocks).  Same for every trunk."""
    P: "OrderedDict[str, Tuple[tuple, str, float]]" = OrderedDict()

    def lin(prefix, out_f, in_f, gain=1.0):
        P[prefix + ".weight"] = ((out_f, in_f), "w", gain)
        P[prefix + ".bias"] = ((out_f,), "b", 1.0)

    def ln(prefix, c):
        P[prefix + ".weight"] = ((c,), "ln_w", 1.0)
        P[prefix + ".bias"] = ((c,), "ln_b", 1.0)

    for i in range(4):
        L = f"memory_attention.layers.{i}."
        for a, kv in (("self_attn", DEC_DIM), ("cross_attn_image", MEM_DIM)):
            lin(L + a + ".q_proj", DEC_DIM, DEC_DIM)
            lin(L + a + ".k_proj", DEC_DIM, kv)
            lin(L + a + ".v_proj", DEC_DIM, kv)
            lin(L + a + ".out_proj", DEC_DIM, DEC_DIM, 0.5)
        lin(L + "linear1", DEC_MLP, DEC_DIM)
        lin(L + "linear2", DEC_DIM, DEC_MLP, 0.5)
        for n in ("norm1", "norm2", "norm3"):
            ln(L + n, DEC_DIM)
    ln("memory_attention.norm", DEC_DIM)
    me = "memory_encoder."
    chans = [1, 4, 16, 64, 256]
    for j in range(4):
        P[f"{me}mask_downsampler.encoder.{3 * j}.weight"] = ((chans[j + 1], chans[j], 3, 3), "w", 1.5)
        P[f"{me}mask_downsampler.encoder.{3 * j}.bias"] = ((chans[j + 1],), "b", 1.0)
        ln(f"{me}mask_downsampler.encoder.{3 * j + 1}", chans[j + 1])
    P[me + "mask_downsampler.encoder.12.weight"] = ((DEC_DIM, DEC_DIM, 1, 1), "w", 1.0)
    P[me + "mask_downsampler.encoder.12.bias"] = ((DEC_DIM,), "b", 1.0)
    P[me + "pix_feat_proj.weight"] = ((DEC_DIM, DEC_DIM, 1, 1), "w", 1.0)
    P[me + "pix_feat_proj.bias"] = ((DEC_DIM,), "b", 1.0)
    for i in range(2):
        f = f"{me}fuser.layers.{i}."
        P[f + "dwconv.weight"] = ((DEC_DIM, 1, 7, 7), "w", 1.0)
        P[f + "dwconv.bias"] = ((DEC_DIM,), "b", 1.0)
        ln(f + "norm", DEC_DIM)
        lin(f + "pwconv1", 4 * DEC_DIM, DEC_DIM)
        lin(f + "pwconv2", DEC_DIM, 4 * DEC_DIM, 0.5)
        P[f + "gamma"] = ((DEC_DIM,), "emb", 0.5)          # layer scale (1e-6 at initialisation upstream; trained values are O(0.1-1))
    P[me + "out_proj.weight"] = ((MEM_DIM, DEC_DIM, 1, 1), "w", 1.0)
    P[me + "out_proj.bias"] = ((MEM_DIM,), "b", 1.0)
    P["maskmem_tpos_enc"] = ((NUM_MASKMEM_CKPT, 1, 1, MEM_DIM), "emb", 0.3)
    P["no_mem_pos_enc"] = ((1, 1, DEC_DIM), "emb", 0.2)
    P["no_obj_ptr"] = ((1, DEC_DIM), "emb", 0.5)
    P["no_obj_embed_spatial"] = ((1, MEM_DIM), "emb", 0.3)
    for l in range(3):
        lin(f"obj_ptr_proj.layers.{l}", DEC_DIM, DEC_DIM, 1.4)
    lin("obj_ptr_tpos_proj", MEM_DIM, DEC_DIM)
    P["mask_downsample.weight"] = ((1, 1, 4, 4), "w", 2.0)
    P["mask_downsample.bias"] = ((1,), "b", 1.0)
    return P


def param_specs(cfg: HieraConfig) -> "OrderedDict[str, Tuple[tuple, str, float]]":
    """name -> (shape, kind, gain).  kind in {w, b, ln_w, ln_b, emb, pe}."""
    P: "OrderedDict[str, Tuple[tuple, str, float]]" = OrderedDict()

    def lin(prefix, out_f, in_f, gain=1.0):
        P[prefix + ".weight"] = ((out_f, in_f), "w", gain)
        P[prefix + ".bias"] = ((out_f,), "b", 1.0)

    def ln(prefix, c):
        P[prefix + ".weight"] = ((c,), "ln_w", 1.0)
        P[prefix + ".bias"] = ((c,), "ln_b", 1.0)

    t = "image_encoder.trunk."
    C0 = cfg.embed_dim
    P[t + "patch_embed.proj.weight"] = ((C0, 3, 7, 7), "w", 1.0)
    P[t + "patch_embed.proj.bias"] = ((C0,), "b", 1.0)
    P[t + "pos_embed"] = ((1, C0) + tuple(cfg.pos_embed_bkg), "pe", 0.5)
    P[t + "pos_embed_window"] = ((1, C0, cfg.window_spec[0], cfg.window_spec[0]), "pe", 0.5)
    for i, (din, dout, heads, win, qs) in enumerate(cfg.block_specs()):
        b = f"{t}blocks.{i}."
        ln(b + "norm1", din)
        lin(b + "attn.qkv", 3 * dout, din)
        lin(b + "attn.proj", dout, dout, 0.5)
        ln(b + "norm2", dout)
        lin(b + "mlp.layers.0", 4 * dout, dout)
        lin(b + "mlp.layers.1", dout, 4 * dout, 0.5)
        if din != dout:
            lin(b + "proj", dout, din)
    chans = cfg.stage_dims[::-1]
    for n, c in enumerate(chans):
        P[f"image_encoder.neck.convs.{n}.conv.weight"] = ((cfg.fpn_dim, c, 1, 1), "w", 1.0)
        P[f"image_encoder.neck.convs.{n}.conv.bias"] = ((cfg.fpn_dim,), "b", 1.0)
    P["no_mem_embed"] = ((1, 1, DEC_DIM), "emb", 0.2)

    pe = "sam_prompt_encoder."
    P[pe + "pe_layer.positional_encoding_gaussian_matrix"] = ((2, DEC_DIM // 2), "pe", 1.0)
    for k in range(4):
        P[pe + f"point_embeddings.{k}.weight"] = ((1, DEC_DIM), "emb", 1.0)
    P[pe + "not_a_point_embed.weight"] = ((1, DEC_DIM), "emb", 1.0)
    P[pe + "no_mask_embed.weight"] = ((1, DEC_DIM), "emb", 0.5)
    P[pe + "mask_downscaling.0.weight"] = ((4, 1, 2, 2), "w", 0.3)
    P[pe + "mask_downscaling.0.bias"] = ((4,), "b", 1.0)
    ln(pe + "mask_downscaling.1", 4)
    P[pe + "mask_downscaling.3.weight"] = ((16, 4, 2, 2), "w", 1.0)
    P[pe + "mask_downscaling.3.bias"] = ((16,), "b", 1.0)
    ln(pe + "mask_downscaling.4", 16)
    P[pe + "mask_downscaling.6.weight"] = ((DEC_DIM, 16, 1, 1), "w", 0.5)
    P[pe + "mask_downscaling.6.bias"] = ((DEC_DIM,), "b", 1.0)

    d = "sam_mask_decoder."

    def attn(prefix, internal):
        lin(prefix + ".q_proj", internal, DEC_DIM)
        lin(prefix + ".k_proj", internal, DEC_DIM)
        lin(prefix + ".v_proj", internal, DEC_DIM)
        lin(prefix + ".out_proj", DEC_DIM, internal, 0.7)

    for l in range(DEC_DEPTH):
        L = f"{d}transformer.layers.{l}."
        attn(L + "self_attn", DEC_DIM)
        ln(L + "norm1", DEC_DIM)
        attn(L + "cross_attn_token_to_image", DEC_DIM // 2)
        ln(L + "norm2", DEC_DIM)
        lin(L + "mlp.layers.0", DEC_MLP, DEC_DIM)
        lin(L + "mlp.layers.1", DEC_DIM, DEC_MLP, 0.7)
        ln(L + "norm3", DEC_DIM)
        ln(L + "norm4", DEC_DIM)
        attn(L + "cross_attn_image_to_token", DEC_DIM // 2)
    attn(d + "transformer.final_attn_token_to_image", DEC_DIM // 2)
    ln(d + "transformer.norm_final_attn", DEC_DIM)
    P[d + "iou_token.weight"] = ((1, DEC_DIM), "emb", 1.0)
    P[d + "mask_tokens.weight"] = ((NUM_MASK_TOKENS, DEC_DIM), "emb", 1.0)
    P[d + "obj_score_token.weight"] = ((1, DEC_DIM), "emb", 1.0)
    # ConvTranspose2d weights are (C_in, C_out, kH, kW)
    P[d + "output_upscaling.0.weight"] = ((DEC_DIM, DEC_DIM // 4, 2, 2), "wT", 1.0)
    P[d + "output_upscaling.0.bias"] = ((DEC_DIM // 4,), "b", 1.0)
    ln(d + "output_upscaling.1", DEC_DIM // 4)
    P[d + "output_upscaling.3.weight"] = ((DEC_DIM // 4, DEC_DIM // 8, 2, 2), "wT", 1.0)
    P[d + "output_upscaling.3.bias"] = ((DEC_DIM // 8,), "b", 1.0)
    P[d + "conv_s0.weight"] = ((DEC_DIM // 8, DEC_DIM, 1, 1), "w", 1.0)
    P[d + "conv_s0.bias"] = ((DEC_DIM // 8,), "b", 1.0)
    P[d + "conv_s1.weight"] = ((DEC_DIM // 4, DEC_DIM, 1, 1), "w", 1.0)
    P[d + "conv_s1.bias"] = ((DEC_DIM // 4,), "b", 1.0)
    for k in range(NUM_MASK_TOKENS):
        h = f"{d}output_hypernetworks_mlps.{k}."
        lin(h + "layers.0", DEC_DIM, DEC_DIM, 1.4)
        lin(h + "layers.1", DEC_DIM, DEC_DIM, 1.4)
        lin(h + "layers.2", DEC_DIM // 8, DEC_DIM, 3.0)
    lin(d + "iou_prediction_head.layers.0", DEC_DIM, DEC_DIM, 1.4)
    lin(d + "iou_prediction_head.layers.1", DEC_DIM, DEC_DIM, 1.4)
    lin(d + "iou_prediction_head.layers.2", NUM_MASK_TOKENS, DEC_DIM, 2.0)
    lin(d + "pred_obj_score_head.layers.0", DEC_DIM, DEC_DIM, 1.4)
    lin(d + "pred_obj_score_head.layers.1", DEC_DIM, DEC_DIM, 1.4)
    lin(d + "pred_obj_score_head.layers.2", 1, DEC_DIM, 2.0)
    return P


def _gen(name: str, shape, kind: str, gain: float, seed: int) -> np.ndarray:
    key = (np.uint64(seed) << np.uint64(32)) | np.uint64(zlib.crc32(name.encode()))
    rng = np.random.Generator(np.random.Philox(key=int(key)))
    x = rng.standard_normal(size=shape, dtype=np.float32)
    if kind == "w":  # (out, in, ...) fan_in = prod(shape[1:])
        fan_in = int(np.prod(shape[1:]))
        x *= np.float32(gain / np.sqrt(fan_in))
    elif kind == "wT":  # ConvTranspose (in, out, kh, kw): each output pixel sees `in` taps
        x *= np.float32(gain / np.sqrt(shape[0]))
    elif kind == "b":
        x *= np.float32(0.1 * gain)
    elif kind == "ln_w":
        x = np.float32(1.0) + np.float32(0.1) * x
    elif kind == "ln_b":
        x *= np.float32(0.1)
    elif kind in ("emb", "pe"):
        x *= np.float32(gain)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(x, dtype=np.float32)


def seeded_weights(cfg: HieraConfig, seed: int = 0, video: bool = False) -> Dict[str, np.ndarray]:
    """Deterministic fp32 weights for every tensor of the image model (video=True: plus the memory path of the video predictor)."""
    specs = param_specs(cfg)
    if video:
        specs = OrderedDict(list(specs.items()) + list(video_param_specs().items()))
    return OrderedDict((n, _gen(n, s, k, g, seed)) for n, (s, k, g) in specs.items())


def fitted_decoder_weights(cfg: HieraConfig, seed: int = 0, video: bool = False, path: Optional[str] = None) -> Dict[str, np.ndarray]:
    """seeded_weights(cfg, seed) with the mask decoder (and the prompt encoder's mask-input branch) replaced by the tensors that
    oracle/fit_decoder_heads.py fitted on the synthetic slices (tests/golden/decoder_fit_large_seed0.npz, float16): the seeded Hiera-L
    encoder with a decoder whose masks are compact objects with a spread of predicted IoU / stability, so that cfgAMG's OWN thresholds and
    both NMS stages (saber/adapters/sam2/amg.py:7-17) leave a non-trivial set of masks on BASELINE configs[1]'s slice.  Not a checkpoint:
    a fixture that gives the generator's filters, the duplicate removal and the paint order something to do (no weights ship offline)."""
    import os
    if cfg.name != "large" or seed != 0:
        raise ValueError("the fitted decoder exists for the seeded Hiera-L model (seed 0) only")
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "decoder_fit_large_seed0.npz")
    W = seeded_weights(cfg, seed, video)
    with np.load(path) as Z:
        for k in Z.files:
            if k.startswith("__"):
                continue
            if k not in W or W[k].shape != Z[k].shape:
                raise ValueError(f"fitted decoder tensor '{k}' does not match the model")
            W[k] = np.ascontiguousarray(Z[k], dtype=np.float32)
    return W


def stress_weights(cfg: HieraConfig, seed: int = 0, gamma_outlier: float = 30.0, massive: float = 200.0) -> Dict[str, np.ndarray]:
    """seeded_weights with the activation statistics trained ViT checkpoints are known for and random initialisation lacks (no real
    SAM2.1 checkpoint can be fetched offline): (a) outlier LayerNorm gains - two channels of every norm1 / norm2
    scaled by `gamma_outlier`, so two input columns of each qkv / fc1 GEMM carry values ~30x the rest; (b) two "massive" residual
    channels - in the second block of stages 1 and 2 two output rows of mlp.layers.1 are scaled by `massive` and biased, which leaves
    |x| ~ 100-1000 in those channels of the residual stream for the rest of the trunk (every later LayerNorm's variance is then set
    by two channels); (c) a 5x larger background pos_embed.  Used by tests only: the precision modes are compared on it."""
    W = seeded_weights(cfg, seed)
    rng = np.random.default_rng(1000 + seed)
    t = "image_encoder.trunk."
    specs = cfg.block_specs()
    for i, (din, dout, heads, win, qs) in enumerate(specs):
        for nm, c in (("norm1", din), ("norm2", dout)):
            ch = rng.choice(c, 2, replace=False)
            W[f"{t}blocks.{i}.{nm}.weight"][ch] *= np.float32(gamma_outlier)
    ends = cfg.stage_ends
    for st in (1, 2):
        i = ends[st - 1] + 2                       # second block of the stage
        dout = specs[i][1]
        ch = rng.choice(dout, 2, replace=False)
        W[f"{t}blocks.{i}.mlp.layers.1.weight"][ch] *= np.float32(massive)
        W[f"{t}blocks.{i}.mlp.layers.1.bias"][ch] += np.float32(massive) * np.float32([1.0, -1.0])
    W[t + "pos_embed"] *= np.float32(5.0)
    return W


def load_checkpoint(path: str, cfg: HieraConfig, video: bool = False) -> Dict[str, np.ndarray]:
    """Read an upstream ``sam2.1_hiera_*.pt`` and keep the image-model tensors.

    Raises ValueError naming any tensor that is missing or mis-shaped, so a wrong
    trunk/checkpoint pairing fails loudly (reference behaviour: hydra instantiation
    fails on a mismatched state dict, saber/adapters/sam2/automask.py:61-63)."""
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "model" in sd:
        sd = sd["model"]
    out = OrderedDict()
    specs = param_specs(cfg)
    if video:
        specs = OrderedDict(list(specs.items()) + list(video_param_specs().items()))
    for name, (shape, _, _) in specs.items():
        if name not in sd:
            raise ValueError(f"checkpoint {path} lacks tensor '{name}'")
        t = sd[name].detach().to(torch.float32).cpu().numpy()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"checkpoint tensor '{name}' has shape {tuple(t.shape)}, expected {tuple(shape)}")
        out[name] = np.ascontiguousarray(t)
    return out


def count_params(cfg: HieraConfig, prefix: str = "") -> int:
    return sum(int(np.prod(s)) for n, (s, _, _) in param_specs(cfg).items() if n.startswith(prefix))
