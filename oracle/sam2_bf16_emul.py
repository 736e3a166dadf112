"""ORACLE (test infrastructure, never shipped): the SAM2.1 image model of oracle/sam2_ref.py restated WITH THE
ENGINE'S SANCTIONED PRECISION: every operand of a matrix product (linear, 1x1 conv, QK^T, PV, the folded
cross-attention products of the mask decoder) is rounded to bf16 exactly where saber_amd/csrc rounds it, all
accumulation / LayerNorm / softmax statistics / residual streams stay fp32.

Purpose ("make parity mean something"): the engine differs from the fp32 oracle (sam2_ref.py) by
~6e-3 relative RMS.  That residual is either the sanctioned bf16 operand rounding or a kernel defect.  This file
separates the two: it is an independent torch restatement that applies the same roundings, so
    engine  vs  this file   must agree to ~1e-3 (fp32 summation order + rare rounding flips only), while
    this file  vs  sam2_ref shows the same ~6e-3 the engine shows.
Parity status of the underlying algorithm: unchanged from sam2_ref.py (**parity unpinned at the `sam2` boundary**,
cross-checked against the independent HF restatement).  Only Hiera-L (no window padding) is restated here.

Where the engine rounds (file:line in saber_amd/csrc):
  * LayerNorm output -> bf16 (layernorm.hip), GEMM operands bf16, fp32 accumulate, bias / GELU / residual in fp32
    (gemm.hip epilogues); qkv and the MLP hidden are stored bf16; GELU is the fitted x*sigmoid(x q(x^2)) form (common.h:41).
  * Hiera attention: S = K.Q^T in fp32, P = bf16(exp2(s*sc - m)) UN-normalised, O = (P.V) / sum(e) -> bf16;
    16/64-key windows in one pass (attention_hiera.hip:127-183), 256-key windows, the q-pooled 256-key block and the three
    global blocks with an online softmax over 128-key blocks in engine token order (attention_hiera.hip:473-554).
  * neck: conv_s0 / conv_s1 are composed with their lateral convs on the host in fp64 and stored bf16 (engine.hip:319-334).
  * decoder: the cross attentions are folded (decoder_fused.hip:3-14): Qt = bf16(s W_k^T q), PEK = bf16(pe W_k^T), the
    per-prompt image-token state X is bf16, P is bf16, ... restated below step by step.
Only tests/ and __graft_entry__.smoke() may import this file.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from saber_amd.model_config import HieraConfig, DEC_HEADS, DYN_MULTIMASK_DELTA, DYN_MULTIMASK_THRESH

LOG2E = 1.4426950408889634


# The 16-bit type every rounding point below rounds to: bf16 (the engine's default arithmetic) or IEEE half (SABER_PRECISION_FP16, round 4:
# the same kernels compiled for fp16 operands - same rounding POINTS, 10 mantissa bits instead of 7).  `with operand_type("fp16"):` switches
# it for the emulation calls inside the block.
_OPERAND_DTYPE = torch.bfloat16


class operand_type:
    def __init__(self, name: str):
        self.dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[name]

    def __enter__(self):
        global _OPERAND_DTYPE
        self.prev, _OPERAND_DTYPE = _OPERAND_DTYPE, self.dtype
        return self

    def __exit__(self, *exc):
        global _OPERAND_DTYPE
        _OPERAND_DTYPE = self.prev
        return False


def bf(x: torch.Tensor) -> torch.Tensor:
    """round to nearest-even to the 16-bit operand type (bf16 unless inside operand_type("fp16")), keep the value in fp32"""
    return x.to(_OPERAND_DTYPE).to(torch.float32)


def lin(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """engine GEMM: bf16 operands, fp32 accumulate, fp32 bias"""
    y = F.linear(bf(x), bf(w))
    return y if b is None else y + b


def gelu_fit(x: torch.Tensor) -> torch.Tensor:
    """common.h gelu_erf: x * sigmoid(x * q(x^2)) in the exp2 domain"""
    x2 = torch.clamp(x * x, max=50.0)
    q = x2 * 1.01426305e-3 + (-1.06775724e-1)
    q = q * x2 + (-2.30112134)
    return x / (1.0 + torch.exp2(x * q))


def ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# ----------------------------------------------------------------------------- engine token order (common.h:perm_index256)
def perm_index256(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    return ((((y >> 6) * 4 + (x >> 6)) << 12) | ((((y >> 3) & 7) * 8 + ((x >> 3) & 7)) << 6) |
            ((((y >> 2) & 1) * 2 + ((x >> 2) & 1)) << 4) | ((((y >> 1) & 1) * 2 + ((x >> 1) & 1)) << 2) | ((y & 1) * 2 + (x & 1)))


def perm_table(stage: int) -> torch.Tensor:
    """idx[y*g + x] = engine row of grid cell (y, x) of the stage-s grid (side g = 256 >> s)"""
    g = 256 >> stage
    yy, xx = np.mgrid[:g, :g]
    idx = perm_index256(yy << stage, xx << stage) >> (2 * stage)
    return torch.from_numpy(idx.reshape(-1).astype(np.int64))


def to_engine_order(x_nhwc: torch.Tensor, stage: int) -> torch.Tensor:
    """(B,g,g,C) -> (B,g*g,C) rows in engine order"""
    B, g, _, C = x_nhwc.shape
    idx = perm_table(stage)
    out = torch.empty(B, g * g, C, dtype=x_nhwc.dtype)
    out[:, idx] = x_nhwc.reshape(B, g * g, C)
    return out


def from_engine_order(x: torch.Tensor, stage: int) -> torch.Tensor:
    """(B,g*g,C) engine order -> (B,C,g,g)"""
    g = 256 >> stage
    idx = perm_table(stage)
    return x[:, idx].reshape(x.shape[0], g, g, -1).permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------- attention with the engine's P rounding
def attn_emul(q, k, v, scale: float, block: Optional[int]):
    """q (G,nq,hd), k, v (G,nk,hd), all bf16-valued.  block None: one-pass softmax; else online softmax over key blocks."""
    sc = scale * LOG2E
    s = torch.matmul(q, k.transpose(-1, -2))              # fp32 accumulate of bf16 products
    if block is None or k.shape[1] <= block:
        m = s.max(-1, keepdim=True).values
        e = torch.exp2(s * sc - m * sc)
        o = torch.matmul(bf(e), v)
        return bf(o * (1.0 / e.sum(-1, keepdim=True)))
    nk = k.shape[1]
    m = torch.full(s.shape[:-1] + (1,), -3.0e38)
    l = torch.zeros_like(m)
    o = torch.zeros(q.shape[0], q.shape[1], v.shape[-1])
    for k0 in range(0, nk, block):
        sb = s[..., k0:k0 + block]
        mn = torch.maximum(m, sb.max(-1, keepdim=True).values * sc)
        alpha = torch.exp2(m - mn)
        e = torch.exp2(sb * sc - mn)
        l = l * alpha + e.sum(-1, keepdim=True)
        o = o * alpha + torch.matmul(bf(e), v[:, k0:k0 + block])
        m = mn
    return bf(o * (1.0 / l))


# ----------------------------------------------------------------------------- Hiera-L encoder
def lin_mx(x_mx: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """engine MXFP8 GEMM (gemm_fp8.hip): both operands already carry MX-quantised values (exact products, fp32 accumulate), fp32 bias"""
    y = F.linear(x_mx, w)
    return y if b is None else y + b


def hiera_block_emul(W, i, spec, x, tokens_per_image, eps, mx: bool = False, embed_dim: int = 144):
    """x: (B, N, din) fp32 residual stream, rows in engine order.  Returns (B, Nq, dout).
    mx: the MXFP8 weight format (engine.hip eng_encode): in stages 2 / 3 the LayerNorm outputs feeding qkv (blocks that keep their width)
    and mlp.layers.0, and the GELU output feeding mlp.layers.1, are MX-quantised (oracle/fp8_ref.mx_quantise) instead of bf16-rounded;
    W must hold the MX-quantised weights (fp8_ref.mx_quantise_encoder_weights)."""
    din, dout, heads, win, qs = spec
    p = f"image_encoder.trunk.blocks.{i}."
    B, N, _ = x.shape
    hd = dout // heads
    mx_mlp = mx and dout >= 4 * embed_dim
    mx_qkv = mx_mlp and din == dout
    if mx_mlp:
        from oracle.fp8_ref import mx_quantise
    if mx_qkv:
        xn8 = mx_quantise(ln(x, W[p + "norm1.weight"], W[p + "norm1.bias"], eps))[0]
        qkv = bf(lin_mx(xn8, W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"]))
        shortcut = x
        return _hiera_block_tail(W, p, spec, x, shortcut, qkv, eps, mx_mlp)
    xn = bf(ln(x, W[p + "norm1.weight"], W[p + "norm1.bias"], eps))
    if din != dout:
        sc = lin(xn, W[p + "proj.weight"], W[p + "proj.bias"])                       # fp32
        shortcut = sc.view(B, N // 4, 4, dout).max(2).values                        # 2x2 pool = 4 consecutive rows
    else:
        shortcut = x
    qkv = bf(lin(xn, W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"]))             # stored bf16
    return _hiera_block_tail(W, p, spec, x, shortcut, qkv, eps, mx_mlp)


def _hiera_block_tail(W, p, spec, x, shortcut, qkv, eps, mx_mlp):
    din, dout, heads, win, qs = spec
    B, N, _ = x.shape
    hd = dout // heads
    nk = win * win if win > 0 else N
    nwin = N // nk
    qkv = qkv.view(B * nwin, nk, 3, heads, hd)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    if qs > 1:
        q = q.reshape(B * nwin, nk // 4, 4, heads, hd).max(2).values                # q-pool on the bf16 values (exact)
    nq = q.shape[1]
    q = q.permute(0, 2, 1, 3).reshape(-1, nq, hd)
    k = k.permute(0, 2, 1, 3).reshape(-1, nk, hd)
    v = v.permute(0, 2, 1, 3).reshape(-1, nk, hd)
    block = None if nk <= 64 else 128       # small kernels: one pass; win256 / large / stream kernels: 128-key blocks
    a = attn_emul(q, k, v, hd ** -0.5, block)
    a = a.view(B * nwin, heads, nq, hd).permute(0, 2, 1, 3).reshape(B, nwin * nq, dout)
    x = shortcut + lin(a, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"])
    if mx_mlp:
        from oracle.fp8_ref import mx_quantise
        y = mx_quantise(ln(x, W[p + "norm2.weight"], W[p + "norm2.bias"], eps))[0]
        h = mx_quantise(gelu_fit(lin_mx(y, W[p + "mlp.layers.0.weight"], W[p + "mlp.layers.0.bias"])))[0]
        return x + lin_mx(h, W[p + "mlp.layers.1.weight"], W[p + "mlp.layers.1.bias"])
    y = bf(ln(x, W[p + "norm2.weight"], W[p + "norm2.bias"], eps))
    h = bf(gelu_fit(lin(y, W[p + "mlp.layers.0.weight"], W[p + "mlp.layers.0.bias"])))
    return x + lin(h, W[p + "mlp.layers.1.weight"], W[p + "mlp.layers.1.bias"])


@torch.no_grad()
def encode_image_emul(W: Dict[str, torch.Tensor], cfg: HieraConfig, pixels: torch.Tensor, mx: bool = False, taps: Optional[list] = None):
    """pixels (B,3,1024,1024) fp32 -> dict image_embed (B,256,64,64), feat_s0 (B,32,256,256), feat_s1 (B,64,128,128).
    mx: emulate the MXFP8 weight format (hiera_block_emul); taps: list that receives the residual stream after every block."""
    assert cfg.name == "large", "the bf16-emulating restatement covers Hiera-L only (no window padding)"
    from oracle.sam2_ref import hiera_pos_embed
    t = "image_encoder.trunk."
    x = F.conv2d(pixels, W[t + "patch_embed.proj.weight"], W[t + "patch_embed.proj.bias"], stride=4, padding=3)   # fp32 direct conv
    x = x.permute(0, 2, 3, 1)
    x = x + hiera_pos_embed(W, cfg, x.shape[1:3])
    x = to_engine_order(x, 0)
    outs = []
    for i, spec in enumerate(cfg.block_specs()):
        x = hiera_block_emul(W, i, spec, x, x.shape[1], cfg.ln_eps, mx=mx, embed_dim=cfg.embed_dim)
        if taps is not None:
            taps.append(x)
        if i in cfg.stage_ends:
            outs.append(bf(x))                                                          # sb[stage]: bf16 copy of the stream
    nk = "image_encoder.neck.convs."
    d = "sam_mask_decoder."

    def w2(name):
        return W[name].flatten(1)

    lat3 = lin(outs[3], w2(nk + "0.conv.weight"), W[nk + "0.conv.bias"])                 # (B,1024,256) fp32
    emb = lin(outs[2], w2(nk + "1.conv.weight"), W[nk + "1.conv.bias"] + W["no_mem_embed"].view(-1))
    emb = emb + lat3.repeat_interleave(4, dim=1)                                          # nearest-2x = row >> 2 in engine order

    def composed(sw, sb, lw, lb):
        wc = (sw.flatten(1).double() @ lw.flatten(1).double()).float()
        bc = (sb.double() + sw.flatten(1).double() @ lb.double()).float()
        return wc, bc
    w1c, b1c = composed(W[d + "conv_s1.weight"], W[d + "conv_s1.bias"], W[nk + "2.conv.weight"], W[nk + "2.conv.bias"])
    w0c, b0c = composed(W[d + "conv_s0.weight"], W[d + "conv_s0.bias"], W[nk + "3.conv.weight"], W[nk + "3.conv.bias"])
    fs1 = lin(outs[1], w1c, b1c)
    fs0 = lin(outs[0], w0c, b0c)
    return {"image_embed": from_engine_order(emb, 2), "feat_s1": from_engine_order(fs1, 1), "feat_s0": from_engine_order(fs0, 0)}


# ----------------------------------------------------------------------------- mask decoder (folded form of decoder_fused.hip)
def _dense_pe_engine(W) -> torch.Tensor:
    """(4096,256) dense PE rows in engine order, as engine.hip:361-373 builds it"""
    from oracle.sam2_ref import dense_pe
    pe = dense_pe(W, 64)[0].permute(1, 2, 0)[None]            # (1,64,64,256)
    return to_engine_order(pe, 2)[0]


def _fold_rows(a, w, scale):
    """dec_fold mode 0: out[p][8h+t][d] = bf16(scale * sum_j a[p][t][16h+j] W[16h+j][d]);  a (P,8,128) fp32, w (128,256)"""
    P = a.shape[0]
    ah = a.view(P, 8, 8, 16).permute(0, 2, 1, 3)                # (P,h,t,16)
    wh = bf(w).view(8, 16, 256)
    out = torch.einsum("phtj,hjd->phtd", ah, wh) * scale
    return bf(out.reshape(P, 64, 256))


def _t2i(X, pek, q_tok, a, Wd, prefix, split):
    """tokens -> image cross attention (dec_t2i_kernel + dec_t2i_finish_kernel).  X (P,4096,256) bf16-valued engine order,
    q_tok (P,8,256) fp32 = queries + tok_pe.  Returns t_att (P,8,128) bf16-valued."""
    P = X.shape[0]
    kscale = 0.25 * LOG2E
    tq = lin(q_tok, Wd[prefix + ".q_proj.weight"], Wd[prefix + ".q_proj.bias"])            # (P,8,128) fp32
    Qt = _fold_rows(tq, Wd[prefix + ".k_proj.weight"], kscale)                            # (P,64,256)
    pq = bf(tq * kscale).view(P, 8, 8, 16).permute(0, 2, 1, 3)                            # (P,h,t,16)
    pekh = pek.view(4096, 8, 16)
    s = torch.einsum("phtj,nhj->phtn", pq, pekh).reshape(P, 64, 4096) + torch.matmul(Qt, X.transpose(1, 2))   # log2 domain
    # online softmax exactly as the kernel partitions the keys: `split` contiguous parts, each walked in 64-key blocks whose
    # two 32-key halves are kept by different waves and merged at the end
    nkeys = 4096 // split
    Zs, ms, ls = [], [], []
    for sp in range(split):
        part_o, part_m, part_l = [], [], []
        for kh in range(2):
            m = torch.full((P, 64, 1), -3.0e38)
            l = torch.zeros(P, 64, 1)
            o = torch.zeros(P, 64, 256)
            for kb in range(nkeys // 64):
                k0 = sp * nkeys + kb * 64 + kh * 32
                sb = s[:, :, k0:k0 + 32]
                mn = torch.maximum(m, sb.max(-1, keepdim=True).values)
                alpha = torch.exp2(m - mn)
                e = torch.exp2(sb - mn)
                l = l * alpha + e.sum(-1, keepdim=True)
                o = o * alpha + torch.matmul(bf(e), X[:, k0:k0 + 32])
                m = mn
            part_o.append(o); part_m.append(m); part_l.append(l)
        mn = torch.maximum(part_m[0], part_m[1])
        a1, a2 = torch.exp2(part_m[0] - mn), torch.exp2(part_m[1] - mn)
        Zs.append(part_o[0] * a1 + part_o[1] * a2); ms.append(mn); ls.append(part_l[0] * a1 + part_l[1] * a2)
    wv, bv = bf(Wd[prefix + ".v_proj.weight"]), Wd[prefix + ".v_proj.bias"]
    if split == 1:
        Z = bf(Zs[0] * (1.0 / ls[0]))                                                     # re-read as a bf16 MFMA operand
    else:
        mm = torch.stack(ms, 0).max(0).values
        wts = [torch.exp2(m_ - mm) for m_ in ms]
        L = sum(w_ * l_ for w_, l_ in zip(wts, ls))
        Z = sum((w_ / L) * z_ for w_, z_ in zip(wts, Zs))                                 # fp32 (finish kernel)
    Zh = Z.view(P, 8, 8, 256)                                                             # (P,h,t,256)
    out = torch.einsum("phtd,hid->pthi", Zh, wv.view(8, 16, 256)).reshape(P, 8, 128) + bv
    return bf(out)


def _i2t(X, peq, k_tok, v_tok, Wd, prefix, lnw, lnb):
    """image -> tokens cross attention + residual + norm4 (dec_i2t_kernel).  Returns the new X (bf16-valued)."""
    P = X.shape[0]
    kscale = 0.25 * LOG2E
    tk = lin(k_tok, Wd[prefix + ".k_proj.weight"], Wd[prefix + ".k_proj.bias"])            # (P,8,128)
    tv = lin(v_tok, Wd[prefix + ".v_proj.weight"], Wd[prefix + ".v_proj.bias"])
    Kt = _fold_rows(tk, Wd[prefix + ".q_proj.weight"], kscale)                            # (P,64,256)
    cb = (tk.view(P, 8, 8, 16) * Wd[prefix + ".q_proj.bias"].view(1, 1, 8, 16)).sum(-1).permute(0, 2, 1).reshape(P, 64) * kscale
    wo = bf(Wd[prefix + ".out_proj.weight"]).view(256, 8, 16)                             # W[d][16h+j]
    Vt = bf(torch.einsum("pthj,dhj->pdht", tv.view(P, 8, 8, 16), wo)).reshape(P, 256, 64)   # VtT[p][d][8h+t]
    kq = bf(tk * kscale).view(P, 8, 8, 16).permute(0, 2, 1, 3)                            # (P,h,t,16)
    s = torch.einsum("nhj,phtj->pnht", peq.view(4096, 8, 16), kq) + torch.matmul(X, Kt.transpose(1, 2)).view(P, 4096, 8, 8)
    s = s + cb.view(P, 1, 8, 8)
    e = torch.exp2(s - s.max(-1, keepdim=True).values)
    pr = bf(e * (1.0 / e.sum(-1, keepdim=True))).reshape(P, 4096, 64)
    y = torch.matmul(pr, Vt.transpose(1, 2)) + Wd[prefix + ".out_proj.bias"] + X
    mean = y.mean(-1, keepdim=True)
    var = torch.clamp((y * y).mean(-1, keepdim=True) - mean * mean, min=0.0)
    return bf((y - mean) * torch.rsqrt(var + 1e-5) * lnw + lnb)


def mask_embed_emul(W, emb_rows, mask_in, clamp_abs=0.0):
    """mask_embed_src_kernel<true>: X0[p] = bf16( bf16(W3).bf16(h2) + (image_embed + b3) ), h2 from the fp32 mask_downscaling head.
    emb_rows (4096,256) engine order; mask_in (P,256,256) fp32."""
    m = "sam_prompt_encoder.mask_downscaling."
    x = mask_in[:, None]
    if clamp_abs > 0:
        x = torch.clamp(x, -clamp_abs, clamp_abs)

    def ln2d(x, w, b):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        return (x - u) * torch.rsqrt(s + 1e-6) * w[None, :, None, None] + b[None, :, None, None]
    x = F.conv2d(x, W[m + "0.weight"], W[m + "0.bias"], stride=2)
    x = gelu_fit(ln2d(x, W[m + "1.weight"], W[m + "1.bias"]))
    x = F.conv2d(x, W[m + "3.weight"], W[m + "3.bias"], stride=2)
    h2 = gelu_fit(ln2d(x, W[m + "4.weight"], W[m + "4.bias"]))                            # (P,16,64,64)
    h2 = to_engine_order(h2.permute(0, 2, 3, 1), 2)                                       # (P,4096,16)
    dense = F.linear(bf(h2), bf(W[m + "6.weight"].flatten(1)))
    return bf(dense + (emb_rows + W[m + "6.bias"])[None])


@torch.no_grad()
def mask_decoder_emul(W, feats, pts: torch.Tensor, labels: Optional[torch.Tensor], multimask_output: bool,
                      mask_in: Optional[torch.Tensor] = None, mask_clamp: float = 0.0, chunk_prompts: Optional[int] = None):
    """engine.hip decode_chunk restated.  feats: dict of (1,C,H,W) fp32 features (the ENGINE's or encode_image_emul's);
    pts (P,2) model pixels; labels (P,) or None (= 1).  Returns low_res (P,M,256,256), iou (P,M), obj (P,), all_masks (P,4,..)."""
    d = "sam_mask_decoder."
    tp = d + "transformer."
    pe_name = "sam_prompt_encoder."
    P = pts.shape[0]
    Pk = chunk_prompts or P                       # the kernels pick their key split from the launch's prompt count
    split = 1
    while split < 8 and Pk * split < 512:
        split *= 2
    emb = to_engine_order(feats["image_embed"].permute(0, 2, 3, 1), 2)[0]                 # (4096,256)
    pe = _dense_pe_engine(W)
    # prompt tokens (prompt_tokens_kernel)
    out_tok = torch.cat([W[d + "obj_score_token.weight"], W[d + "iou_token.weight"], W[d + "mask_tokens.weight"]], 0)
    G = W[pe_name + "pe_layer.positional_encoding_gaussian_matrix"]
    xy = 2.0 * ((pts + 0.5) / 1024.0) - 1.0
    ang = 6.283185307179586 * (xy[:, :1] * G[0][None] + xy[:, 1:] * G[1][None])
    lab = torch.ones(P, dtype=torch.int64) if labels is None else labels.to(torch.int64)
    pt_e = torch.cat([torch.sin(ang), torch.cos(ang)], -1)
    pemb = torch.stack([W[pe_name + f"point_embeddings.{k}.weight"][0] for k in range(4)], 0)
    nap = W[pe_name + "not_a_point_embed.weight"][0]
    pt_e = torch.where((lab >= 0)[:, None], pt_e + pemb[lab.clamp(min=0) & 3], nap[None].expand(P, -1))
    tok_pe = torch.cat([out_tok[None].expand(P, -1, -1), pt_e[:, None], nap[None, None].expand(P, 1, -1)], 1)   # (P,8,256)
    queries = tok_pe.clone()
    if mask_in is None:
        X = bf(emb + W[pe_name + "no_mask_embed.weight"][0])[None].expand(P, -1, -1)
    else:
        X = mask_embed_emul(W, emb, mask_in, mask_clamp)

    def proj_pe(wname):                      # engine.hip:536-549: bf16 GEMM with bf16 output
        return bf(F.linear(bf(pe), bf(W[wname])))

    def self_attn(prefix, qk_in, v_in):
        tq = lin(qk_in, W[prefix + ".q_proj.weight"], W[prefix + ".q_proj.bias"]).view(P, 8, 8, 32).transpose(1, 2)
        tk = lin(qk_in, W[prefix + ".k_proj.weight"], W[prefix + ".k_proj.bias"]).view(P, 8, 8, 32).transpose(1, 2)
        tv = lin(v_in, W[prefix + ".v_proj.weight"], W[prefix + ".v_proj.bias"]).view(P, 8, 8, 32).transpose(1, 2)
        a = torch.softmax(torch.matmul(tq, tk.transpose(-1, -2)) * (32 ** -0.5), -1) @ tv       # fp32 (dec_attn_fewkeys)
        return lin(bf(a.transpose(1, 2).reshape(P, 8, 256)), W[prefix + ".out_proj.weight"], W[prefix + ".out_proj.bias"])

    def ln5(x, prefix):
        return ln(x, W[prefix + ".weight"], W[prefix + ".bias"], 1e-5)

    def t2i_block(prefix, ln_prefix, X, queries):
        att = _t2i(X, proj_pe(prefix + ".k_proj.weight"), queries + tok_pe, None, W, prefix, split)
        queries = queries + lin(att, W[prefix + ".out_proj.weight"], W[prefix + ".out_proj.bias"])
        return ln5(queries, ln_prefix)

    for l in range(2):
        L = f"{tp}layers.{l}."
        if l == 0:
            queries = self_attn(L + "self_attn", queries, queries)
        else:
            queries = queries + self_attn(L + "self_attn", queries + tok_pe, queries)
        queries = ln5(queries, L + "norm1")
        queries = t2i_block(L + "cross_attn_token_to_image", L + "norm2", X, queries)
        h = bf(torch.relu(lin(queries, W[L + "mlp.layers.0.weight"], W[L + "mlp.layers.0.bias"])))
        queries = ln5(queries + lin(h, W[L + "mlp.layers.1.weight"], W[L + "mlp.layers.1.bias"]), L + "norm3")
        X = _i2t(X, proj_pe(L + "cross_attn_image_to_token.q_proj.weight"), queries + tok_pe, queries, W,
                 L + "cross_attn_image_to_token", W[L + "norm4.weight"], W[L + "norm4.bias"])
    queries = t2i_block(tp + "final_attn_token_to_image", tp + "norm_final_attn", X, queries)

    def mlp3(prefix, x, sigmoid=False):
        h = bf(torch.relu(lin(x, W[prefix + ".layers.0.weight"], W[prefix + ".layers.0.bias"])))
        h = bf(torch.relu(lin(h, W[prefix + ".layers.1.weight"], W[prefix + ".layers.1.bias"])))
        y = lin(h, W[prefix + ".layers.2.weight"], W[prefix + ".layers.2.bias"])
        return torch.sigmoid(y) if sigmoid else y
    iou4 = mlp3(d + "iou_prediction_head", queries[:, 1], True)
    obj = mlp3(d + "pred_obj_score_head", queries[:, 0])[:, 0]
    hyper = torch.stack([mlp3(f"{d}output_hypernetworks_mlps.{k}", queries[:, 2 + k]) for k in range(4)], 1)   # (P,4,32)
    # upscaling head (dec_upscale_kernel): ConvT1 on bf16 X with bf16 weights, + bias + feat_s1, LN64, GELU -> bf16, ConvT2, + feat_s0, GELU
    src = from_engine_order(X, 2)                                                         # (P,256,64,64), bf16-valued
    up = F.conv_transpose2d(src, bf(W[d + "output_upscaling.0.weight"]), W[d + "output_upscaling.0.bias"], stride=2) + feats["feat_s1"]
    u = up.mean(1, keepdim=True)
    sgm = (up - u).pow(2).mean(1, keepdim=True)
    up = (up - u) * torch.rsqrt(sgm + 1e-6) * W[d + "output_upscaling.1.weight"][None, :, None, None] + W[d + "output_upscaling.1.bias"][None, :, None, None]
    up = bf(gelu_fit(up))
    up = gelu_fit(F.conv_transpose2d(up, bf(W[d + "output_upscaling.3.weight"]), W[d + "output_upscaling.3.bias"], stride=2) + feats["feat_s0"])
    masks4 = torch.matmul(hyper, up.flatten(2)).view(P, 4, 256, 256)
    if multimask_output:
        return masks4[:, 1:], iou4[:, 1:], obj, masks4, iou4
    best = torch.argmax(iou4[:, 1:], -1)
    ar = torch.arange(P)
    f = masks4[:, 0].flatten(1)
    ai, au = (f > DYN_MULTIMASK_DELTA).sum(-1).float(), (f > -DYN_MULTIMASK_DELTA).sum(-1).float()
    stable = torch.where(au > 0, ai / au, torch.ones_like(au)) >= DYN_MULTIMASK_THRESH
    masks = torch.where(stable[:, None, None], masks4[:, 0], masks4[:, 1:][ar, best])[:, None]
    iou = torch.where(stable, iou4[:, 0], iou4[:, 1:][ar, best])[:, None]
    return masks, iou, obj, masks4, iou4


# ----------------------------------------------------------------------------- image predictor on the emulated arithmetic
class ImagePredictorEmul:
    """oracle.sam2_ref.ImagePredictorRef with the engine's precision: same interface, so oracle.amg_ref drives it unchanged."""

    def __init__(self, W: Dict[str, torch.Tensor], cfg: HieraConfig, launch_prompts: int = 8):
        self.W, self.cfg, self.res = W, cfg, cfg.image_size
        self.launch_prompts = launch_prompts      # the engine picks its key split from the prompts per launch (<= 64 -> split 8)
        self.feats, self.orig_hw = None, None

    @torch.no_grad()
    def set_image(self, image: np.ndarray):
        from oracle.sam2_ref import sam2_transforms
        self.orig_hw = image.shape[:2]
        self.feats = encode_image_emul(self.W, self.cfg, sam2_transforms(image, self.res))

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
    def _predict(self, pts, labels, mask_input=None, multimask_output=True):
        mi = None if mask_input is None else mask_input[:, 0]
        low, iou, _, _, _ = mask_decoder_emul(self.W, self.feats, pts[:, 0], labels[:, 0], multimask_output, mask_in=mi,
                                              chunk_prompts=self.launch_prompts)
        masks = F.interpolate(low, self.orig_hw, mode="bilinear", align_corners=False)
        return masks, iou, torch.clamp(low, -32.0, 32.0)
