"""TEST INFRASTRUCTURE (CPU oracle) - the fp8 weight format of include/saber_amd.h (saber_engine_set_weight_format), restated in numpy:
OCP e4m3fn values (1-4-3, bias 7, max 448, no infinities; published format: OCP 8-bit Floating Point Specification v1.0) with one
power-of-two scale per output row.  No reference anchor: the reference has no fp8 path (BASELINE.json configs[4] is this build's own
extension); the rounding is pinned against torch.float8_e4m3fn in tests/test_oracle_fp8.py."""
import numpy as np


def e4m3_round(x: np.ndarray) -> np.ndarray:
    """round-to-nearest-even onto the e4m3fn grid, saturating at +-448 (fp32 in, fp32 out)"""
    x = np.asarray(x, dtype=np.float32)
    a = np.abs(x).astype(np.float64)
    with np.errstate(divide="ignore"):
        e2 = np.where(a > 0, np.floor(np.log2(np.where(a > 0, a, 1.0))), -6.0)
    e2 = np.maximum(e2, -6.0)
    quantum = np.exp2(e2 - 3.0)
    q = np.minimum(np.rint(a / quantum) * quantum, 448.0)
    return (np.sign(x) * q).astype(np.float32)


def quantise_rows(w: np.ndarray) -> np.ndarray:
    """per-output-row power-of-two scale, e4m3 values, returned de-quantised (fp32)"""
    w = np.asarray(w, dtype=np.float32)
    w2 = w.reshape(w.shape[0], -1)
    mx = np.abs(w2).max(axis=1, keepdims=True)
    with np.errstate(divide="ignore"):
        scale = np.where(mx > 0, np.exp2(np.ceil(np.log2(np.where(mx > 0, mx, 1.0) / 448.0))), 1.0).astype(np.float32)
    return (e4m3_round(w2 / scale) * scale).reshape(w.shape)


# ----------------------------------------------------------------------------- OCP MX (e4m3 elements, e8m0 scale per 32 K-elements)
# What the fp8-MFMA weight format runs on (saber_amd/csrc/gemm_fp8.hip; include/saber_amd.h SABER_WEIGHTS_MXFP8).  New component: no
# reference anchor.  Rule shared by every quantiser of the engine: per block of 32 consecutive K-elements the
# scale is the smallest power of two 2^e with amax <= 448 * 2^e (e >= -127; an all-zero block gets e = -127), elements are
# round-to-nearest-even e4m3fn of value / 2^e (never saturating, by construction).
def mx_scale_exp(amax):
    """torch fp32 tensor of block maxima -> int32 exponents e"""
    import torch
    m, ex = torch.frexp(amax.float())                       # amax = m 2^ex, m in [0.5, 1)
    e = ex.to(torch.int32) - 1 - 8 + (m * 2.0 > 1.75).to(torch.int32)
    return torch.where(amax > 0, e, torch.full_like(e, -127)).clamp_(min=-127)


def mx_quantise(x):
    """x: torch fp32 [..., K], K % 32 == 0 -> (de-quantised fp32 [..., K], element bytes uint8 [..., K], scale bytes uint8 [..., K / 32])"""
    import torch
    x = x.float()
    xb = x.reshape(x.shape[:-1] + (x.shape[-1] // 32, 32))
    e = mx_scale_exp(xb.abs().amax(-1))
    inv = torch.exp2(-e.float())[..., None]
    q8 = (xb * inv).to(torch.float8_e4m3fn)
    deq = (q8.float() * torch.exp2(e.float())[..., None]).reshape(x.shape)
    return deq, q8.view(torch.uint8).reshape(x.shape), (e + 127).to(torch.uint8)


def mx_scale_panel(scale_bytes, rows_padded: int):
    """scale bytes [rows][K / 32] -> the engine's K-step-major panel [Kp / 128][rows_padded][4] (Kp = K rounded up to 128; padding: unit scale 127)"""
    import torch
    rows, nb = scale_bytes.shape
    nbp = (nb + 3) // 4 * 4
    out = torch.full((nbp // 4, rows_padded, 4), 127, dtype=torch.uint8)
    sp = torch.full((rows, nbp), 127, dtype=torch.uint8)
    sp[:, :nb] = scale_bytes
    out[:, :rows] = sp.reshape(rows, nbp // 4, 4).permute(1, 0, 2)
    return out.contiguous()


def quantise_encoder_weights(W: dict, cfg) -> dict:
    """The tensors SABER_WEIGHTS_FP8_E4M3 covers: qkv / proj / fc1 / fc2 of the Hiera blocks of stages 2 and 3."""
    out = dict(W)
    first = cfg.stages[0] + cfg.stages[1]
    for i in range(first, sum(cfg.stages)):
        b = f"image_encoder.trunk.blocks.{i}."
        for n in ("attn.qkv", "attn.proj", "mlp.layers.0", "mlp.layers.1"):
            out[b + n + ".weight"] = quantise_rows(W[b + n + ".weight"])
    return out


def mx_quantise_encoder_weights(W: dict, cfg) -> dict:
    """numpy weight dict with the MXFP8 weight format applied (engine.hip: Finalizer::quantise_mx on the tensors SABER_WEIGHTS_MXFP8
    selects): attn.qkv of the stage-2 / stage-3 blocks that keep their width, mlp.layers.0 / mlp.layers.1 of every stage-2 / stage-3 block"""
    import torch
    out = dict(W)
    for i, (din, dout, heads, win, qs) in enumerate(cfg.block_specs()):
        if dout < 4 * cfg.embed_dim:
            continue
        b = f"image_encoder.trunk.blocks.{i}."
        names = ["mlp.layers.0.weight", "mlp.layers.1.weight"] + (["attn.qkv.weight"] if din == dout else [])
        for nm in names:
            out[b + nm] = mx_quantise(torch.from_numpy(np.asarray(W[b + nm], dtype=np.float32)))[0].numpy()
    return out
