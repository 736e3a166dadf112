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


def quantise_encoder_weights(W: dict, cfg) -> dict:
    """The tensors SABER_WEIGHTS_FP8_E4M3 covers: qkv / proj / fc1 / fc2 of the Hiera blocks of stages 2 and 3."""
    out = dict(W)
    first = cfg.stages[0] + cfg.stages[1]
    for i in range(first, sum(cfg.stages)):
        b = f"image_encoder.trunk.blocks.{i}."
        for n in ("attn.qkv", "attn.proj", "mlp.layers.0", "mlp.layers.1"):
            out[b + n + ".weight"] = quantise_rows(W[b + n + ".weight"])
    return out
