"""The e4m3 rounding of oracle/fp8_ref.py against torch's own float8_e4m3fn conversion (CPU)."""
import numpy as np
import torch

from oracle import fp8_ref


def test_e4m3_round_matches_torch_float8():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 1, 20000), rng.normal(0, 100, 20000), rng.uniform(-2 ** -5, 2 ** -5, 20000),
                        np.array([0.0, 448.0, -448.0, 460.0, 2 ** -9, 2 ** -10, 1.5 * 2 ** -9, 0.4375, 17.0, 18.0, 19.0, 1e-8])]).astype(np.float32)
    x = np.clip(x, -448, 448)                 # (torch's conversion does not saturate beyond the format's range)
    want = torch.from_numpy(x).to(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(fp8_ref.e4m3_round(x), want)


def test_row_scales_are_powers_of_two_and_use_the_range():
    rng = np.random.default_rng(1)
    w = (rng.normal(0, 1, (64, 96)) * rng.uniform(1e-3, 10, (64, 1))).astype(np.float32)
    q = fp8_ref.quantise_rows(w)
    rel = np.abs(q - w).max(axis=1) / np.abs(w).max(axis=1)
    assert rel.max() < 2 ** -4                                     # half a quantum of the top binade, relative to the row maximum
    # the dequantised values are exact in bf16 (4 significant bits times a power of two)
    t = torch.from_numpy(q)
    assert torch.equal(t.to(torch.bfloat16).float(), t)


def test_mx_quantise_rule_and_layout():
    """OCP MX restatement (oracle/fp8_ref.py: mx_quantise, mx_scale_panel): per 32-element block the scale is the smallest power of two with
    amax <= 448 * scale, elements are e4m3 round-to-nearest-even of value / scale (torch.float8_e4m3fn as the independent rounding), an
    all-zero block gets the smallest scale, and the K-step-major panel holds byte (row, block) at [block // 4][row][block % 4]."""
    import torch
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 160, generator=g) * torch.exp2(torch.randint(-12, 12, (37, 5), generator=g).float()).repeat_interleave(32, 1)
    x[3, 32:64] = 0.0
    x[5, 0] = 448.0 * 4; x[5, 1:32] = 0.0           # amax exactly at 448 * 2^2: that scale, no larger
    x[6, 0] = 449.0                                 # just above 448: the next power of two
    deq, q8, sb = fp8_ref.mx_quantise(x)
    assert q8.dtype == torch.uint8 and q8.shape == x.shape and sb.shape == (37, 5)
    sc = torch.exp2(sb.float() - 127)
    amax = x.abs().reshape(37, 5, 32).amax(-1)
    nz = amax > 0
    assert (amax[nz] <= 448 * sc[nz]).all() and (amax[nz] > 224 * sc[nz]).all()          # smallest such power of two
    assert int(sb[3, 1]) == 0 and int(sb[5, 0]) == 127 + 2 and int(sb[6, 0]) == 127 + 1
    ref = (x.reshape(37, 5, 32) / sc[..., None]).to(torch.float8_e4m3fn)
    assert torch.equal(ref.view(torch.uint8).reshape(x.shape), q8)
    assert torch.equal(deq, (ref.float() * sc[..., None]).reshape(x.shape))
    assert (deq - x).abs().max() <= (amax.max() / 14).item()                                # half a step at the top of the range (32 / 448)
    # 1.75 * 2^k boundary of the rule (448 = 1.75 * 2^8): mantissa exactly 1.75 stays, one ulp above moves up
    b = torch.tensor([[1.75 * 2 ** -3] + [0.0] * 31, [float(np.nextafter(np.float32(1.75 * 2 ** -3), np.float32(1)))] + [0.0] * 31])
    _, _, sb2 = fp8_ref.mx_quantise(b)
    assert int(sb2[0, 0]) == 127 - 3 - 8 and int(sb2[1, 0]) == 127 - 3 - 8 + 1
    panel = fp8_ref.mx_scale_panel(sb, 64)
    assert panel.shape == (2, 64, 4) and int(panel[1, 10, 0]) == int(sb[10, 4]) and int(panel[0, 10, 3]) == int(sb[10, 3]) and int(panel[1, 10, 1]) == 127 and int(panel[0, 40, 0]) == 127


def test_mx_weight_selection():
    """mx_quantise_encoder_weights touches exactly what SABER_WEIGHTS_MXFP8 puts on the fp8 MFMA (engine.hip: finalize): fc1 / fc2 of every
    stage-2 / stage-3 block, qkv of those that keep their width; everything else is returned as it came."""
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    cfg = get_config("tiny")
    W = seeded_weights(cfg, 0)
    Wq = fp8_ref.mx_quantise_encoder_weights(W, cfg)
    changed = sorted(k for k in W if not np.array_equal(W[k], Wq[k]))
    want = []
    for i, (din, dout, heads, win, qs) in enumerate(cfg.block_specs()):
        if dout >= 4 * cfg.embed_dim:
            b = f"image_encoder.trunk.blocks.{i}."
            want += [b + "mlp.layers.0.weight", b + "mlp.layers.1.weight"] + ([b + "attn.qkv.weight"] if din == dout else [])
    assert changed == sorted(want) and len(want) > 0
    k = want[0]
    assert np.abs(Wq[k] - W[k]).max() <= np.abs(W[k]).max() / 14
