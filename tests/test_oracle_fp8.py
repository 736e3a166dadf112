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
