"""CPU: weight specs / checkpoint loader and the AMG oracle's utilities."""
import os

import numpy as np
import pytest
import torch

from saber_amd.model_config import get_config
from saber_amd import weights


def test_param_counts_match_architecture():
    assert weights.count_params(get_config("large"), "image_encoder.") == 212_679_392 or abs(weights.count_params(get_config("large"), "image_encoder.") - 212.7e6) < 0.1e6
    assert abs(weights.count_params(get_config("tiny"), "image_encoder.") - 27.2e6) < 0.3e6
    specs = get_config("large").block_specs()
    assert len(specs) == 48 and specs[2] == (144, 288, 4, 8, 2) and specs[23][3] == 0 and specs[44] == (576, 1152, 16, 16, 2)


def test_seeded_weights_are_deterministic_and_order_independent():
    cfg = get_config("tiny")
    a = weights.seeded_weights(cfg, 3)
    b = weights.seeded_weights(cfg, 3)
    k = "image_encoder.trunk.blocks.4.attn.qkv.weight"
    assert np.array_equal(a[k], b[k]) and not np.array_equal(a[k], weights.seeded_weights(cfg, 4)[k])
    assert np.array_equal(weights._gen(k, a[k].shape, "w", 1.0, 3), a[k])


def test_checkpoint_round_trip(tmp_path):
    cfg = get_config("tiny")
    W = weights.seeded_weights(cfg, 1)
    path = os.path.join(tmp_path, "sam2.1_hiera_tiny.pt")
    torch.save({"model": {k: torch.from_numpy(v) for k, v in W.items()} | {"memory_attention.x": torch.zeros(3)}}, path)
    L = weights.load_checkpoint(path, cfg)
    assert list(L) == list(W) and all(np.array_equal(L[k], W[k]) for k in W)
    bad = dict(W)
    bad.pop("no_mem_embed")
    torch.save({"model": {k: torch.from_numpy(v) for k, v in bad.items()}}, path)
    with pytest.raises(ValueError, match="no_mem_embed"):
        weights.load_checkpoint(path, cfg)
    with pytest.raises(ValueError):
        weights.load_checkpoint(path, get_config("large"))


def test_pretrained_weight_naming(monkeypatch, tmp_path):
    from saber_amd import pretrained_weights as pw
    cfg, ck = pw.get_sam2_checkpoint("large")
    assert cfg == "configs/sam2.1/sam2.1_hiera_l.yaml" and ck.endswith("sam2.1_hiera_large.pt")
    assert pw.get_sam2_checkpoint("base")[0].endswith("sam2.1_hiera_b+.yaml")
    with pytest.raises(ValueError):
        pw.get_sam2_checkpoint("huge")
    monkeypatch.setenv("SABER_AMD_CHECKPOINTS", str(tmp_path))
    monkeypatch.delenv("SABER_AMD_SEEDED_WEIGHTS", raising=False)
    with pytest.raises(FileNotFoundError):
        pw.resolve_weights("large")
    monkeypatch.setenv("SABER_AMD_SEEDED_WEIGHTS", "1")
    assert pw.resolve_weights("large") == {"seed": 0}


def test_amg_utilities():
    from oracle import amg_ref
    g = amg_ref.build_all_layer_point_grids(32, 2, 2)
    assert [len(x) for x in g] == [1024, 256, 64]
    assert np.allclose(g[0][0], [1 / 64, 1 / 64]) and np.allclose(g[0][33], [3 / 64, 3 / 64])
    boxes, layers = amg_ref.generate_crop_boxes((1024, 1024), 2, 512 / 1500)
    assert len(boxes) == 21 and layers.count(1) == 4 and layers.count(2) == 16
    assert boxes[0] == [0, 0, 1024, 1024] and boxes[1] == [0, 0, 687, 687] and boxes[2] == [0, 338, 687, 1024]
    b = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10]], dtype=np.float32)
    s = np.array([0.5, 0.9, 0.3, 0.5], dtype=np.float32)
    assert amg_ref.nms(b, s, 0.5).tolist() == [1, 2]
    assert amg_ref.nms(b, s, 0.7).tolist() == [1, 0, 2]   # IoU(0,1)=0.68 survives; 3 duplicates 0 (stable order keeps 0)
    m = torch.zeros(2, 8, 8, dtype=torch.bool)
    m[0, 2:5, 3:7] = True
    assert amg_ref.batched_mask_to_box(m).tolist() == [[3, 2, 6, 4], [0, 0, 0, 0]]
    near = amg_ref.is_box_near_crop_edge(torch.tensor([[5.0, 100, 300, 400], [100.0, 100, 300, 400]]), [0, 338, 687, 1024], [0, 0, 1024, 1024])
    assert near.tolist() == [False, False]
    near = amg_ref.is_box_near_crop_edge(torch.tensor([[100.0, 2, 300, 400]]), [0, 338, 687, 1024], [0, 0, 1024, 1024])
    assert near.tolist() == [True]


def test_gelu_fit_matches_exact_erf_gelu():
    """The kernels' GELU (saber_amd/csrc/common.h gelu_erf, fitted by tools/fit_gelu.py) restated in float32 numpy against
    the exact erf GELU of the reference (torch.nn.GELU() in sam2 hieradet MLP / mask decoder upscaling)."""
    import numpy as np
    from scipy.special import erfc
    x = np.linspace(-30, 30, 600001).astype(np.float32)
    x2 = np.minimum(x * x, np.float32(50.0))
    q = x2 * np.float32(1.01426305e-3) + np.float32(-1.06775724e-1)
    q = q * x2 + np.float32(-2.30112134)
    with np.errstate(over="ignore"):
        got = x / (np.float32(1.0) + np.exp2(x * q))
    ref = x.astype(np.float64) * 0.5 * erfc(-x.astype(np.float64) / np.sqrt(2.0))
    assert np.max(np.abs(got - ref)) < 4e-5
