"""The AMG utilities of oracle/amg_ref.py (restated from the published upstream sam2/utils/amg.py; the package itself is absent here) against
the independent restatement that ships with `transformers` (models/sam2/image_processing_sam2.py): point grids, crop boxes of every layer,
stability score, mask -> box, near-crop-edge test.  CPU only."""
import numpy as np
import pytest
import torch

import ast
import importlib.util
import math
import os
import types
from itertools import product

from oracle import amg_ref


def _hf_functions():
    """The module imports torchvision (absent from this image) for its NMS; the five utilities compared here need torch only, so their
    definitions are compiled straight from the installed file."""
    spec = importlib.util.find_spec("transformers")
    if spec is None:
        pytest.skip("transformers is not installed")
    path = os.path.join(list(spec.submodule_search_locations)[0], "models", "sam2", "image_processing_sam2.py")
    if not os.path.exists(path):
        pytest.skip("this transformers build has no sam2 image processor")
    want = {"_compute_stability_score", "_batched_mask_to_box", "_is_box_near_crop_edge", "_generate_per_layer_crops", "_build_point_grid"}
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert {n.name for n in body} == want
    ns = {"torch": torch, "math": math, "product": product, "Any": object}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return types.SimpleNamespace(**{k: ns[k] for k in want})


hf = _hf_functions()


def test_point_grids_and_crop_boxes():
    for n in (1, 4, 8, 16, 32):
        assert np.abs(amg_ref.build_point_grid(n) - hf._build_point_grid(n).numpy()).max() < 1e-6
    for (h, w) in ((1024, 1024), (512, 384), (700, 1300), (333, 257)):
        for layers in (0, 1, 2, 3):
            for ratio in (512 / 1500, 0.2):
                a, la = amg_ref.generate_crop_boxes((h, w), layers, ratio)
                b, lb = hf._generate_per_layer_crops(layers, ratio, (h, w))
                assert a == b and la == lb
                assert len(a) == sum(4 ** i for i in range(layers + 1))


def test_stability_box_and_edge_filters():
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(7, 3, 64, 80, generator=g) * 2
    for thr, off in ((0.0, 1.0), (0.0, 0.7), (0.5, 0.3)):
        a = amg_ref.calculate_stability_score(logits, thr, off)
        b = hf._compute_stability_score(logits, thr, off)
        assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))
    masks = logits > 1.0
    masks[0, 0] = False                                      # an empty mask: [0, 0, 0, 0]
    masks[1, 1] = True                                       # a full one
    a = amg_ref.batched_mask_to_box(masks.flatten(0, 1))
    b = hf._batched_mask_to_box(masks.flatten(0, 1))
    assert torch.equal(a, b) and a[0].tolist() == [0, 0, 0, 0] and a[4].tolist() == [0, 0, 79, 63]
    boxes = torch.tensor([[0, 0, 50, 50], [5, 5, 300, 200], [19, 30, 100, 100], [21, 30, 100, 100], [100, 100, 492, 480], [100, 100, 470, 400]], dtype=torch.float)
    for crop, orig in (([0, 0, 512, 512], [0, 0, 512, 512]), ([256, 128, 768, 640], [0, 0, 1024, 1024]), ([512, 0, 1024, 512], [0, 0, 1024, 1024])):
        a = amg_ref.is_box_near_crop_edge(boxes, crop, orig)
        b = hf._is_box_near_crop_edge(boxes, crop, orig)
        assert torch.equal(a, b)
