"""GPU parity of saber_separate_masks (3-D connected components on the device) - integer work, bit-exact:
against the fixtures captured from the reference's own separate_masks (tests/golden/saber_glue.npz, oracle/make_golden.py)
and against the oracle restatement (oracle/saber_ref.py, pinned by the same fixtures on the CPU) on larger volumes."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "saber_glue.npz"), allow_pickle=False)


def run(engine, vol, min_mask_area):
    d = torch.from_numpy(np.ascontiguousarray(vol).astype(np.uint16).view(np.int16)).cuda()
    out, n = engine.separate_masks(d, min_mask_area=min_mask_area)
    lab = out.cpu().numpy().view(np.uint32)
    assert n == int(lab.max())
    return lab


def test_reference_fixtures(engine):
    assert np.array_equal(run(engine, G["sep_in"], 100), G["sep_out_default"])
    assert np.array_equal(run(engine, np.zeros((3, 8, 8), np.uint16), 100), G["sep_out_empty"])
    from saber_amd.segmenters import utils
    assert np.array_equal(run(engine, G["sep_in"], 5), utils.separate_masks(G["sep_in"], min_mask_area=5))


def _blobs(shape, n, seed, rmin=3, rmax=12):
    rng = np.random.default_rng(seed)
    Z, H, W = shape
    vol = np.zeros(shape, np.uint16)
    zz, yy, xx = np.mgrid[:Z, :H, :W]
    for k in range(n):
        cz, cy, cx, r = rng.integers(0, Z), rng.integers(0, H), rng.integers(0, W), rng.integers(rmin, rmax)
        vol[(zz - cz) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = (k % 60000) + 1
    return vol


@pytest.mark.parametrize("shape,area", [((24, 200, 333), 0), ((24, 200, 333), 10), ((9, 65, 1030), 1), ((1, 300, 300), 0), ((40, 64, 64), 3)])
def test_blobs_match_oracle(engine, shape, area):
    from oracle import saber_ref
    vol = _blobs(shape, 60, seed=shape[1] + area)
    assert np.array_equal(run(engine, vol, area), saber_ref.separate_masks(vol, min_mask_area=area))


@pytest.mark.parametrize("density", [0.05, 0.2, 0.5, 0.9])
def test_speckle_26_connectivity(engine, density):
    """random voxels: thousands of small components, diagonal-only contacts, label order = order of first voxels"""
    from oracle import saber_ref
    rng = np.random.default_rng(int(density * 100))
    vol = (rng.uniform(size=(12, 90, 131)) < density).astype(np.uint16) * 7
    for area in (0, 1):
        assert np.array_equal(run(engine, vol, area), saber_ref.separate_masks(vol, min_mask_area=area))


def test_edge_shapes(engine):
    from oracle import saber_ref
    full = np.ones((5, 70, 129), np.uint16)
    assert np.array_equal(run(engine, full, 0), saber_ref.separate_masks(full, 0))
    one = np.zeros((4, 8, 200), np.uint16); one[2, 3, 199] = 9
    assert np.array_equal(run(engine, one, 0), saber_ref.separate_masks(one, 0))
    assert run(engine, one, 1).max() == 0                      # 1 voxel < 10
    # two long rows that only touch diagonally across a plane boundary, and a run that spans several 64-voxel chunks
    v = np.zeros((3, 6, 300), np.uint16); v[0, 1, :150] = 1; v[1, 2, 150:] = 2; v[2, 5, 10:290] = 3
    assert np.array_equal(run(engine, v, 0), saber_ref.separate_masks(v, 0))
    with pytest.raises(ValueError):
        engine.separate_masks(torch.zeros((1, 4, 4), dtype=torch.int16).cuda()[:, :, :0].contiguous())


def test_full_size_volume_and_timing(engine):
    """config-3-sized stitch: 64 x 1024 x 1024 label planes (the reference runs scipy on the host for this)"""
    from oracle import saber_ref
    rng = np.random.default_rng(3)
    Z = 64
    vol = np.zeros((Z, 1024, 1024), np.uint16)
    zz = np.arange(Z)[:, None, None]
    yy, xx = np.mgrid[:1024, :1024]
    for k in range(120):
        cz, cy, cx, r = rng.integers(0, Z), rng.integers(0, 1024), rng.integers(0, 1024), rng.integers(10, 60)
        m = (yy - cy) ** 2 + (xx - cx) ** 2
        for z in range(max(0, cz - r), min(Z, cz + r + 1)):
            vol[z][m < r * r - (z - cz) ** 2] = k + 1
    d = torch.from_numpy(vol.view(np.int16)).cuda()
    engine.separate_masks(d, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, n = engine.separate_masks(d, 100)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = saber_ref.separate_masks(vol, 100)
    t_cpu = time.perf_counter() - t0
    print(f"separate_masks 64x1024x1024: device {t_gpu * 1e3:.1f} ms, host scipy {t_cpu * 1e3:.0f} ms, {n} labels")
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref)
