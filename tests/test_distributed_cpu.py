"""CPU, world_size 2 over gloo: the z-sharded volume driver gathers every rank's label planes and all ranks
stitch the identical volume (N>1 path of bench.py / slice_by_slice_device)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _planes(Z, H, W):
    rng = np.random.default_rng(0)
    vol = np.zeros((Z, H, W), dtype=np.uint16)
    zz, yy, xx = np.mgrid[:Z, :H, :W]
    for k in range(6):
        cz, cy, cx, r = rng.integers(0, Z), rng.integers(8, H - 8), rng.integers(8, W - 8), rng.integers(4, 9)
        vol[(zz - cz) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = k + 1
    return vol


def _worker(rank, world, port, Z, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from saber_amd.segmenters.slice_driver import segment_volume_sharded, shard_bounds
    vol = _planes(Z, 40, 48)
    seen = []

    def slice_fn(z):
        seen.append(z)
        return torch.from_numpy(vol[z].astype(np.int16))

    out = segment_volume_sharded(vol, slice_fn, stitch=True, min_mask_area=1)
    z0, z1 = shard_bounds(Z, world, rank)
    assert seen == list(range(z0, z1))
    np.save(os.path.join(out_dir, f"out{rank}.npy"), out)
    dist.destroy_process_group()


@pytest.mark.parametrize("Z", [7, 8])
def test_sharded_volume_driver_gloo(tmp_path, Z):
    port = 29500 + (os.getpid() % 2000) + Z
    mp.spawn(_worker, args=(2, port, Z, str(tmp_path)), nprocs=2, join=True)
    from saber_amd.segmenters import utils
    ref = utils.separate_masks(_planes(Z, 40, 48), min_mask_area=1)
    a, b = np.load(tmp_path / "out0.npy"), np.load(tmp_path / "out1.npy")
    assert np.array_equal(a, ref) and np.array_equal(b, ref)


def test_shard_bounds_cover_everything():
    from saber_amd.segmenters.slice_driver import shard_bounds
    for Z in (1, 7, 64, 512):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(Z, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == Z
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_video_window_shares_partition_the_window():
    """adapters/sam2/video.py: window_shares - contiguous, in order, ceil(k / world) each until the frames run out (rank r's share starts at
    r * per, which is what lets the all-gathered rows be imported as slots 0..k-1 without reordering)"""
    pytest.importorskip("torch")
    try:
        from saber_amd.adapters.sam2.video import window_shares
    except Exception as ex:            # the module binds the HIP library at import; without it this host check cannot run
        pytest.skip(str(ex)[:80])
    for k in range(0, 22):
        for world in (1, 2, 3, 4, 8):
            sh = window_shares(k, world)
            per = -(-k // world) if k else 0
            assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == k
            assert all(a <= b and b - a <= per for a, b in sh) and all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
            assert all(a == r * per for r, (a, b) in enumerate(sh) if b > a)
