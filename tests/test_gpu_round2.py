"""GPU: direct tests of shipped kernels / entry points that round 1 only reached indirectly:
saber_mask_pair_intersections against flat @ flat.T, the W > 1024 form of K8, K0 on (H,W,3) input against the reference's own
output, the RCCL branch of the sharded volume driver at world size 1, the caller's current device after a C-ABI call."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def ptr(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr())


def test_pair_intersections_against_matmul(engine):
    """reference: remove_duplicate_masks builds IoU from flat @ flat.T counts (saber/segmenters/utils.py:21-29)"""
    rng = np.random.default_rng(5)
    for (n, H, W) in ((1, 64, 64), (7, 96, 160), (33, 130, 75), (40, 1024, 1024), (9, 96, 160), (251, 256, 256)):     # both kernels: one block per pair; 8 x 8 pairs per block
        masks = rng.uniform(size=(n, H, W)) > rng.uniform(0.3, 0.9, size=(n, 1, 1))
        masks[0] = False                                   # an empty mask
        W32 = (W + 31) // 32
        pad = np.zeros((n, H, W32 * 32), bool)
        pad[..., :W] = masks
        packed = np.packbits(pad, axis=-1, bitorder="little").view(np.int32)
        got = engine.pair_intersections(torch.from_numpy(np.ascontiguousarray(packed)).cuda(), H, W).cpu().numpy()
        flat = masks.reshape(n, -1).astype(np.float64)
        ref = (flat @ flat.T).astype(np.int64)
        assert np.array_equal(got.astype(np.int64), ref), (n, H, W)


@pytest.mark.parametrize("size,crop", [((1536, 2048), (0, 0, 2048, 1536)), ((1536, 2048), (600, 300, 1200, 900)), ((2048, 2048), (1000, 900, 1048, 1148))])
def test_mask_post_wide_images(gpu_lib, size, crop):
    """K8 for W > 1024 (mask_post_kernel<false>): 2048^2 micrographs are ordinary inputs (the reference warns above 1280 px only)"""
    from saber_amd.engine import unpack_bits
    H, W = size
    x0, y0, cw, ch = crop
    g = torch.Generator().manual_seed(cw + H)
    n = 4
    low = F.interpolate(torch.randn(n, 1, 16, 16, generator=g) * 4, size=(256, 256), mode="bicubic")[:, 0].contiguous()
    low[3] = -5.0
    full = F.interpolate(low[:, None], size=(ch, cw), mode="bilinear", align_corners=False)[:, 0]
    thr, off = 0.0, 0.7
    ref_mask = torch.zeros(n, H, W, dtype=torch.bool)
    ref_mask[:, y0:y0 + ch, x0:x0 + cw] = full > thr
    bits = torch.zeros(n, H, W // 32, dtype=torch.int32, device="cuda")
    stats = torch.zeros(n, 8, dtype=torch.int32, device="cuda")
    st_ = gpu_lib.saber_k_mask_post(ptr(low.cuda()), n, x0, y0, cw, ch, H, W, thr, off, ptr(bits), ptr(stats), None)
    assert st_ == 0, gpu_lib.saber_k_last_error()
    torch.cuda.synchronize()
    got = unpack_bits(bits, W)
    st = stats.cpu().numpy()
    for i in range(n):
        assert np.logical_xor(got[i], ref_mask[i].numpy()).sum() <= 4, i
        assert abs(int(st[i, 1]) - int((full[i] > thr + off).sum())) <= 4 and abs(int(st[i, 2]) - int((full[i] > thr - off).sum())) <= 4
        assert int(st[i, 0]) == int(got[i].sum())
        if got[i].any():
            ys, xs = np.where(got[i])
            assert (st[i, 3], st[i, 4], st[i, 5], st[i, 6]) == (xs.min(), ys.min(), xs.max(), ys.max())


def test_prepare_rgb_against_reference_fixture():
    """(H,W,3) input: the reference's prepare filters over all three axes (tests/golden/saber_rgb_prepare.npz was written by the
    imported reference, oracle/make_golden_rgb.py)"""
    from saber_amd.utils import preprocessing as prep
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "saber_rgb_prepare.npz"))
    out = prep.prepare(g["rgb_in"], to_rgb=False)            # reference signature, weight-less K0 handle
    assert out.shape == g["rgb_in"].shape and out.dtype == torch.float32
    err = np.abs(out.cpu().numpy() - g["rgb_out"]).max()
    print("K0 rgb max abs diff", err)
    assert err < 2e-4
    gray = prep.prepare(g["rgb_in"][..., 0].copy(), to_rgb=True)
    from oracle import saber_ref
    assert np.abs(gray.cpu().numpy() - saber_ref.prepare(g["rgb_in"][..., 0])).max() < 2e-4


def test_sharded_driver_over_rccl_world_size_one(engine):
    """The NCCL (= RCCL) branch of segment_volume_sharded on real hardware: init_process_group('nccl'), uneven Z, device stitch.
    World size 1 on the one-GPU box; the collective call path (all_gather_into_tensor on a byte view) is the one N > 1 takes."""
    import torch.distributed as dist
    from saber_amd.segmenters.slice_driver import segment_volume_sharded
    from saber_amd.segmenters import utils
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        assert dist.get_backend() == "nccl"
        rng = np.random.default_rng(2)
        Z, H, W = 5, 96, 128
        vol = np.zeros((Z, H, W), np.uint16)
        zz, yy, xx = np.mgrid[:Z, :H, :W]
        for k in range(5):
            cz, cy, cx, r = rng.integers(0, Z), rng.integers(10, H - 10), rng.integers(10, W - 10), rng.integers(6, 14)
            vol[(zz - cz) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = k + 1
        dev_planes = torch.from_numpy(vol.view(np.int16)).cuda()
        out = segment_volume_sharded(vol, lambda z: dev_planes[z], stitch=True, min_mask_area=1, engine=engine)
        assert np.array_equal(out, utils.separate_masks(vol, min_mask_area=1))
        # and the collective itself, as the N > 1 branch issues it (byte view of int16 planes)
        full = torch.empty_like(dev_planes)
        dist.all_gather_into_tensor(full.view(torch.uint8), dev_planes.view(torch.uint8))
        torch.cuda.synchronize()
        assert torch.equal(full, dev_planes)
    finally:
        if created:
            dist.destroy_process_group()


def test_c_abi_call_leaves_callers_device_alone(engine):
    """entry points bind to the engine's device for the call only"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    cur = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(cur)) == 0
    before = cur.value
    engine.prepare(torch.zeros(64, 64, device="cuda"))
    assert hip.hipGetDevice(C.byref(cur)) == 0 and cur.value == before


@pytest.mark.parametrize("shape", [(3, 17, 1024), (5, 40, 100), (2, 9, 37), (70, 1024, 1024)])
def test_unpack_masks_against_numpy(gpu_lib, shape):
    """saber_k_unpack_masks (the bool arrays of the AMG dict list) against numpy.unpackbits on the same words, ragged widths included;
    70 x 1024^2 crosses the 64 MiB staging pass of engine.unpack_bits"""
    from saber_amd.engine import unpack_bits
    n, H, W = shape
    rng = np.random.default_rng(W + n)
    W32 = (W + 31) // 32
    words = rng.integers(0, 2 ** 32, size=(n, H, W32), dtype=np.uint64).astype(np.uint32)
    ref = np.unpackbits(words.view(np.uint8).reshape(n, H, -1), axis=-1, bitorder="little")[..., :W].astype(bool)
    got = unpack_bits(torch.from_numpy(words.view(np.int32)).cuda(), W)
    assert got.dtype == bool and got.shape == (n, H, W) and np.array_equal(got, ref)
    assert unpack_bits(torch.zeros((0, H, W32), dtype=torch.int32, device="cuda"), W).shape == (0, H, W)


def test_remove_duplicate_masks_device_rows_equal_host(engine):
    """utils.remove_duplicate_masks with the dicts' device rows attached (EngineMaskGenerator.generate) keeps exactly what the host form
    keeps from the bool arrays (saber/segmenters/utils.py:5-86), subsets and reorderings included"""
    from saber_amd.engine import unpack_bits
    from saber_amd.segmenters import utils
    rng = np.random.default_rng(3)
    n, H, W = 24, 128, 160
    base = rng.uniform(size=(8, H, W)) > 0.6
    masks = np.stack([base[i % 8] ^ (rng.uniform(size=(H, W)) > (0.999 if i % 3 else 0.9)) for i in range(n)])   # near and far copies
    pad = np.zeros((n, H, ((W + 31) // 32) * 32), bool)
    pad[..., :W] = masks
    bits = torch.from_numpy(np.packbits(pad, axis=-1, bitorder="little").view(np.int32).copy()).cuda()
    rows = utils.DeviceMaskRows(engine, bits, H, W)
    seg = unpack_bits(bits, W)
    stab = rng.uniform(0.9, 1.0, n)
    def dicts(with_rows, order):
        return [dict(segmentation=seg[i], area=int(seg[i].sum()), stability_score=float(stab[i]), tag=i,
                     **({utils.DEVICE_ROW_KEY: (rows, i)} if with_rows else {})) for i in order]
    for order in (list(range(n)), list(rng.permutation(n)), list(rng.permutation(n)[:11])):
        host = [m["tag"] for m in utils.remove_duplicate_masks(dicts(False, order))]
        dev = [m["tag"] for m in utils.remove_duplicate_masks(dicts(True, order))]
        assert host == dev and len(host) < len(order)
    mixed = dicts(True, range(n))
    mixed[3].pop(utils.DEVICE_ROW_KEY)                           # one dict without its row: the host form runs
    assert [m["tag"] for m in utils.remove_duplicate_masks(mixed)] == [m["tag"] for m in utils.remove_duplicate_masks(dicts(False, range(n)))]


def test_flat_slices_through_prepare_and_the_slice_step(engine):
    """Edge inputs of prep.prepare (saber/utils/preprocessing.py:4-37): a constant slice has zero local variance everywhere, and the
    reference's epsilons turn it into an all-zero image ((x - mean) / (0 + 1e-8) = 0, then (0 - 0) / (0 + 1e-8) = 0); a slice with a flat
    half keeps exact zeros there.  The engine's K0 must give the same images (no 1e8-amplified rounding residue), and the slice step must
    come back with a label plane, not an error."""
    from oracle import saber_ref
    from saber_amd.engine import make_amg_params
    from saber_amd.segmenters.slice_driver import segment_slice_to_plane
    rng = np.random.default_rng(12)
    const = np.full((1024, 1024), 32768, dtype=np.uint16)
    half = const.copy()
    half[:, 512:] = np.clip(rng.normal(32768, 3000, (1024, 512)), 0, 65535).astype(np.uint16)
    for name, sl in (("constant", const), ("flat half", half)):
        ref = saber_ref.prepare(sl.astype(np.float32))
        got = engine.prepare(torch.from_numpy(sl).cuda()).cpu().numpy()
        assert np.isfinite(got).all() and np.abs(got - ref).max() < 2e-4, (name, float(np.abs(got - ref).max()))
        if name == "constant":
            assert not ref.any() and not got.any()
        plane, n = segment_slice_to_plane(engine, torch.from_numpy(sl).cuda(), make_amg_params(dict(npoints=4, crop_n_layers=0)), min_mask_area=50)
        assert plane.shape == (1024, 1024) and plane.dtype == torch.uint16 and int(plane.cpu().numpy().max()) <= n


def test_m2m_src_assembled_in_kernel_equals_materialised(engine):
    """Round 3 experiment (opt-in, SABER_AMD_XBUILD=1; measured slower, see engine.hip / DESIGN.md): the m2m prompts' src = image_embed +
    mask-prompt embedding assembled tile by tile inside layer 0's dec_t2i / dec_i2t (XBuild) from the 16-channel hidden vectors instead of
    being written by mask_embed_src_kernel and read back twice: bit-identical outputs (same MFMA, same C operand, same bf16 rounding of
    the tile), single-slot and multi-slot batches, ragged prompt counts."""
    import os
    rng = np.random.default_rng(21)
    img = torch.from_numpy(rng.uniform(0, 1, (1024, 1024)).astype(np.float32)).cuda()
    engine.encode(img, [[0, 0, 1024, 1024], [100, 50, 700, 650]], slot0=0)
    for n in (1, 7, 32, 45):
        pts = torch.tensor(rng.uniform(0, 1024, (n, 2)).astype(np.float32)).cuda()
        mi = (torch.from_numpy(rng.normal(0, 6, (n, 256, 256)).astype(np.float32))).cuda()
        for slot in (0, 1):
            ref = engine.decode_points(pts, slot=slot, multimask=False, mask_input=mi)
            torch.cuda.synchronize()
            os.environ["SABER_AMD_XBUILD"] = "1"
            try:
                got = engine.decode_points(pts, slot=slot, multimask=False, mask_input=mi)
                torch.cuda.synchronize()
            finally:
                del os.environ["SABER_AMD_XBUILD"]
            for a, b in zip(got, ref):
                assert torch.equal(a, b), (n, slot)


def test_back_to_back_encodes_keep_their_own_crop_boxes(engine):
    """saber_encode stages the crop boxes in pinned memory and copies them asynchronously; two calls without a stream
    synchronisation in between must not see each other's boxes (ring of pinned slots guarded by events)."""
    rng = np.random.default_rng(33)
    img = torch.from_numpy(rng.uniform(0, 1, (1024, 1024)).astype(np.float32)).cuda()
    a, b = [0, 0, 512, 512], [300, 200, 1024, 900]
    engine.encode(img, [a], slot0=0)
    ref_a = engine.get_features(0)["image_embed"].clone()
    engine.encode(img, [b], slot0=0)
    ref_b = engine.get_features(0)["image_embed"].clone()
    torch.cuda.synchronize()
    for _ in range(3):                       # (more calls than would fit one slot)
        engine.encode(img, [a], slot0=0)
        engine.encode(img, [b], slot0=1)      # no synchronisation between the two
    fa, fb = engine.get_features(0)["image_embed"], engine.get_features(1)["image_embed"]
    torch.cuda.synchronize()
    assert torch.equal(fa, ref_a) and torch.equal(fb, ref_b)
