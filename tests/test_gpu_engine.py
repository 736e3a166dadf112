"""GPU parity of the engine (C-ABI in include/saber_amd.h) against the CPU oracle on the same seeded
weights and inputs.

Tolerances.  The engine runs GEMM operands in bf16 with fp32 accumulation, fp32 residual stream,
fp32 LayerNorm/softmax statistics.  Against the FP32 oracle this gives a relative RMS error of a few 1e-3
per block output that accumulates over 48 blocks; the bounds below are stated per tensor as relative RMS
(||a-b|| / ||b||) and are <= 2x the values measured on an MI355X, so a regression of 2x
fails.  That this residual IS the sanctioned bf16 operand rounding and not a kernel defect is shown separately:
tests/test_gpu_parity_bf16.py compares the engine with an oracle that rounds where the engine rounds and
asserts the north star's 1e-3 there.  Mask-level parity is stated as IoU of thresholded masks.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


@pytest.fixture(scope="module")
def image():
    rng = np.random.default_rng(7)
    img = rng.uniform(0, 1, (1024, 1024)).astype(np.float32)
    yy, xx = np.mgrid[:1024, :1024]
    for _ in range(10):
        cy, cx = rng.integers(100, 924, 2)
        r = rng.integers(30, 120)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] *= 0.3
    return img


@pytest.fixture(scope="module")
def oracle_feats(image, oracle_large):
    from oracle import sam2_ref
    cfg, W = oracle_large
    torch.set_num_threads(max(1, torch.get_num_threads()))
    taps = {"blocks": []}
    with torch.no_grad():
        pix = sam2_ref.sam2_transforms(np.repeat(image[..., None], 3, 2))
        feats = sam2_ref.encode_image(W, cfg, pix, taps)
    return feats, taps


def test_encode_parity(engine, image, oracle_feats):
    feats, _ = oracle_feats
    engine.encode(torch.from_numpy(image).cuda())
    got = engine.get_features(0)
    torch.cuda.synchronize()
    errs = {k: rel_rms(got[k].cpu(), feats[k][0]) for k in ("image_embed", "feat_s0", "feat_s1")}
    print("encoder rel-rms:", errs)
    assert errs["feat_s0"] < 5.6e-3 and errs["feat_s1"] < 8.4e-3 and errs["image_embed"] < 1.2e-2, errs   # measured 2.8e-3 / 4.2e-3 / 6.0e-3


def test_encode_crops_and_rgb(engine, image, oracle_large):
    """A crop is resized to 1024^2 exactly as SAM2Transforms does; RGB input takes the 3-channel path."""
    from oracle import sam2_ref
    cfg, W = oracle_large
    rgb = np.stack([image, image[::-1].copy(), image[:, ::-1].copy()], axis=-1).copy()
    crop = [100, 200, 697, 797]
    with torch.no_grad():
        pix = sam2_ref.sam2_transforms(rgb[crop[1]:crop[3], crop[0]:crop[2]])
        feats = sam2_ref.encode_image(W, cfg, pix)
    engine.encode(torch.from_numpy(rgb).cuda(), [crop], slot0=1)
    got = engine.get_features(1)
    err = rel_rms(got["image_embed"].cpu(), feats["image_embed"][0])
    print("crop/rgb image_embed rel-rms:", err)
    assert err < 1.3e-2


def test_decode_parity(engine, image, oracle_large, oracle_feats):
    from oracle import sam2_ref
    cfg, W = oracle_large
    feats, _ = oracle_feats
    engine.encode(torch.from_numpy(image).cuda())
    rng = np.random.default_rng(3)
    pts = torch.tensor(rng.uniform(0, 1024, (8, 2)).astype(np.float32))
    lab = torch.ones(8, 1, dtype=torch.int64)
    low, iou, obj = engine.decode_points(pts.cuda(), slot=0, multimask=True)
    torch.cuda.synchronize()
    # (a) decoder in isolation: oracle decoder fed with the ENGINE's features
    gf = {k: v.cpu()[None] for k, v in engine.get_features(0).items()}
    with torch.no_grad():
        sp, de = sam2_ref.prompt_encoder(W, pts[:, None], lab, None)
        r_low, r_iou, r_obj, _, _ = sam2_ref.mask_decoder(W, gf, sp, de, True)
    e_low, e_iou, e_obj = rel_rms(low.cpu(), r_low), (iou.cpu() - r_iou).abs().max().item(), (obj.cpu() - r_obj[:, 0]).abs().max().item()
    print("decoder-only: low-res rel-rms", e_low, "iou abs", e_iou, "obj abs", e_obj)
    assert e_low < 1.1e-2 and e_iou < 1.2e-2 and e_obj < 4e-2            # measured 5.5e-3 / 5.6e-3 / ~1e-2
    agree = ((low.cpu() > 0) == (r_low > 0)).float().mean().item()
    assert agree > 0.997, agree
    # (b) end to end against the oracle's own features
    with torch.no_grad():
        o_low, o_iou, _, _, _ = sam2_ref.mask_decoder(W, feats, sp, de, True)
    print("end-to-end: low-res rel-rms", rel_rms(low.cpu(), o_low), "iou abs", (iou.cpu() - o_iou).abs().max().item())
    assert rel_rms(low.cpu(), o_low) < 1.6e-2                            # measured 7.8e-3
    # (c) m2m pass: mask prompt + dynamic single-mask selection
    mi = torch.clamp(low[:, 0], -32, 32).contiguous()
    low2, iou2, _ = engine.decode_points(pts.cuda(), slot=0, multimask=False, mask_input=mi)
    with torch.no_grad():
        sp, de = sam2_ref.prompt_encoder(W, pts[:, None], lab, mi.cpu()[:, None])
        r_low2, r_iou2, _, _, _ = sam2_ref.mask_decoder(W, gf, sp, de, False)
    e2 = rel_rms(low2.cpu(), r_low2)
    print("m2m decoder-only: low-res rel-rms", e2, "iou abs", (iou2.cpu() - r_iou2).abs().max().item())
    assert e2 < 1.2e-2


def _match_masks(a, b):
    """greedy IoU matching of two lists of bool masks -> list of best IoUs for masks of b"""
    out = []
    for mb in b:
        best = 0.0
        for ma in a:
            inter = np.logical_and(ma, mb).sum()
            uni = np.logical_or(ma, mb).sum()
            best = max(best, inter / uni if uni else 1.0)
        out.append(best)
    return out


@pytest.mark.parametrize("layers,nms", [(0, 1.0), (1, 0.95)])
def test_amg_parity(engine, image, layers, nms):
    """engine AMG against the fp32 oracle's (oracle/sam2_ref + oracle/amg_ref on the same seeded weights and image; the oracle side is
    the committed result of oracle/make_golden_amg_cases.py - it costs 30-100 s of host time per case on the GPU box)"""
    from conftest import amg_case
    from saber_amd.engine import make_amg_params, unpack_bits
    # nms 1.0 disables box suppression (IoU > 1 never holds): every mask that passes the score filters is compared
    amg = dict(npoints=6, crop_n_layers=layers, box_nms_thresh=nms, pred_iou_thresh=0.5, stability_score_thresh=0.8)
    ref = amg_case(f"fp32_l{layers}")
    bits, meta = engine.amg_generate(torch.from_numpy(image).cuda(), make_amg_params(amg), max_masks=512)
    got = unpack_bits(bits, 1024)
    print(f"AMG layers={layers}: oracle {len(ref)} masks, engine {len(meta)} masks")
    assert len(ref) > 0
    # bf16 vs fp32 logits can flip a borderline filter decision (the emulating-oracle test is the tight one): the sets must agree
    # up to 5 % (measured: same count) and matched masks must coincide (measured median IoU 0.994)
    assert abs(len(ref) - len(meta)) <= max(2, int(0.05 * len(ref)))
    ious = _match_masks(list(got[:, 2::4, 2::4]), [r["segmentation"] for r in ref])      # quarter-resolution samples of both sides
    good = np.mean(np.array(ious) > 0.97)
    print("matched IoU: median", float(np.median(ious)), "fraction>0.97", float(good))
    assert good >= 0.9 and np.median(ious) >= 0.988
    for m, g in zip(meta, got):
        assert m.area == int(g.sum())
        ys, xs = np.where(g)
        assert [m.bbox_xywh[0], m.bbox_xywh[1], m.bbox_xywh[2], m.bbox_xywh[3]] == [xs.min(), ys.min(), xs.max() - xs.min(), ys.max() - ys.min()]


def test_amg_parity_non_square_image(engine):
    """ragged input: a 600 x 840 image (crop boxes, point grids, bilinear up-sampling to a non-square crop, bit rows that are not a
    multiple of 32 wide)"""
    from conftest import amg_case
    from saber_amd.engine import make_amg_params, unpack_bits
    rng = np.random.default_rng(21)
    H, W = 600, 840
    img = rng.uniform(0, 1, (H, W)).astype(np.float32)
    yy, xx = np.mgrid[:H, :W]
    for _ in range(8):
        cy, cx, r = rng.integers(60, H - 60), rng.integers(60, W - 60), rng.integers(25, 90)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] *= 0.3
    amg = dict(npoints=6, crop_n_layers=0, box_nms_thresh=1.0, pred_iou_thresh=0.5, stability_score_thresh=0.8)
    ref = amg_case("fp32_ragged")          # the fp32 oracle's AMG on this image (oracle/make_golden_amg_cases.py)
    bits, meta = engine.amg_generate(torch.from_numpy(img).cuda(), make_amg_params(amg), max_masks=512)
    got = unpack_bits(bits, W)
    print(f"non-square AMG: oracle {len(ref)} masks, engine {len(meta)} masks")
    assert len(ref) >= 5 and got.shape[1:] == (H, W)
    assert abs(len(ref) - len(meta)) <= max(2, int(0.05 * len(ref)))
    ious = _match_masks(list(got[:, 2::4, 2::4]), [r["segmentation"] for r in ref])      # quarter-resolution samples of both sides
    print("non-square matched IoU: median", float(np.median(ious)), "fraction>0.97", float(np.mean(np.array(ious) > 0.97)))
    assert np.mean(np.array(ious) > 0.97) >= 0.9
    for m, g in zip(meta, got):
        assert m.area == int(g.sum())


def test_label_plane(engine):
    rng = np.random.default_rng(0)
    H, W = 96, 160
    masks = rng.uniform(size=(5, H, W)) > 0.7
    packed = np.packbits(masks, axis=-1, bitorder="little").view(np.uint32).astype(np.int32)
    order = [3, 0, 4, 1, 2]
    plane = engine.label_plane(torch.from_numpy(packed).cuda(), order, H, W).cpu().numpy()
    ref = np.zeros((H, W), np.uint16)
    for i, mi in enumerate(order):  # reference paint loop: saber/segmenters/propagation.py:185-186
        ref[masks[mi]] = i + 1
    assert np.array_equal(plane, ref)


def test_errors_surface_as_exceptions(engine):
    with pytest.raises(ValueError):  # crop box outside the image
        engine.encode(torch.zeros(64, 64).cuda(), [[0, 0, 65, 64]])
    with pytest.raises(ValueError):  # more crops than resident slots
        engine.encode(torch.zeros(64, 64).cuda(), [[0, 0, 64, 64]] * (engine.max_images + 1))
    from saber_amd.engine import Engine
    with pytest.raises(ValueError):
        Engine("huge")


# ------------------------------------------------------------------------------------------------ the other trunks
# SAM2AdapterConfig.cfg defaults to "small" (saber/adapters/base.py:11); tiny/small/base+ use 14x14 / 7x7 windows that the
# reference pads (70^2 / 35^2 grids) and head dims 96 / 56: the engine keeps those stages in a padded window-major layout.
@pytest.fixture(scope="module", params=["tiny", "small", "base"])
def trunk_case(request):
    from oracle import sam2_ref
    from saber_amd.engine import Engine
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    cfg = get_config(request.param)
    Wnp = seeded_weights(cfg, 0)
    eng = Engine(request.param, device=0, weights=Wnp, max_images=2, max_prompts=8)
    yield request.param, cfg, sam2_ref.to_torch(Wnp), eng
    eng.close()


def test_encode_decode_parity_other_trunks(trunk_case, image):
    from oracle import sam2_ref
    name, cfg, W, eng = trunk_case
    with torch.no_grad():
        feats = sam2_ref.encode_image(W, cfg, sam2_ref.sam2_transforms(np.repeat(image[..., None], 3, 2)))
    # slot 1 of a 2-image batch: the padded layout's per-image row bookkeeping is exercised too
    eng.encode(torch.from_numpy(image).cuda(), [[0, 0, 512, 512], [0, 0, 1024, 1024]], slot0=0)
    got = eng.get_features(1)
    torch.cuda.synchronize()
    errs = {k: rel_rms(got[k].cpu(), feats[k][0]) for k in ("image_embed", "feat_s0", "feat_s1")}
    print(name, "encoder rel-rms:", errs)
    assert errs["feat_s0"] < 5.0e-3 and errs["feat_s1"] < 8.0e-3 and errs["image_embed"] < 1.2e-2, errs   # measured 2.5e-3 / 4.0e-3 / 5.0-5.8e-3
    pts = torch.tensor([[300.0, 420.0], [800.0, 128.0], [512.0, 512.0]])
    lab = torch.ones(3, 1, dtype=torch.int64)
    low, iou, obj = eng.decode_points(pts.cuda(), slot=1, multimask=True)
    with torch.no_grad():
        sp, de = sam2_ref.prompt_encoder(W, pts[:, None], lab, None)
        o_low, o_iou, _, _, _ = sam2_ref.mask_decoder(W, feats, sp, de, True)
    e = rel_rms(low.cpu(), o_low)
    print(name, "end-to-end low-res rel-rms", e, "iou abs", (iou.cpu() - o_iou).abs().max().item())
    assert e < 2e-2 and (iou.cpu() - o_iou).abs().max().item() < 2e-2       # measured 8e-3 - 1e-2


def test_config1_tiny_against_hf_validated_golden():
    """BASELINE configs[0]: 512x512 micrograph (the reference's own synthetic test recipe), Hiera-tiny, ONE point prompt at the image
    centre, multimask.  tests/golden/sam2_tiny_seed0.npz holds the oracle's outputs that the independent HF Sam2Model reproduced."""
    import os
    from oracle.make_golden_model import config1_image
    from saber_amd.engine import Engine
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    M = np.load(os.path.join(os.path.dirname(__file__), "golden", "sam2_tiny_seed0.npz"))
    eng = Engine("tiny", device=0, weights=seeded_weights(get_config("tiny"), 0), max_images=1, max_prompts=8)
    try:
        eng.encode(torch.from_numpy(config1_image()).cuda())
        f = eng.get_features(0)
        pts = torch.tensor([[512.0, 512.0]])  # (256, 256) of the 512^2 image in model pixels
        low, iou, obj = eng.decode_points(pts.cuda(), slot=0, multimask=True)
        torch.cuda.synchronize()
        g = {"image_embed": f["image_embed"][::8, ::4, ::4], "feat_s0": f["feat_s0"][::4, ::16, ::16], "feat_s1": f["feat_s1"][::8, ::8, ::8]}
        errs = {k: rel_rms(g[k].cpu(), torch.from_numpy(M[k + "_sub"])) for k in g}
        e_low = rel_rms(low[0, :, ::4, ::4].cpu(), torch.from_numpy(M["low_res_sub"]))
        e_iou = float(np.abs(iou.cpu().numpy() - M["iou"]).max())
        print("config 1 (tiny) vs golden:", errs, "low-res", e_low, "iou", e_iou)
        assert errs["feat_s0"] < 6e-3 and errs["feat_s1"] < 8e-3 and errs["image_embed"] < 1.1e-2, errs     # measured 2.9e-3 - 5.3e-3
        assert e_low < 1.7e-2 and e_iou < 7e-3                                                            # measured 8.2e-3 / 3.3e-3
        sign = ((low[0, :, ::4, ::4].cpu().numpy() > 0) == (M["low_res_sub"] > 0)).mean()
        assert sign > 0.99, sign
        # m2m refinement of the first mask (mask prompt = clamped low-res logits), single-mask output
        mi = torch.clamp(low[:, 0], -32, 32).contiguous()
        low2, iou2, _ = eng.decode_points(pts.cuda(), slot=0, multimask=False, mask_input=mi)
        e2 = rel_rms(low2[0, :, ::4, ::4].cpu(), torch.from_numpy(M["m2m_low_res_sub"]))
        print("config 1 m2m low-res", e2)
        assert e2 < 3.6e-2                                                                                # measured 1.8e-2
    finally:
        eng.close()
