"""GPU: the automatic mask generator at cfgAMG's OWN thresholds with both NMS stages ON (saber/adapters/sam2/amg.py:7-17: pred_iou 0.7,
stability 0.92 / offset 0.7, box NMS 0.7; upstream crop NMS 0.7), on weights whose masks are compact objects.

The seeded Hiera-L encoder with the mask decoder that oracle/fit_decoder_heads.py fitted on the synthetic slices
(tests/golden/decoder_fit_large_seed0.npz, saber_amd.weights.fitted_decoder_weights): predicted IoUs spread over [0, 0.9], stability scores
over [0.5, 1], objects of 1 000 - 30 000 pixels - so the score filters, the per-crop and the cross-crop box NMS, the duplicate removal, the
area sort and the paint order all do non-trivial work, which the seeded decoder's image-sized blobs never allowed.
  * exact mode against the fp32 oracle's committed result (oracle/make_golden_amg.py VARIANT=fitted): same masks, same ORDER;
  * fp16 (the adapter's default) and bf16 against the same golden: the masks well inside the thresholds all have a twin;
  * a tomogram through the product's z-loop: the device-resident loop equals the reference-shaped host loop bit for bit, with several
    label ids per plane (paint order / overwrite / dedup / ascending-area sort at work)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
AMG_DEFAULT = dict(npoints=32, crop_n_layers=2)             # everything else: cfgAMG's / upstream's defaults (saber_amd.engine.make_amg_params)


@pytest.fixture(scope="module")
def fitted_weights():
    from saber_amd.model_config import get_config
    from saber_amd.weights import fitted_decoder_weights
    cfg = get_config("large")
    return cfg, fitted_decoder_weights(cfg, 0)


def _run(eng, seed=0):
    from oracle import saber_ref
    from saber_amd.engine import make_amg_params, unpack_bits
    img = eng.prepare(torch.from_numpy(saber_ref.synthetic_slice(seed=seed)).cuda())
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        bits, meta = eng.amg_generate(img, make_amg_params(AMG_DEFAULT), max_masks=4096)
        st.synchronize()
    got = unpack_bits(bits, 1024)[:, 2::4, 2::4] if len(meta) else np.zeros((0, 256, 256), dtype=bool)
    return meta, got


def _match(G, meta, got):
    """best IoU (on the quarter-resolution samples) of every oracle mask among the engine's masks, and the index of that twin"""
    ref = np.unpackbits(G["quarter_bits"], axis=-1).astype(bool)
    n_ref = int(G["count"])
    if n_ref == 0 or len(meta) == 0:
        return np.zeros(n_ref), np.zeros(n_ref, dtype=int)
    gf, rf = got.reshape(len(meta), -1).astype(np.float32), ref.reshape(n_ref, -1).astype(np.float32)
    inter = rf @ gf.T
    iou = inter / np.maximum(rf.sum(1)[:, None] + gf.sum(1)[None] - inter, 1)
    return iou.max(1), iou.argmax(1)


def _golden():
    p = os.path.join(GOLD, "amg_default_grid_fitted_seed0.npz")
    if not os.path.exists(p):
        pytest.skip("golden not generated: VARIANT=fitted python -m oracle.make_golden_amg (~25 min of CPU)")
    return np.load(p)


def test_exact_mode_equals_oracle_at_cfgamg_defaults(fitted_weights):
    from saber_amd.engine import Engine
    G = _golden()
    cfg, W = fitted_weights
    eng = Engine("large", device=0, weights=W, max_images=21, max_prompts=1024, precision="exact")
    try:
        meta, got = _run(eng)
        n_ref, n_got = int(G["count"]), len(meta)
        best, twin = _match(G, meta, got)
        dev = 1.0 - best
        print(f"fitted decoder, cfgAMG defaults + both NMS, EXACT: oracle {n_ref} masks, engine {n_got}; |IoU - 1| median {np.median(dev):.2e} max {dev.max():.2e}")
        assert n_ref >= 5, "the golden itself is degenerate"
        assert n_got == n_ref
        assert np.array_equal(twin, np.arange(n_ref)), "the engine's masks are the oracle's, in the oracle's ORDER (NMS rank order, crop by crop, then the cross-crop NMS)"
        assert np.median(dev) <= 1e-3 and dev.max() < 2e-2
        piou = np.array([m.predicted_iou for m in meta]); stab = np.array([m.stability_score for m in meta])
        assert np.abs(piou - G["predicted_iou"]).max() < 1e-3 and np.abs(stab - G["stability_score"]).max() < 2e-3
        area = np.array([m.area for m in meta])
        assert np.abs(area - G["area"]).max() <= np.maximum(4, 2e-3 * G["area"]).max()
        assert np.allclose(np.array([list(m.crop_box_xywh) for m in meta]), G["crop_box"]) and np.allclose(np.array([list(m.point_xy) for m in meta]), G["point"])
    finally:
        eng.close()


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_production_arithmetic_against_oracle_at_cfgamg_defaults(fitted_weights, precision):
    """A candidate whose predicted IoU / stability score sits within the arithmetic's error of a threshold may fall on either side, and with box
    NMS on such a flip can also change WHICH of two overlapping candidates survives: the oracle masks that are clear of both thresholds (by
    3x the mode's end-to-end error) must each have a twin; the counts agree within the number of near-threshold masks."""
    from saber_amd.engine import Engine
    G = _golden()
    cfg, W = fitted_weights
    tol = {"fp16": 4e-3, "bf16": 2.5e-2}[precision]
    eng = Engine("large", device=0, weights=W, max_images=21, max_prompts=1024, precision=precision)
    try:
        meta, got = _run(eng)
        n_ref, n_got = int(G["count"]), len(meta)
        best, _ = _match(G, meta, got)
        clear = (G["predicted_iou"] > 0.7 + tol) & (G["stability_score"] > 0.92 + tol)
        print(f"fitted decoder, cfgAMG defaults + both NMS, {precision}: oracle {n_ref} masks, engine {n_got}; {int(clear.sum())} oracle masks clear of the thresholds by {tol}: "
              f"matched IoU min {best[clear].min() if clear.any() else float('nan'):.4f}; all: median {np.median(best):.4f}, > 0.9: {(best > 0.9).mean():.2f}")
        assert clear.sum() >= 2
        assert (best[clear] > 0.9).all()
        assert abs(n_got - n_ref) <= int((~clear).sum()) + 1
    finally:
        eng.close()


def test_tomogram_paints_many_ids_and_device_loop_equals_host_loop():
    """slice_by_slice (saber/segmenters/propagation.py:163-189) at cfgAMG's own thresholds on the fitted decoder: every plane carries several
    ids, so `idx + 1` in ascending-area order, later masks overwriting earlier ones, and remove_duplicate_masks (segmenters/utils.py:5-86)
    all act; the device-resident loop must reproduce the reference-shaped host loop bit for bit."""
    os.environ["SABER_AMD_SEEDED_WEIGHTS"] = "fitted"
    try:
        from oracle import saber_ref
        from saber_amd.adapters.base import SAM2AdapterConfig
        from saber_amd.adapters.sam2.amg import cfgAMG
        from saber_amd.segmenters.propagation import propagationSegmenter
        amg = cfgAMG(sam2_cfg="large")
        assert (amg.npoints, amg.crop_n_layers, amg.pred_iou_thresh, amg.stability_score_thresh, amg.box_nms_thresh) == (32, 2, 0.7, 0.92, 0.7)
        seg = propagationSegmenter(deviceID=0, cfg=SAM2AdapterConfig(cfg="large", amg_cfg=amg, min_mask_area=50), min_mask_area=50)
        vol = saber_ref.synthetic_volume(seed=1, depth=10)
        planes = seg.slice_by_slice_device(vol, stitch=False)
        per_plane = [int(p.max()) for p in planes]
        print("fitted decoder, cfgAMG defaults: label ids per plane", per_plane)
        assert np.median(per_plane) >= 10 and max(per_plane) >= 15       # ">= 10 ids per plane"
        # ids are list positions (propagation.py:185-186) - but NOT contiguous here: masks are painted in ascending-area order and a later
        # (larger) mask overwrites an earlier one, so an object nested in a larger mask of the same plane disappears from it.  With the
        # seeded decoder's blobs that never happened (tests/test_gpu_config2_volume.py asserts contiguity on those); here it must:
        gaps = [int(p.max()) - int((np.unique(p) > 0).sum()) for p in planes]
        print("ids painted over completely, per plane:", gaps)
        assert max(gaps) >= 1
        sub = np.ascontiguousarray(vol[[1, 2, 7]]).astype(np.float32)
        ref = seg.slice_by_slice(sub)
        dev = seg.slice_by_slice_device(sub)
        assert ref.max() >= 3 and np.array_equal(ref, dev)
        # paint order: where two masks of a plane overlap the LARGER one (painted later: ascending-area sort, propagation.py:185-186) wins.
        # Recomputed on the host from the adapter's mask list of one slice.
        masks = seg.segment_image(sub[0], display=False)
        assert len(masks) >= 3
        if masks:
            areas = [int(m["area"]) for m in masks]
            assert areas == sorted(areas)
            plane = np.zeros(sub[0].shape, dtype=np.uint16)
            for i, m in enumerate(masks):
                plane[m["segmentation"]] = i + 1
            assert np.array_equal(plane, seg.slice_by_slice_device(sub[:1], stitch=False)[0])
    finally:
        os.environ["SABER_AMD_SEEDED_WEIGHTS"] = "1"
