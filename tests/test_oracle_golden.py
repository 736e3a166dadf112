"""CPU: the oracle and the product's host glue against fixtures captured from the REFERENCE's own code
(tests/golden/saber_glue.npz, written by oracle/make_golden.py importing /root/reference) and against
HF-validated model outputs (tests/golden/sam2_tiny_seed0.npz, oracle/make_golden_model.py)."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "saber_glue.npz"), allow_pickle=False)


def test_oracle_prepare_matches_reference():
    from oracle import saber_ref
    raw = saber_ref.synthetic_slice(seed=int(G["prep_raw_seed"]))
    p = saber_ref.prepare(raw.astype(np.float32), to_rgb=True)
    assert p.dtype == np.float32 and p.shape == (1024, 1024, 3)
    assert np.array_equal(p[::16, ::16, 0], G["prep_out_sub"])          # bit-exact: same scipy calls
    assert np.allclose([p.min(), p.max(), p.mean(), p.std()], G["prep_out_stats"], rtol=0, atol=1e-7)
    assert bool(G["prep_out_rgb_equal"])
    small = G["prep_small_in"]
    assert np.array_equal(saber_ref.contrast(small, 3), G["prep_small_contrast"])
    assert np.array_equal(saber_ref.prepare(small), G["prep_small_out"])


def test_host_normalize_project_match_reference():
    from saber_amd.utils import preprocessing as prep
    from oracle import saber_ref
    vol = G["vol_in"]
    for fn in (prep.project_tomogram, saber_ref.project_tomogram):
        assert np.array_equal(fn(vol, 10, 3), G["proj_z10_d3"])
        assert np.array_equal(fn(vol, 1, 5), G["proj_z1_d5"])
        assert np.array_equal(fn(vol, 7, None), G["proj_z7"])
        assert np.array_equal(fn(vol), G["proj_all"])
    assert np.array_equal(prep.normalize(vol), G["vol_normalized"])


def test_z_gaussian_matches_reference():
    from saber_amd.segmenters.tomo import gaussian_smoothing_z, make_gaussian_kernel
    assert np.allclose(make_gaussian_kernel(5), G["gauss_kernel_s5"], rtol=0, atol=1e-7)
    out = gaussian_smoothing_z(G["vol_in"], 5, dim=0)
    assert np.allclose(out, G["gauss_z_s5"], rtol=0, atol=2e-6)  # conv1d vs correlate1d summation order


def test_remove_duplicate_masks_matches_reference():
    from saber_amd.segmenters import utils
    masks, stab = G["dedup_masks"], G["dedup_stab"]
    dicts = [{"segmentation": m, "area": int(m.sum()), "stability_score": float(s), "id": i} for i, (m, s) in enumerate(zip(masks, stab))]
    kept = utils.remove_duplicate_masks(dicts)
    assert [d["id"] for d in kept] == G["dedup_kept_ids"].tolist()
    assert len(kept) < len(dicts)  # the fixture does contain duplicates
    # the count-driven form used by the device path decides identically
    flat = masks.reshape(len(masks), -1).astype(np.float32)
    inter = np.rint(flat @ flat.T).astype(np.int64)
    keep = utils.duplicate_groups_from_counts([int(m.sum()) for m in masks], inter, stab.tolist())
    assert keep == G["dedup_kept_ids"].tolist()
    assert utils.remove_duplicate_masks([]) == []


def test_oracle_separate_masks_matches_reference():
    from oracle import saber_ref
    assert np.array_equal(saber_ref.separate_masks(G["sep_in"]), G["sep_out_default"])
    assert np.array_equal(saber_ref.separate_masks(G["sep_in"], min_mask_area=5), utils_sep5())
    assert np.array_equal(saber_ref.separate_masks(np.zeros((3, 8, 8), np.uint16)), G["sep_out_empty"])


def test_oracle_smoothing_matches_reference():
    """oracle restatement of fast_3d_gaussian_smoothing / gaussian_smoothing_3d against the reference's own outputs
    (tests/golden/saber_smooth.npz, oracle/make_golden_smooth.py)"""
    from oracle import saber_ref
    S = np.load(os.path.join(os.path.dirname(__file__), "golden", "saber_smooth.npz"), allow_pickle=False)
    for name, inp, scale in (("a_out_s075", "a_in", 0.075), ("a_out_s05", "a_in", 0.05), ("b_out_s05", "b_in", 0.05), ("c_out_s075", "c_in", 0.075)):
        out = saber_ref.fast_3d_gaussian_smoothing(S[inp], scale)
        assert out.dtype == np.uint8 and np.array_equal(out, S[name]), name
    assert set(np.unique(S["b_out_s05"])) == {0, 3, 300 & 255, 70, 200}      # uint8 result array: 300 and 515 wrap, 515 lands on 3
    assert np.array_equal(saber_ref.fast_3d_gaussian_smoothing(np.zeros((4, 8, 8), np.uint32)), S["empty_out"])
    for n, m in (("a3", S["a_in"] == 3), ("c1", S["c_in"] == 1)):
        assert np.abs(saber_ref.gaussian_smoothing_3d(m, float(S[f"field_{n}_sigma"])) - S[f"field_{n}"]).max() < 1e-6
    est = [saber_ref.estimate_feature_size_3d(S["a_in"] == k, 0.05) for k in range(1, 12)]
    assert np.array_equal(np.array(est), S["sigma_est"])
    assert len(saber_ref.gaussian_taps_3d(0.09)) == 1 and len(saber_ref.gaussian_taps_3d(1.5)) == 11


def utils_sep5():
    from saber_amd.segmenters import utils
    return utils.separate_masks(G["sep_in"], min_mask_area=5)


def test_separate_masks_matches_reference():
    from saber_amd.segmenters import utils
    assert np.array_equal(utils.separate_masks(G["sep_in"]), G["sep_out_default"])
    out5 = utils.separate_masks(G["sep_in"], min_mask_area=5)
    assert out5.dtype == np.uint32 and np.array_equal(out5, G["sep_out_min5"])
    assert out5.max() > 0
    assert np.array_equal(utils.separate_masks(np.zeros((3, 8, 8), np.uint16)), G["sep_out_empty"])


class _FakeAdapter:
    def __init__(self):
        self.masks, self.stab = G["sbs_masks"], G["sbs_stab"]

    def segment_image_2d(self, image, text_prompt=None, threshold=None):
        z = int(round(float(image[0, 0])))
        return [{"segmentation": m, "area": int(m.sum()), "stability_score": float(s), "bbox": [0, 0, 1, 1]}
                for m, s in zip(self.masks[z], self.stab[z])]

    def reset_state(self):
        pass


@pytest.fixture()
def fake_segmenter(monkeypatch):
    import saber_amd.segmenters.base as sbase
    from saber_amd.segmenters.propagation import propagationSegmenter
    from saber_amd.adapters.sam2.amg import cfgAMG
    monkeypatch.setattr(sbase, "get_adapter", lambda cfg, dev: _FakeAdapter())
    return propagationSegmenter(amg_cfg=cfgAMG(), min_mask_area=30)


def test_slice_by_slice_matches_reference(fake_segmenter):
    volume = np.zeros((6, 48, 64), dtype=np.float32)
    for z in range(6):
        volume[z] = z
    out = fake_segmenter.slice_by_slice(volume, None)
    assert out.dtype == np.uint32 and np.array_equal(out, G["sbs_out"])
    planes = np.zeros((6, 48, 64), dtype=np.uint16)
    for z in range(6):
        for idx, m in enumerate(fake_segmenter.segment_image(volume[z], display=False)):
            planes[z][m["segmentation"]] = idx + 1
    assert np.array_equal(planes, G["sbs_planes"])  # filter + dedup + ascending-area order + overwrite order


def test_sliding_windows_and_config_defaults(fake_segmenter):
    for shp in [(1024, 1024), (600, 900), (300, 300)]:
        assert np.array_equal(np.array(fake_segmenter.get_sliding_windows(shp)), G[f"windows_{shp[0]}x{shp[1]}"])
    from saber_amd.adapters.sam2.amg import cfgAMG
    from saber_amd.adapters.base import SAM2AdapterConfig
    d = cfgAMG().dict()
    assert sorted(d.keys()) == G["cfgamg_defaults_keys"].tolist()
    assert [str(d[k]) for k in sorted(d.keys())] == G["cfgamg_defaults_vals"].tolist()
    c = SAM2AdapterConfig()
    assert [c.model_type, c.cfg, str(c.checkpoint), str(c.num_maskmem), str(c.light_modality), str(c.min_mask_area)] == G["adaptercfg_defaults"].tolist()
    with pytest.raises(ValueError):
        SAM2AdapterConfig(cfg="huge")
    with pytest.raises(ValueError):
        cfgAMG(sam2_cfg="huge")
    with pytest.raises(ValueError):
        cfgAMG(npoints=0)


def test_oracle_model_reproduces_hf_validated_outputs():
    """BASELINE config 1: 512x512 micrograph, Hiera-tiny, one centre point prompt, CPU."""
    from oracle import sam2_ref
    from oracle.make_golden_model import config1_image
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    M = np.load(os.path.join(os.path.dirname(__file__), "golden", "sam2_tiny_seed0.npz"))
    assert M["hf_max_abs_diff"].max() < 5e-4  # the fixture was accepted only after agreeing with the HF restatement
    cfg = get_config("tiny")
    P = sam2_ref.ImagePredictorRef(seeded_weights(cfg, 0), cfg)
    P.set_image(config1_image())
    pts = P.transform_coords(torch.tensor([[256.0, 256.0]]), True, (512, 512))
    lab = torch.ones(1, 1, dtype=torch.int64)
    low, iou, obj, _, _ = P.predict_lowres(pts[:, None], lab, None, True)
    masks, _, lowc = P._predict(pts[:, None], lab, None, True)
    low2, iou2, _, _, _ = P.predict_lowres(pts[:, None], lab, lowc[:, :1], False)
    tol = dict(rtol=0, atol=2e-4)
    assert np.allclose(P.feats["image_embed"][0, ::8, ::4, ::4].numpy(), M["image_embed_sub"], **tol)
    assert np.allclose(P.feats["feat_s0"][0, ::4, ::16, ::16].numpy(), M["feat_s0_sub"], **tol)
    assert np.allclose(P.feats["feat_s1"][0, ::8, ::8, ::8].numpy(), M["feat_s1_sub"], **tol)
    assert np.allclose(low[0, :, ::4, ::4].numpy(), M["low_res_sub"], rtol=0, atol=1e-3)
    assert np.allclose(iou.numpy(), M["iou"], rtol=0, atol=1e-5) and np.allclose(obj.numpy(), M["obj"], rtol=0, atol=1e-4)
    assert np.allclose(low2[0, :, ::4, ::4].numpy(), M["m2m_low_res_sub"], rtol=0, atol=1e-3)
    assert masks.shape == (1, 3, 512, 512)
    assert np.abs((masks[0] > 0).sum((-1, -2)).numpy() - M["mask_area"]).max() <= 2


def test_prepare_rgb_restatement_matches_reference_fixture():
    """(H,W,3) input through prepare(): captured from the imported reference by oracle/make_golden_rgb.py"""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "saber_rgb_prepare.npz"))
    from oracle import saber_ref
    out = saber_ref.prepare(g["rgb_in"], to_rgb=False)
    assert out.shape == g["rgb_in"].shape and np.array_equal(out.astype(np.float32), g["rgb_out"])


# ------------------------------------------------------------------------------------------ classifier filter glue
def _cls_golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "saber_classifier_glue.npz"))


def test_classifier_crops_and_preprocess_match_reference():
    import torch
    from oracle import classifier_ref as cr
    G = _cls_golden()
    ti, tm = torch.from_numpy(G["image"]), torch.from_numpy(G["masks"])
    ic, mc = cr.crop_and_resize_adaptive(ti[None], tm[1])
    assert np.array_equal(ic.numpy(), G["crop1_image"]) and np.array_equal(mc.numpy(), G["crop1_mask"])
    ci, cm = cr.apply_crops(ti, tm)
    assert np.array_equal(ci.numpy()[7], G["crops_invalid_image"]) and np.array_equal((cm.numpy() > 0).sum(axis=(1, 2)), G["crops_mask_area"])
    batch, valid = cr.preprocess(ci, cm, 250)
    assert valid == G["valid"].tolist() and np.array_equal(batch.numpy(), G["batch"])
    mf = cr.mask_features(torch.from_numpy(G["feats"]), batch[:, 1:2])
    assert np.array_equal(mf.numpy(), G["masked_feats"])


def test_classifier_resolution_matches_reference():
    from oracle import classifier_ref as cr
    G = _cls_golden()
    dicts = [{"segmentation": m.astype(bool), "area": int(m.sum())} for m in G["masks"][:6]]
    inst = cr.convert_predictions_to_masks(G["pred"], list(dicts), 1, 32)
    assert len(inst) == int(G["inst_n"])
    assert np.array_equal(np.stack([m["segmentation"] for m in inst]).astype(np.uint8), G["inst_seg"])
    assert [m["area"] for m in inst] == G["inst_area"].tolist() and [m["bbox"] for m in inst] == G["inst_bbox"].tolist()
    assert np.allclose([m["predicted_iou"] for m in inst], G["inst_conf"], rtol=0, atol=1e-7)
    sem = cr.convert_predictions_to_masks(G["pred"], list(dicts), 0, 32)
    assert np.array_equal(np.stack([np.asarray(m["segmentation"]) for m in sem]).astype(np.uint8), G["sem_seg"])
    assert [m["area"] for m in sem] == G["sem_area"].tolist() and [m["label"] for m in sem] == G["sem_label"].tolist()


def test_classifier_head_restatement_equals_torch_modules():
    """The functional head of the oracle against torch.nn modules assembled in the layer order of the reference's source
    (saber/classifier/models/SAM2.py:58-92) and loaded through load_state_dict with the reference's parameter names."""
    import torch
    import torch.nn as nn
    from oracle import classifier_ref as cr

    class Head(nn.Module):
        def __init__(self, nc, h=256):
            super().__init__()
            d = [h, h // 2]
            self.projection = nn.Sequential(nn.Conv2d(512, d[0], 1), nn.BatchNorm2d(d[0]), nn.PReLU(), nn.Dropout2d(0.05),
                                            nn.Conv2d(d[0], d[0], 3, padding=1), nn.BatchNorm2d(d[0]), nn.PReLU(), nn.MaxPool2d(2, 2), nn.Dropout2d(0.1),
                                            nn.Conv2d(d[0], d[1], 3, padding=1), nn.BatchNorm2d(d[1]), nn.PReLU(), nn.MaxPool2d(2, 2), nn.Dropout2d(0.2))
            self.classifier = nn.Sequential(nn.Linear(d[1], 64), nn.LayerNorm(64), nn.PReLU(), nn.Dropout(0.1), nn.Linear(64, nc))

        def forward(self, f):
            f = self.projection(f)
            return self.classifier(nn.functional.adaptive_avg_pool2d(f, (1, 1)).view(f.size(0), -1))

    W = cr.seeded_head(3, 0)
    m = Head(3).eval()
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys)
    x = torch.from_numpy(np.random.default_rng(0).normal(0, 1, (2, 512, 64, 64)).astype(np.float32))
    with torch.no_grad():
        a = m(x)
        b = cr.head({k: torch.from_numpy(v) for k, v in W.items()}, x)
    assert torch.allclose(a, b, rtol=0, atol=1e-5)


def test_product_classifier_resolution_matches_reference():
    """saber_amd.filters.masks (host part of the classifier filter) against the reference's outputs."""
    from saber_amd.filters import masks as fm
    G = _cls_golden()
    dicts = [{"segmentation": m.astype(bool), "area": int(m.sum())} for m in G["masks"][:6]]
    inst = fm.convert_predictions_to_masks(G["pred"], list(dicts), 1, 32)
    assert np.array_equal(np.stack([m["segmentation"] for m in inst]).astype(np.uint8), G["inst_seg"])
    assert [m["area"] for m in inst] == G["inst_area"].tolist() and [m["bbox"] for m in inst] == G["inst_bbox"].tolist()
    assert np.allclose([m["predicted_iou"] for m in inst], G["inst_conf"], rtol=0, atol=1e-7)
    sem = fm.convert_predictions_to_masks(G["pred"], list(dicts), 0, 32)
    assert np.array_equal(np.stack([np.asarray(m["segmentation"]) for m in sem]).astype(np.uint8), G["sem_seg"])
    assert [m["area"] for m in sem] == G["sem_area"].tolist()


def test_amg_case_fixture_is_what_the_oracle_produces():
    """tests/golden/amg_cases_large_seed0.npz (the oracle side of the large-trunk AMG parity tests on the GPU) is regenerated for its
    cheapest case - fp32 oracle, one crop layer, 36 grid prompts + 108 m2m refinements - and compared mask by mask, so a change of the
    oracle cannot leave a stale fixture behind unnoticed."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    import make_golden_amg_cases as M
    from conftest import amg_case
    from oracle.amg_ref import amg_from_saber_cfg
    from oracle.sam2_ref import ImagePredictorRef
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    cfg = get_config("large")
    amg = dict(npoints=6, box_nms_thresh=1.0, pred_iou_thresh=0.5, stability_score_thresh=0.8, crop_n_layers=0)
    fresh = amg_from_saber_cfg(ImagePredictorRef(seeded_weights(cfg, 0), cfg), amg).generate(np.repeat(M.image_1024()[..., None], 3, 2))
    stored = amg_case("fp32_l0")
    assert len(fresh) == len(stored) > 50
    for f, s in zip(fresh, stored):
        a, b = f["segmentation"][2::4, 2::4], s["segmentation"]
        assert (a & b).sum() / max(1, (a | b).sum()) > 0.999
