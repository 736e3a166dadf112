"""Classifier filter on the engine: crops and area filter against the reference's own outputs (golden fixture), the head
against the fp32 oracle on identical embeddings, Predictor.predict / batch_predict end to end against the oracle, and the saber2D hook."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "saber_classifier_glue.npz"))
NC = 3


@pytest.fixture(scope="module")
def setup():
    from oracle import classifier_ref as cr
    from saber_amd.classifier.models.predictor import Predictor
    from saber_amd.engine import Engine
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    cfg = get_config("tiny")
    W = seeded_weights(cfg, 0)
    eng = Engine("tiny", weights=W, max_images=4, max_prompts=64)
    Wh = cr.seeded_head(NC, 0)
    config = {"model": {"num_classes": NC}, "amg_params": {"sam2_cfg": "tiny"}}
    pred = Predictor(None, None, config=config, head_weights=Wh, engine=eng)
    ref = cr.PredictorRef(W, cfg, Wh, NC)
    return eng, pred, ref, cr


def test_crops_and_area_filter_match_reference_outputs(setup):
    eng, pred, ref, cr = setup
    probs = pred.predict(G["image"], G["masks"])
    crops, cm = pred.last_crops(len(G["masks"]))
    torch.cuda.synchronize()
    valid = G["valid"].tolist()
    # the fixture holds the reference's crops of the RAW image (apply_crops); predict z-scores the image first (NormalizeIntensity) and
    # bilinear resizing commutes with that affine map up to fp32 rounding (values of a few units -> a few 1e-6)
    img = torch.from_numpy(G["image"])
    want = (G["batch"][:, 0] - float(img.mean())) / float(img.std(unbiased=False))
    err = np.abs(crops.cpu().numpy()[valid] - want).max()
    assert err < 2e-5, err
    assert np.array_equal(cm.cpu().numpy()[valid], G["batch"][:, 1].astype(np.uint8))
    assert np.array_equal((cm.cpu().numpy() > 0).sum(axis=(1, 2)), G["crops_mask_area"])
    invalid = [i for i in range(len(G["masks"])) if i not in valid]
    assert np.all(probs[invalid] == 0) and np.allclose(probs[valid].sum(axis=1), 1.0, atol=1e-5)


def test_head_matches_oracle_on_identical_embeddings(setup):
    eng, pred, ref, cr = setup
    k = 3
    rng = np.random.default_rng(11)
    emb = torch.from_numpy(rng.normal(0, 0.5, (k, 256, 64, 64)).astype(np.float32))
    emb = emb.to(torch.bfloat16).float()                       # bf16-representable: the first GEMM's operand rounding is then exact
    masks = torch.from_numpy(G["batch"][:k, 1].astype(np.uint8))
    dummy = torch.zeros((64, 64), dtype=torch.float32, device="cuda")
    eng.encode(dummy, crop_boxes=[(0, 0, 64, 64)] * k)
    keep = []
    for b in range(k):
        t = emb[b].permute(1, 2, 0).reshape(4096, 256).contiguous().cuda()
        keep.append(t)
        eng._check(eng.lib.saber_set_embed_tokens(eng.h, b, C.c_void_p(t.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    probs = pred.head(masks)
    with torch.no_grad():
        taps = {}
        logits = cr.head(ref.Wh, cr.mask_features(emb, masks[:, None].float()), taps)
        want = torch.softmax(logits, dim=1).numpy()
    err = np.abs(probs - want).max()
    print("head probs max|diff|", err, "probs", want[0])
    assert err < 1e-3, err                                      # measured 4.9e-4 (bf16 operands of three GEMMs, fp32 accumulate)
    assert np.array_equal(probs.argmax(1), want.argmax(1))


def test_predict_end_to_end_against_oracle(setup):
    eng, pred, ref, cr = setup
    sel = [0, 1, 3, 5]                                          # three ordinary masks + the 4x3-pixel one (filtered by min_area after resize?)
    masks = G["masks"][sel]
    got = pred.batch_predict(G["image"], masks, batch_size=3)   # two groups: 3 + 1
    want = ref.batch_predict(G["image"], masks, batch_size=3)
    err = np.abs(got - want).max()
    print("e2e probs max|diff|", err, got, want)
    assert np.array_equal(got.sum(1) > 0, want.sum(1) > 0)
    assert err < 2.5e-3, err                                    # measured 1.06e-3 (bf16 encoder + head against the fp32 oracle)
    assert np.array_equal(got.argmax(1)[want.max(1) - np.sort(want, 1)[:, -2] > 0.05], want.argmax(1)[want.max(1) - np.sort(want, 1)[:, -2] > 0.05])


def test_saber2d_hook_runs_the_classifier(setup):
    eng, pred, ref, cr = setup
    from saber_amd.filters import masks as fm
    dicts = [{"segmentation": m.astype(bool), "area": int(m.sum()), "bbox": [0, 0, 1, 1]} for m in G["masks"][:6]]
    out = fm.apply_classifier(G["image"], list(dicts), pred, 1, 32)
    want = cr.apply_classifier(G["image"], list(dicts), ref, 1, 32)
    assert len(out) == len(want)
    for a, b in zip(out, want):
        assert np.array_equal(a["segmentation"], b["segmentation"]) and a["area"] == b["area"] and abs(a["predicted_iou"] - b["predicted_iou"]) < 2.5e-3
    sem = fm.apply_classifier(G["image"], list(dicts), pred, 0, 32)
    assert [m["label"] for m in sem] == [1, 2]


def test_reference_script_assertions_finite_and_repeatable(setup):
    """The reference's classifier test script (saber/classifier/tests/test_sam3_classifier.py:73-77,144-145): probabilities of the right
    shape, finite, and two runs on the same input equal (there within 1e-4; here bit-identical)."""
    eng, pred, ref, cr = setup
    a = pred.predict(G["image"], G["masks"][:3])
    b = pred.predict(G["image"], G["masks"][:3])
    assert a.shape == (3, NC) and np.isfinite(a).all() and np.array_equal(a, b)
