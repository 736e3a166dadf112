"""hipGraph replay of the AMG launch sequences (BASELINE configs[4]): replayed results are bit-identical to the eager
ones, the sequences really are captured and replayed, and switching to an image of another size or to other AMG parameters falls back to
eager for the first sight of the new shapes instead of replaying a stale graph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(eng, img, params):
    bits, meta = eng.amg_generate(img, params, max_masks=8192)
    torch.cuda.synchronize()
    return bits.cpu().numpy().copy(), [(m.predicted_iou, m.stability_score, m.area, tuple(m.bbox_xywh), tuple(m.point_xy), tuple(m.crop_box_xywh)) for m in meta]


def test_graph_replay_is_bit_identical_to_eager():
    from oracle import saber_ref
    from saber_amd.engine import Engine, make_amg_params
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    cfg = get_config("tiny")
    w = seeded_weights(cfg, 0)
    eng = Engine("tiny", weights=w, max_images=5, max_prompts=256)
    ref = Engine("tiny", weights=w, max_images=5, max_prompts=256)
    ref.set_graphs(False)
    params = make_amg_params(dict(npoints=8, crop_n_layers=1, pred_iou_thresh=0.0, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0))
    imgs = [eng.prepare(torch.from_numpy(saber_ref.synthetic_slice(seed=s, size=512)).cuda()) for s in (1, 2, 3)]
    # slice 0: eager (first sight), slice 1: captured and launched, slice 2 and 3: replayed
    for k, i in enumerate((0, 1, 2, 0)):
        b, m = _run(eng, imgs[i], params)
        b0, m0 = _run(ref, imgs[i], params)
        assert len(m) > 0 and m == m0, f"call {k}"
        assert np.array_equal(b, b0), f"call {k}"
        cap, rep = eng.graph_stats()
        if k == 0:
            assert (cap, rep) == (0, 0)
        if k == 1:
            assert cap > 0 and rep == 0
        if k >= 2:
            assert rep > 0
    assert ref.graph_stats() == (0, 0)
    ncap = eng.graph_stats()[0]
    # other parameters (fewer grid points) and another image size: new sequences, first eager then captured; the old graphs stay valid
    p2 = make_amg_params(dict(npoints=4, crop_n_layers=0, pred_iou_thresh=0.0, stability_score_thresh=0.0, box_nms_thresh=1.0))
    small = eng.prepare(torch.from_numpy(saber_ref.synthetic_slice(seed=4, size=256)).cuda())
    for img, p in ((imgs[0], p2), (small, params), (imgs[0], p2), (small, params), (imgs[1], params)):
        b, m = _run(eng, img, p)
        b0, m0 = _run(ref, img, p)
        assert m == m0 and np.array_equal(b, b0)
    assert eng.graph_stats()[0] > ncap


def test_amg_result_does_not_depend_on_the_encoder_batch_size():
    """the crops of a slice go through the encoder max_images at a time; with fewer slots than crops (two encoder batches, the second
    one holding a single layer-1 crop) the driver must return the same masks as with all crops in one batch"""
    from oracle import saber_ref
    from saber_amd.engine import Engine, make_amg_params
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    w = seeded_weights(get_config("tiny"), 0)
    params = make_amg_params(dict(npoints=8, crop_n_layers=1, pred_iou_thresh=0.0, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0))
    img = None
    results = []
    for mi in (5, 4, 2):
        eng = Engine("tiny", weights=w, max_images=mi, max_prompts=256)
        if img is None:
            img = eng.prepare(torch.from_numpy(saber_ref.synthetic_slice(seed=7, size=512)).cuda())
        results.append(_run(eng, img, params))
        eng.close()
    for b, m in results[1:]:
        assert m == results[0][1] and np.array_equal(b, results[0][0])


def test_iou_pruning_of_the_m2m_pass_changes_nothing_but_the_time():
    """saber_engine_set_iou_pruning: m2m candidates whose four IoU predictions are all <= pred_iou_thresh skip the mask upscaling (they cannot
    pass the generator's IoU filter whichever plane the dynamic multimask selection picks).  Masks, order and metadata must be identical with
    pruning on and off, at a threshold that prunes a part of the candidates (the quantile of the unfiltered IoUs), eagerly and under graph
    replay; with the threshold at 0 nothing is pruned."""
    from oracle import saber_ref
    from saber_amd.engine import Engine, make_amg_params
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    cfg = get_config("tiny")
    w = seeded_weights(cfg, 0)
    eng = Engine("tiny", weights=w, max_images=5, max_prompts=256)
    ref = Engine("tiny", weights=w, max_images=5, max_prompts=256)
    ref.set_iou_pruning(False)
    try:
        img = eng.prepare(torch.from_numpy(saber_ref.synthetic_slice(seed=4, size=512)).cuda())
        base = dict(npoints=8, crop_n_layers=1, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0)
        _, meta = eng.amg_generate(img, make_amg_params(dict(base, pred_iou_thresh=0.0)), max_masks=8192)
        assert eng.last_pruning()[0] == 0                          # (no threshold: nothing can be pruned)
        ious = np.sort(np.array([m.predicted_iou for m in meta]))
        for q in (0.5, 0.9):
            thr = float(ious[int(q * len(ious))])
            params = make_amg_params(dict(base, pred_iou_thresh=thr))
            for rep in range(3):                         # eager, captured, replayed
                b, m = _run(eng, img, params)
                pruned, seen = eng.last_pruning()
                b0, m0 = _run(ref, img, params)
                assert ref.last_pruning() == (0, 0)
                assert m == m0 and np.array_equal(b, b0) and len(m) > 0
                assert 0 < pruned < seen
            print(f"pred_iou_thresh {thr:.4f} (quantile {q}): {len(m)} masks, {pruned} of {seen} m2m candidates pruned before the upscaling")
    finally:
        eng.close(); ref.close()


def test_device_side_filters_and_nms_equal_the_host_restatement():
    """saber_engine_set_device_amg: IoU / stability / crop-edge filters, the per-crop box NMS, the cross-crop NMS and the compaction as
    kernels (amg_device.hip) against the host restatement they replace (amg.hip), which is itself pinned to the oracle's AMG driver by
    tests/test_gpu_engine.py / test_gpu_parity_bf16.py: identical masks, order and records over filter / NMS settings that exercise every
    branch (NMS on and off, one crop and three layers of crops, single-mask and multimask, with and without the m2m pass), ONE host
    synchronisation per call instead of three."""
    from oracle import saber_ref
    from saber_amd.engine import Engine, make_amg_params
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    cfg = get_config("tiny")
    w = seeded_weights(cfg, 0)
    dev = Engine("tiny", weights=w, max_images=21, max_prompts=256)
    host = Engine("tiny", weights=w, max_images=21, max_prompts=256)
    host.set_device_amg(False)
    try:
        img = dev.prepare(torch.from_numpy(saber_ref.synthetic_slice(seed=6, size=512)).cuda())
        _, meta = host.amg_generate(img, make_amg_params(dict(npoints=8, crop_n_layers=1, pred_iou_thresh=0.0, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0)), max_masks=8192)
        ious = np.sort(np.array([m.predicted_iou for m in meta]))
        stabs = np.sort(np.array([m.stability_score for m in meta]))
        q = lambda a, f: float(a[int(f * (len(a) - 1))])
        # (the seeded model's masks are image-sized blobs: box NMS at the usual 0.7 leaves one mask, so the thresholds that exercise partial
        # suppression sit close to 1)
        cases = [dict(npoints=8, crop_n_layers=1, pred_iou_thresh=0.0, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0),
                 dict(npoints=8, crop_n_layers=1, pred_iou_thresh=q(ious, 0.3), stability_score_thresh=q(stabs, 0.3), box_nms_thresh=0.7, crop_nms_thresh=0.7),
                 dict(npoints=8, crop_n_layers=1, pred_iou_thresh=q(ious, 0.2), stability_score_thresh=0.0, box_nms_thresh=0.97, crop_nms_thresh=0.9),
                 dict(npoints=8, crop_n_layers=1, pred_iou_thresh=0.0, stability_score_thresh=q(stabs, 0.2), box_nms_thresh=0.99, crop_nms_thresh=0.98),
                 dict(npoints=8, crop_n_layers=2, pred_iou_thresh=q(ious, 0.5), stability_score_thresh=0.0, box_nms_thresh=0.995, crop_nms_thresh=0.95),
                 dict(npoints=16, crop_n_layers=0, pred_iou_thresh=q(ious, 0.2), stability_score_thresh=q(stabs, 0.5), box_nms_thresh=0.98, crop_nms_thresh=0.7),
                 dict(npoints=8, crop_n_layers=1, pred_iou_thresh=q(ious, 0.4), stability_score_thresh=0.0, box_nms_thresh=0.99, crop_nms_thresh=0.97, multimask_output=False),
                 dict(npoints=8, crop_n_layers=1, pred_iou_thresh=0.0, stability_score_thresh=q(stabs, 0.6), box_nms_thresh=0.985, crop_nms_thresh=0.99, use_m2m=False),
                 dict(npoints=8, crop_n_layers=1, pred_iou_thresh=0.999, stability_score_thresh=0.0, box_nms_thresh=0.7, crop_nms_thresh=0.7)]      # nothing survives
        for ci, c in enumerate(cases):
            params = make_amg_params(c)
            for rep in range(2):                                     # eager and captured / replayed decode sequences
                b, m = _run(dev, img, params)
                sd = dev.lib.saber_amg_last_syncs(dev.h)
                b0, m0 = _run(host, img, params)
                sh = host.lib.saber_amg_last_syncs(host.h)
                assert m == m0, f"case {ci}: records differ ({len(m)} vs {len(m0)})"
                assert np.array_equal(b[:len(m)], b0[:len(m0)]), f"case {ci}: masks differ"
            print(f"AMG case {ci}: {len(m)} masks, host synchronisations {sd} (device path) vs {sh} (host path)")
            assert sd == 1 and sh >= 2
        # too small a capacity: the call reports the count it needs and Engine.amg_generate retries with it, as on the host path
        bits, meta2 = dev.amg_generate(img, make_amg_params(cases[0]), max_masks=4)      # 4 < count: Engine retries with the reported capacity
        assert len(meta2) == len(meta)
    finally:
        dev.close(); host.close()
