import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import saber_ref
from saber_amd.engine import Engine, make_amg_params
from saber_amd.model_config import get_config
from saber_amd.weights import seeded_weights
cfg = get_config("large"); W = seeded_weights(cfg, 0)
eng = Engine("large", weights=W, max_images=21, max_prompts=1024)
img = eng.prepare(torch.from_numpy(saber_ref.synthetic_slice(seed=0)).cuda())
base = dict(npoints=32, crop_n_layers=2, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    _, meta = eng.amg_generate(img, make_amg_params(dict(base, pred_iou_thresh=0.0)), max_masks=16384)
    ious = np.sort(np.array([m.predicted_iou for m in meta])); thr = float(ious[-250])
    params = make_amg_params(dict(base, pred_iou_thresh=thr))
    for dev in (True, False):
        eng.set_device_amg(dev)
        for _ in range(2): eng.amg_generate(img, params, max_masks=4096)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): b, m = eng.amg_generate(img, params, max_masks=4096)
        torch.cuda.synchronize(); print("device" if dev else "host", len(m), (time.perf_counter() - t0) / 4 * 1e3, "ms", flush=True)
    from saber_amd.segmenters.slice_driver import segment_slice_to_plane
    pool = [torch.from_numpy(saber_ref.synthetic_slice(seed=i)).cuda() for i in range(2)]
    for dev in (True, False, True, False):
        eng.set_device_amg(dev)
        segment_slice_to_plane(eng, pool[0], params, min_mask_area=50, max_masks=4096)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(4): pl, n = segment_slice_to_plane(eng, pool[i % 2], params, min_mask_area=50, max_masks=4096)
        torch.cuda.synchronize(); print("slice->plane", "device" if dev else "host", n, (time.perf_counter() - t0) / 4 * 1e3, "ms", flush=True)
        t0 = time.perf_counter()
        for i in range(4):
            img2 = eng.prepare(pool[i % 2]); b, m = eng.amg_generate(img2, params, max_masks=4096)
        torch.cuda.synchronize(); print("   amg_generate only", len(m), (time.perf_counter() - t0) / 4 * 1e3, "ms", flush=True)
