"""Where a drop-in saber2D.segment_image call spends its time once a slice carries a few hundred masks: the AMG on the device, the
bool arrays handed to the host (unpack), duplicate removal.  Seeded weights; score filters set as bench.py's tail mode sets them
(pred_iou_thresh = the slice's own quantile keeping ~250 masks, stability / NMS off).  Usage: python tools/adapter_hostprof.py [n_keep]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    keep = int(sys.argv[1]) if len(sys.argv) > 1 else 250
    from saber_amd.adapters.sam2.automask import EngineMaskGenerator
    from saber_amd.engine import Engine, make_amg_params, unpack_bits
    from saber_amd.model_config import get_config
    from saber_amd.segmenters import utils
    from saber_amd.weights import seeded_weights
    cfg = get_config("large")
    eng = Engine("large", device=0, weights=seeded_weights(cfg, 0), max_images=21, max_prompts=1024)
    rng = np.random.default_rng(0)
    img = torch.from_numpy(rng.uniform(0, 1, (1024, 1024)).astype(np.float32)).cuda()
    base = dict(npoints=32, crop_n_layers=2, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0)
    _, meta = eng.amg_generate(img, make_amg_params(dict(base, pred_iou_thresh=0.0)), max_masks=16384)
    ious = np.sort(np.array([m.predicted_iou for m in meta]))
    gen = EngineMaskGenerator(eng, dict(base, pred_iou_thresh=float(ious[-keep])), max_masks=4096)
    gen.generate(img)

    def t(f, reps=3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, r

    ms_dev, (bits, meta) = t(lambda: gen.generate_device(img))
    ms_unpack, seg = t(lambda: unpack_bits(bits, 1024))
    ms_gen, masks = t(lambda: gen.generate(img))
    ms_dd_dev, kept_dev = t(lambda: utils.remove_duplicate_masks(masks))
    plain = [{k: v for k, v in m.items() if k != utils.DEVICE_ROW_KEY} for m in masks]
    ms_dd_host, kept_host = t(lambda: utils.remove_duplicate_masks(plain), reps=1)
    assert [id(m["segmentation"]) for m in kept_dev] == [id(m["segmentation"]) for m in kept_host]
    print({"masks": len(meta), "amg_device_ms": round(ms_dev, 1), "unpack_to_host_ms": round(ms_unpack, 1), "generate_ms": round(ms_gen, 1),
           "dedup_device_rows_ms": round(ms_dd_dev, 1), "dedup_host_matmul_ms": round(ms_dd_host, 1), "kept": len(kept_dev)})


if __name__ == "__main__":
    main()
