"""Frames per second of the SAM2 video path: SAM2Adapter.segment_volume on a synthetic tomogram, Hiera-L, seeded weights
(object-score bias +3 so that the seeded head reports 'present').  One object seeded in the middle slice, forward + backward propagation.
    python tools/video_bench.py [--frames 32] [--trunk large]
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(trunk="large", frames=32, size=256, reps=2, window=16):
    from saber_amd.adapters.base import SAM2AdapterConfig
    from saber_amd.adapters.sam2.predictor import SAM2Adapter
    from saber_amd.adapters.sam2.video import VideoPredictor
    from saber_amd.engine import Engine
    from saber_amd.model_config import get_config
    from saber_amd.weights import param_specs, seeded_weights
    cfg = get_config(trunk)
    W = seeded_weights(cfg, 0, video=True)
    k = "sam_mask_decoder.pred_obj_score_head.layers.2.bias"
    W[k] = W[k] + np.float32(3.0)
    img_keys = set(param_specs(cfg).keys())
    eng = Engine(trunk, device=0, weights={n: v for n, v in W.items() if n in img_keys}, max_images=window, max_prompts=8)
    vp = VideoPredictor(eng, W, num_maskmem=2)
    rng = np.random.default_rng(42)
    tomo = rng.uniform(-1, 1, (frames, size, size)).astype(np.float32)
    yy, xx = np.mgrid[:size, :size]
    seed = ((yy - size // 2) ** 2 + (xx - size // 2) ** 2 < (size // 6) ** 2).astype(np.float32)
    ad = SAM2Adapter(SAM2AdapterConfig(cfg=trunk), device="cuda:0")
    ad._video_predictor = vp
    times = []
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ad.set_volume(tomo)                     # host tomogram -> (Z,1024,1024) frame stack in HBM
        vol = ad.segment_volume(frames // 2, masks=[seed], min_presence_score=0.0)      # -> (Z,H,W) uint16 on the host
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = min(times[1:])
    eng.profile_begin()
    ad.set_volume(tomo)
    ad.segment_volume(frames // 2, masks=[seed], min_presence_score=0.0)
    prof = eng.profile_end()
    out = {"what": f"SAM2Adapter.set_volume + segment_volume (host tomogram in, host label volume out), {trunk} trunk, {window} frames per encoder pass, {frames} frames of {size}x{size} (resized to 1024^2), one object, forward + backward",
           "frames_per_s": frames / dt, "ms_per_frame": dt / frames * 1e3, "voxels": int((vol > 0).sum()),
           "engine_kernel_classes_ms_per_frame": {k2: round(v["ms"] / frames, 3) for k2, v in prof.items() if v["launches"]}}
    eng.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--trunk", default="large")
    a = ap.parse_args()
    print(json.dumps(run(a.trunk, a.frames)))
