"""Worker of tests/test_gpu_video.py::test_window_encodes_sharded_over_ranks: one rank of a gloo group on cuda:0 (or the single-process
baseline when WORLD_SIZE is unset) runs SAM2Adapter.set_volume + segment_volume on the reference's synthetic recipe and saves what it got."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out_path):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    from saber_amd.adapters.base import SAM2AdapterConfig
    from saber_amd.adapters.sam2.predictor import SAM2Adapter
    from saber_amd.adapters.sam2.video import VideoPredictor
    from saber_amd.engine import Engine
    from saber_amd.model_config import get_config
    from saber_amd.weights import param_specs, seeded_weights
    cfg = get_config("tiny")
    W = seeded_weights(cfg, 0, video=True)
    W["sam_mask_decoder.pred_obj_score_head.layers.2.bias"] = W["sam_mask_decoder.pred_obj_score_head.layers.2.bias"] + np.float32(3.0)
    img_keys = set(param_specs(cfg).keys())
    eng = Engine("tiny", device=0, weights={k: v for k, v in W.items() if k in img_keys}, max_images=3, max_prompts=8)
    vp = VideoPredictor(eng, W, num_maskmem=2)
    rng = np.random.default_rng(42)
    tomo = rng.uniform(-1, 1, (7, 128, 128)).astype(np.float32)
    yy, xx = np.mgrid[:128, :128]
    seed = ((yy - 64) ** 2 + (xx - 64) ** 2 < (128 // 6) ** 2).astype(np.float32)
    ad = SAM2Adapter(SAM2AdapterConfig(cfg="tiny"), device="cuda:0")
    ad._video_predictor = vp
    ad.set_volume(tomo)
    vol = ad.segment_volume(3, masks=[seed], min_presence_score=0.0)
    torch.cuda.synchronize()
    np.savez(out_path, vol=vol, scores=ad.frame_scores, sharded=np.array(vp._ranks()[0]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main(sys.argv[1])
