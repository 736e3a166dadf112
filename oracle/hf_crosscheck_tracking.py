"""ORACLE validation of the TRACKING LOOP (run in the authoring container; never imported by the product): oracle/sam2_video_ref.py's
VideoPredictorRef against the independent `transformers` Sam2VideoModel + Sam2VideoInferenceSession with shared weights, on the reference's
own synthetic tomogram recipe - mask prompt on one frame, propagation forward and backward, num_maskmem = 2 as SABER sets it.  What this
pins beyond the per-module checks of hf_crosscheck_video.py: which memories and object pointers a tracked frame attends to, their temporal
encodings and order, the pointer projection, best-of-three mask selection, object-score gating (NO_OBJ_SCORE, no_obj_ptr,
no_obj_embed_spatial), memory encoding of tracked frames, both propagation directions.

    python -m oracle.hf_crosscheck_tracking            # tiny trunk, 5 frames
"""
import sys

import numpy as np
import torch

from saber_amd.model_config import get_config
from saber_amd.weights import param_specs, seeded_weights
from oracle import sam2_video_ref as V
from oracle.hf_crosscheck import hf_config, key_map


def build(trunk: str = "tiny", seed: int = 0, num_maskmem: int = 2):
    from transformers import Sam2VideoConfig, Sam2VideoModel
    cfg = get_config(trunk)
    W = seeded_weights(cfg, seed, video=True)
    k = "sam_mask_decoder.pred_obj_score_head.layers.2.bias"
    W[k] = W[k] + np.float32(3.0)                 # the seeded head otherwise says "absent" on every frame
    vc = hf_config(cfg).vision_config
    model = Sam2VideoModel(Sam2VideoConfig(vision_config=vc, num_maskmem=num_maskmem)).eval()
    sd = model.state_dict()
    touched = set()
    for name in param_specs(cfg):                 # image encoder, prompt encoder, mask decoder, no_mem_embed
        for hk, tr in key_map(name):
            t = torch.from_numpy(np.asarray(W[name], dtype=np.float32))
            if tr is None:
                assert sd[hk].shape == t.shape, (name, hk, tuple(sd[hk].shape), tuple(t.shape))
                sd[hk].copy_(t)
            else:
                sd[hk][tr[1]].copy_(t[0])
            touched.add(hk)
    # memory attention / memory encoder: HF's (randomised) tensors are the shared truth, exported under upstream names
    g = torch.Generator().manual_seed(100 + seed)
    for hk, v in sd.items():
        if hk.startswith(("memory_attention.", "memory_encoder.")) and v.dtype.is_floating_point and "rotary" not in hk:
            v.copy_(torch.randn(v.shape, generator=g) * (0.5 if v.ndim == 1 else 1.0 / max(1, v[0].numel()) ** 0.5))
            touched.add(hk)
    W = dict(W)
    W.update({k2: v.numpy() for k2, v in V.from_hf_memory_attention({k[len("memory_attention."):]: v for k, v in sd.items() if k.startswith("memory_attention.")}).items()})
    W.update({k2: v.numpy() for k2, v in V.from_hf_memory_encoder({k[len("memory_encoder."):]: v for k, v in sd.items() if k.startswith("memory_encoder.")}).items()})
    # the small video tensors by name
    extra = {"memory_temporal_positional_encoding": np.asarray(W["maskmem_tpos_enc"])[:num_maskmem],
             "no_memory_positional_encoding": np.zeros((1, 1, 256), np.float32),
             "no_object_pointer": np.asarray(W["no_obj_ptr"]).reshape(1, 256),
             "mask_downsample.weight": W["mask_downsample.weight"], "mask_downsample.bias": W["mask_downsample.bias"],
             "object_pointer_proj.proj_in.weight": W["obj_ptr_proj.layers.0.weight"], "object_pointer_proj.proj_in.bias": W["obj_ptr_proj.layers.0.bias"],
             "object_pointer_proj.layers.0.weight": W["obj_ptr_proj.layers.1.weight"], "object_pointer_proj.layers.0.bias": W["obj_ptr_proj.layers.1.bias"],
             "object_pointer_proj.proj_out.weight": W["obj_ptr_proj.layers.2.weight"], "object_pointer_proj.proj_out.bias": W["obj_ptr_proj.layers.2.bias"],
             "temporal_positional_encoding_projection_layer.weight": W["obj_ptr_tpos_proj.weight"],
             "temporal_positional_encoding_projection_layer.bias": W["obj_ptr_tpos_proj.bias"],
             "occlusion_spatial_embedding_parameter": np.asarray(W["no_obj_embed_spatial"]).reshape(1, 64)}
    for hk, a in extra.items():
        t = torch.from_numpy(np.asarray(a, dtype=np.float32))
        assert sd[hk].shape == t.shape, (hk, tuple(sd[hk].shape), tuple(t.shape))
        sd[hk].copy_(t)
        touched.add(hk)
    missing = [k for k in sd if k not in touched and "rotary" not in k]
    assert not missing, f"HF tensors left at their initial values: {missing[:8]}"
    model.load_state_dict(sd)
    return cfg, W, model


@torch.no_grad()
def crosscheck(trunk: str = "tiny", Z: int = 5, start: int = 2, verbose: bool = True):
    from transformers.models.sam2_video.modeling_sam2_video import Sam2VideoInferenceSession
    cfg, W, model = build(trunk)
    rng = np.random.default_rng(42)
    tomo = rng.uniform(-1, 1, (Z, 128, 128)).astype(np.float32)                 # saber/adapters/sam3/tests/test_tomogram_predictor.py:67-68
    yy, xx = np.mgrid[:128, :128]
    seed_mask = ((yy - 64) ** 2 + (xx - 64) ** 2 < (128 // 6) ** 2).astype(np.float32)
    frames = V.load_tomogram_frames(tomo)                                        # (Z,3,1024,1024), what the adapter feeds the predictor
    # ---- oracle
    P = V.VideoPredictorRef(W, cfg, num_maskmem=2, cond_memory_from_full_res_mask=True)
    P.init_state(frames, video_hw=(1024, 1024))
    P.add_new_mask(start, 1, seed_mask)
    ref = {}
    for rev in (False, True):
        for t, ids, logits in P.propagate_in_video(start, None, reverse=rev):
            ref[(t, rev)] = logits.clone()
    # ---- transformers
    sess = Sam2VideoInferenceSession(video=frames, video_height=1024, video_width=1024, dtype=torch.float32)
    obj_idx = sess.obj_id_to_idx(1)
    m = torch.from_numpy(seed_mask)[None, None]
    m = (torch.nn.functional.interpolate(m, size=(1024, 1024), mode="bilinear", align_corners=False, antialias=True) >= 0.5).float()
    sess.add_mask_inputs(obj_idx, start, m)
    sess.obj_with_new_inputs = [1]
    model(inference_session=sess, frame_idx=start)
    got = {}
    for rev in (False, True):
        for out in model.propagate_in_video_iterator(sess, start_frame_idx=start, reverse=rev):
            got[(out.frame_idx, rev)] = (out.pred_masks.float(), out.object_score_logits.float())
    assert set(got) == set(ref), (sorted(got), sorted(ref))
    worst = 0.0
    for key in sorted(ref):
        t, rev = key
        low_ref = (P.out[1]["cond"].get(t) or P.out[1]["non_cond"][t])
        d_low = (got[key][0].reshape(256, 256) - low_ref["pred_masks"].reshape(256, 256)).abs().max().item()
        d_obj = abs(float(got[key][1].reshape(-1)[0]) - float(low_ref["object_score_logits"].reshape(-1)[0]))
        scale = low_ref["pred_masks"].abs().max().item()
        worst = max(worst, d_low / max(scale, 1e-6))
        if verbose:
            print(f"frame {t} {'bwd' if rev else 'fwd'}: low-res logits max|diff| {d_low:.3e} (scale {scale:.1f}), object score |diff| {d_obj:.3e}")
    return worst


if __name__ == "__main__":
    w = crosscheck(sys.argv[1] if len(sys.argv) > 1 else "tiny")
    print(f"worst relative difference of the low-res logits: {w:.3e}")
