"""Generates tests/golden/sam2_tiny_seed0.npz: outputs of oracle/sam2_ref.py for the seeded Hiera-tiny weights on
BASELINE config 1 (512x512 synthetic micrograph of the reference's own test recipe,
saber/adapters/sam3/tests/test_sam3_image.py:308-315; 1 point prompt at the image centre), AFTER asserting that the
independent `transformers` Sam2Model restatement reproduces them (authoring container only).

    python -m oracle.make_golden_model
"""
import os

import numpy as np
import torch

from oracle import sam2_ref
from oracle.hf_crosscheck import hf_config, load_into_hf
from saber_amd.model_config import get_config
from saber_amd.weights import seeded_weights

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sam2_tiny_seed0.npz")


def config1_image():
    rng = np.random.default_rng(42)
    img = rng.uniform(0, 0.2, (512, 512)).astype(np.float32)
    yy, xx = np.mgrid[:512, :512]
    for _ in range(8):
        cy, cx = rng.integers(60, 452, 2)
        r = rng.integers(20, 50)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] += 0.6
    return np.repeat(img[..., None], 3, 2)


def main():
    from transformers import Sam2Model
    cfg = get_config("tiny")
    W = seeded_weights(cfg, 0)
    P = sam2_ref.ImagePredictorRef(W, cfg)
    img = config1_image()
    P.set_image(img)
    pts = P.transform_coords(torch.tensor([[256.0, 256.0]]), True, (512, 512))
    lab = torch.ones(1, 1, dtype=torch.int64)
    low, iou, obj, _, _ = P.predict_lowres(pts[:, None], lab, None, True)
    masks, _, lowc = P._predict(pts[:, None], lab, None, True)
    low2, iou2, _, _, _ = P.predict_lowres(pts[:, None], lab, lowc[:, :1], False)
    model = load_into_hf(Sam2Model(hf_config(cfg)).eval(), W)
    with torch.no_grad():
        pix = sam2_ref.sam2_transforms(img)
        emb = model.get_image_embeddings(pix)
        out = model(image_embeddings=emb, input_points=pts[None, :, None], input_labels=lab[None].int(), multimask_output=True)
        out2 = model(image_embeddings=emb, input_points=pts[None, :, None], input_labels=lab[None].int(), input_masks=lowc[:, :1], multimask_output=False)
    checks = {"image_embed": (P.feats["image_embed"] - emb[2]).abs().max().item(), "low": (low - out.pred_masks[0]).abs().max().item(),
              "iou": (iou - out.iou_scores[0]).abs().max().item(), "obj": (obj - out.object_score_logits[0]).abs().max().item(),
              "m2m_low": (low2 - out2.pred_masks[0]).abs().max().item()}
    print(checks)
    assert max(checks.values()) < 5e-4, checks
    np.savez_compressed(OUT, image_embed_sub=P.feats["image_embed"][0, ::8, ::4, ::4].numpy(), feat_s0_sub=P.feats["feat_s0"][0, ::4, ::16, ::16].numpy(),
                        feat_s1_sub=P.feats["feat_s1"][0, ::8, ::8, ::8].numpy(), low_res_sub=low[0, :, ::4, ::4].numpy(), iou=iou.numpy(), obj=obj.numpy(),
                        mask_area=(masks[0] > 0).sum((-1, -2)).numpy(), m2m_low_res_sub=low2[0, :, ::4, ::4].numpy(), m2m_iou=iou2.numpy(),
                        hf_max_abs_diff=np.array(list(checks.values())))
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
