"""GPU: the precision modes on weights with the activation statistics of TRAINED checkpoints (saber_amd.weights.stress_weights: outlier
LayerNorm gains x30, two massive residual channels, a large pos_embed) instead of the well-conditioned random initialisation every other
model-parity number is measured on (no real SAM2.1 checkpoint can be fetched offline).

Asserted: the exact mode stays within the north star's 1e-3 of the fp32 oracle on these weights.  Reported and bounded at 2x measured:
the price of bf16 operands and of the e4m3 weight format relative to the exact mode on the same handle / same weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


def test_precision_modes_on_stress_weights():
    from oracle import saber_ref, sam2_ref
    from saber_amd.engine import Engine
    from saber_amd.model_config import get_config
    from saber_amd.weights import stress_weights
    cfg = get_config("large")
    Wnp = stress_weights(cfg, 0)
    W = sam2_ref.to_torch(Wnp)
    img = saber_ref.prepare(saber_ref.synthetic_slice(seed=4).astype(np.float32))
    taps = {"blocks": []}
    with torch.no_grad():
        feats = sam2_ref.encode_image(W, cfg, sam2_ref.sam2_transforms(np.repeat(img[..., None], 3, 2)), taps)
    x_last = taps["blocks"][cfg.stage_ends[2]]                              # end of stage 2 (the massive channels live in stages 1 and 2)
    absx = x_last.abs().flatten(0, 2)
    top = absx.max(0).values.topk(4).values
    print("stress weights: |x| of the four largest residual channels at the end of stage 2:", [round(float(v), 1) for v in top],
          "median channel max:", round(float(absx.max(0).values.median()), 2))
    assert float(top[1]) > 8 * float(absx.max(0).values.median())          # the massive channels did form
    t = torch.from_numpy(img).cuda()
    eng = Engine("large", device=0, weights=Wnp, max_images=1, max_prompts=16, precision="exact")
    eng8 = Engine("large", device=0, weights=Wnp, max_images=1, max_prompts=16, weight_format="fp8")
    try:
        rng = np.random.default_rng(11)
        pts = torch.tensor(rng.uniform(0, 1024, (8, 2)).astype(np.float32))
        lab = torch.ones(8, 1, dtype=torch.int64)
        with torch.no_grad():
            sp, de = sam2_ref.prompt_encoder(W, pts[:, None], lab, None)
            o_low, o_iou, _, _, _ = sam2_ref.mask_decoder(W, feats, sp, de, True)
        eng.encode(t)
        fx = {k: v.clone() for k, v in eng.get_features(0).items()}
        lx, ix, _ = eng.decode_points(pts.cuda(), slot=0, multimask=True)
        ex = {k: rel_rms(fx[k].cpu(), feats[k][0]) for k in fx}
        e_low = rel_rms(lx.cpu(), o_low)
        print("EXACT vs fp32 oracle on stress weights: features", ex, f"low-res logits {e_low:.2e}, iou abs {(ix.cpu() - o_iou).abs().max().item():.2e}")
        assert max(ex.values()) < 1e-3 and e_low < 1e-3 and (ix.cpu() - o_iou).abs().max().item() < 1e-3
        eng.set_precision("bf16")
        eng.encode(t)
        fb = eng.get_features(0)
        lb, ib, _ = eng.decode_points(pts.cuda(), slot=0, multimask=True)
        eb = {k: rel_rms(fb[k], fx[k]) for k in fx}
        print("price of bf16 on stress weights (vs exact): features", eb, f"low-res logits {rel_rms(lb, lx):.2e}, iou abs {(ib - ix).abs().max().item():.2e}, "
              f"mask sign agreement {((lb > 0) == (lx > 0)).float().mean().item():.5f}")
        eng8.encode(t)
        f8 = eng8.get_features(0)
        l8, i8, _ = eng8.decode_points(pts.cuda(), slot=0, multimask=True)
        e8 = {k: rel_rms(f8[k], fx[k]) for k in fx}
        print("price of e4m3 weights on stress weights (vs exact): features", e8, f"low-res logits {rel_rms(l8, lx):.2e}, iou abs {(i8 - ix).abs().max().item():.2e}, "
              f"mask sign agreement {((l8 > 0) == (lx > 0)).float().mean().item():.5f}")
        engm = Engine("large", device=0, weights=Wnp, max_images=1, max_prompts=16, weight_format="mxfp8")
        try:
            engm.encode(t)
            fm = engm.get_features(0)
            lm, im_, _ = engm.decode_points(pts.cuda(), slot=0, multimask=True)
            em = {k: rel_rms(fm[k], fx[k]) for k in fx}
            em_low = rel_rms(lm, lx)
            print("price of MXFP8 operands (fp8 MFMA) on stress weights (vs exact): features", em, f"low-res logits {em_low:.2e}, iou abs {(im_ - ix).abs().max().item():.2e}, "
                  f"mask sign agreement {((lm > 0) == (lx > 0)).float().mean().item():.5f}")
        finally:
            engm.close()
        # measured on an MI355X: image_embed 3.6e-2, low-res logits 2.35e-2, sign agreement 0.9923 (per-row e4m3 WEIGHTS with bf16 activations: 2.8e-2 /
        # 2.2e-2 / 0.9929 - the block scales keep the massive channels' blocks from costing the other blocks their resolution); bounds 2x measured
        assert em["image_embed"] < 7.3e-2 and em_low < 4.7e-2
        # bounds: 2x the values measured on an MI355X: bf16 4.4e-3 / 4.9e-3 / 6.1e-3 (image_embed / feat_s0 / feat_s1),
        # low-res logits 8.0e-3; e4m3 weights: image_embed 2.8e-2, low-res logits 2.2e-2
        assert eb["image_embed"] < 9e-3 and eb["feat_s0"] < 1e-2 and eb["feat_s1"] < 1.25e-2 and rel_rms(lb, lx) < 1.6e-2
        assert e8["image_embed"] < 6e-2 and rel_rms(l8, lx) < 4.5e-2
    finally:
        eng.close()
        eng8.close()
