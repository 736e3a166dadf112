"""GPU: parity that means something.  The engine runs matrix products on bf16 operands with fp32 accumulation;
against the fp32 oracle that costs ~6e-3 relative RMS on the encoder output.  oracle/sam2_bf16_emul.py is an independent torch
restatement that applies the SAME roundings at the same places, so

    engine  vs  bf16-emulating oracle   <= 1e-3  (north-star tolerance: only fp32 summation order and rare rounding flips remain)
    bf16-emulating oracle vs fp32 oracle ~ 6e-3  (the sanctioned precision itself; asserted on the CPU in test_oracle_bf16_emul.py)

A kernel defect (wrong tile, stale LDS read, dropped k-step) shows up in the first line; it cannot hide under the second.
Bounds are <= 2x the values measured on an MI355X (printed by the tests, recorded in DESIGN.md section 3).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item()


@pytest.fixture(scope="module")
def image():
    rng = np.random.default_rng(7)
    img = rng.uniform(0, 1, (1024, 1024)).astype(np.float32)
    yy, xx = np.mgrid[:1024, :1024]
    for _ in range(10):
        cy, cx = rng.integers(100, 924, 2)
        r = rng.integers(30, 120)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] *= 0.3
    return img


@pytest.fixture(scope="module")
def emul_feats(image, oracle_large):
    from oracle import sam2_ref, sam2_bf16_emul as E
    cfg, W = oracle_large
    pix = sam2_ref.sam2_transforms(np.repeat(image[..., None], 3, 2))
    return E.encode_image_emul(W, cfg, pix)


def test_encoder_vs_bf16_emulating_oracle(engine, image, emul_feats):
    engine.encode(torch.from_numpy(image).cuda())
    got = engine.get_features(0)
    torch.cuda.synchronize()
    errs = {k: rel_rms(got[k].cpu(), emul_feats[k][0]) for k in ("image_embed", "feat_s0", "feat_s1")}
    print("encoder vs bf16-emulating oracle, rel-rms:", errs)
    assert max(errs.values()) < 1e-3, errs


def test_decoder_vs_bf16_emulating_oracle(engine, image, oracle_large):
    """decoder in isolation: the emulating decoder is fed the ENGINE's features; first pass (multimask) and m2m pass
    (mask prompt, dynamic single-mask selection); point labels 0 / 1 / -1 mixed"""
    from oracle import sam2_bf16_emul as E
    cfg, W = oracle_large
    engine.encode(torch.from_numpy(image).cuda())
    rng = np.random.default_rng(3)
    pts = torch.tensor(rng.uniform(0, 1024, (8, 2)).astype(np.float32))
    gf = {k: v.cpu()[None] for k, v in engine.get_features(0).items()}
    for lab in (None, torch.tensor([1, 0, 1, -1, 1, 0, 1, 1], dtype=torch.int32)):
        low, iou, obj = engine.decode_points(pts.cuda(), slot=0, multimask=True, labels=None if lab is None else lab.cuda())
        torch.cuda.synchronize()
        r_low, r_iou, r_obj, _, _ = E.mask_decoder_emul(W, gf, pts, lab, True)
        e_low, e_iou, e_obj = rel_rms(low.cpu(), r_low), (iou.cpu() - r_iou).abs().max().item(), (obj.cpu() - r_obj).abs().max().item()
        sign = ((low.cpu() > 0) == (r_low > 0)).float().mean().item()
        print(f"decoder vs emul (labels={'ones' if lab is None else 'mixed'}): low-res rel-rms {e_low:.3e}, iou abs {e_iou:.3e}, obj abs {e_obj:.3e}, sign agreement {sign:.6f}")
        assert e_low < 1e-3 and e_iou < 1e-3 and e_obj < 5e-3 and sign > 0.9995
    low, _, _ = engine.decode_points(pts.cuda(), slot=0, multimask=True)
    mi = torch.clamp(low[:, 0], -32, 32).contiguous()
    low2, iou2, _ = engine.decode_points(pts.cuda(), slot=0, multimask=False, mask_input=mi)
    torch.cuda.synchronize()
    r2, ri2, _, _, _ = E.mask_decoder_emul(W, gf, pts, None, False, mask_in=mi.cpu())
    e2, ei2 = rel_rms(low2.cpu(), r2), (iou2.cpu() - ri2).abs().max().item()
    print(f"m2m decoder vs emul: low-res rel-rms {e2:.3e}, iou abs {ei2:.3e}")
    assert e2 < 1e-3 and ei2 < 1e-3


def test_end_to_end_vs_bf16_emulating_oracle(engine, image, emul_feats, oracle_large):
    """encoder + decoder, nothing shared: emulating encoder features -> emulating decoder vs the engine end to end"""
    from oracle import sam2_bf16_emul as E
    cfg, W = oracle_large
    engine.encode(torch.from_numpy(image).cuda())
    pts = torch.tensor([[256.0, 256.0], [700.0, 300.0], [512.0, 512.0], [100.0, 900.0], [900.0, 120.0], [400.0, 640.0], [50.0, 50.0], [1000.0, 1000.0]])
    low, iou, _ = engine.decode_points(pts.cuda(), slot=0, multimask=True)
    torch.cuda.synchronize()
    r_low, r_iou, _, _, _ = E.mask_decoder_emul(W, emul_feats, pts, None, True)
    e, ei = rel_rms(low.cpu(), r_low), (iou.cpu() - r_iou).abs().max().item()
    # |IoU - 1| of the thresholded masks, per mask
    g, r = low.cpu() > 0, r_low > 0
    inter, uni = (g & r).flatten(2).sum(-1).double(), (g | r).flatten(2).sum(-1).double()
    miou = torch.where(uni > 0, inter / uni, torch.ones_like(uni))
    print(f"end to end vs emul: low-res rel-rms {e:.3e}, iou-head abs {ei:.3e}, mask |IoU-1| median {float((1 - miou).median()):.2e} max {float((1 - miou).max()):.2e}")
    assert e < 2e-3 and ei < 2e-3
    assert float((1 - miou).median()) < 1e-3


def test_amg_masks_vs_bf16_emulating_oracle(engine, image, oracle_large):
    """AMG mask sets: the emulating predictor under the oracle's AMG driver vs the engine's AMG.  Same count, and per matched
    mask |IoU - 1| with median <= 2e-3 (north star: |IoU - 1| < 1e-3)."""
    from oracle import sam2_bf16_emul as E
    from oracle.amg_ref import amg_from_saber_cfg
    from saber_amd.engine import make_amg_params, unpack_bits
    cfg, W = oracle_large
    amg = dict(npoints=6, crop_n_layers=1, box_nms_thresh=1.0, pred_iou_thresh=0.5, stability_score_thresh=0.8)
    ref = amg_from_saber_cfg(E.ImagePredictorEmul(W, cfg), amg).generate(np.repeat(image[..., None], 3, 2))
    bits, meta = engine.amg_generate(torch.from_numpy(image).cuda(), make_amg_params(amg), max_masks=512)
    got = unpack_bits(bits, 1024)
    print(f"AMG vs emul: oracle {len(ref)} masks, engine {len(meta)} masks")
    assert len(ref) >= 10
    assert abs(len(ref) - len(meta)) <= max(1, int(0.05 * len(ref)))
    ious = []
    for r in ref:
        mb = r["segmentation"]
        best = 0.0
        for ma in got:
            uni = np.logical_or(ma, mb).sum()
            best = max(best, np.logical_and(ma, mb).sum() / uni if uni else 1.0)
        ious.append(best)
    dev = 1.0 - np.array(ious)
    print(f"matched |IoU-1|: median {np.median(dev):.2e}, 90th pct {np.quantile(dev, 0.9):.2e}, max {dev.max():.2e}")
    assert np.median(dev) <= 2e-3
    assert np.quantile(dev, 0.9) <= 2e-2
