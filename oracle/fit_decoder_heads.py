"""Test infrastructure (never imported by the product): fits the MASK DECODER of the seeded Hiera-L model so that the automatic mask
generator's filters mean something on the synthetic slices.

Why.  With seeded (untrained) weights every candidate mask is an image-sized blob, every box is its crop's box and all stability scores
lie in [0.90, 0.92]: cfgAMG's own thresholds (saber/adapters/sam2/amg.py:7-17: pred_iou 0.7, stability 0.92, box NMS 0.7) leave 0-1 masks,
so the parity goldens ran with NMS off and hand-picked thresholds, and paint order / dedup / area sort (saber/segmenters/base.py:159-176,
propagation.py:185-188) never saw more than a few masks per plane.  No checkpoint is available offline, so this script makes the one thing
a checkpoint would provide - a decoder whose outputs are compact objects with a spread of predicted IoU and stability - out of the
synthetic data itself: the image encoder stays the seeded one (frozen; features are computed once per crop), only `sam_mask_decoder.*`
and the prompt encoder's mask-input branch (about 4.1 M parameters) are fitted for a few hundred Adam steps on the synthetic slices'
own blob geometry (oracle.saber_ref.synthetic_slice draws soft-edged ellipses: their parameters are re-derived here from the same
generator state).

Targets per point prompt (all three crop scales of the default pyramid, so that crops see the objects 1.7x / 2.8x larger):
  mask token 1 = the smallest ellipse containing the click, token 2 = the same ellipse at 1.3x its radius, token 3 = at 0.7x,
  token 0 (single-mask output, used by the m2m refinement) = the ellipse; clicks on the background: all empty;
  m2m refinement (mask_input = a first-pass logit plane, multimask_output=False) = the target of the plane it was given;
  IoU head = the IoU each predicted mask actually has with its target (detached), which is what gives pred_iou_thresh something to cut.

The result is stored as tests/golden/decoder_fit_large_seed0.npz: the fitted tensors in float16 (the delta against the seeded model is the
whole tensor: they replace the seeded ones), < 10 MB.  saber_amd.weights.fitted_decoder_weights() overlays them on seeded_weights(cfg, 0).
Runs on the GPU box with torch (about 10 minutes: `python -m oracle.fit_decoder_heads`), or on CPU (hours)."""
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
OUT = os.environ.get("FIT_OUT", os.path.join(ROOT, "tests", "golden", "decoder_fit_large_seed0.npz"))


def blob_params(seed, size=1024, n_blobs=24):
    """The ellipses oracle.saber_ref.synthetic_slice(seed) draws: the same generator calls in the same order."""
    rng = np.random.default_rng(seed)
    rng.normal(32768.0, 3000.0, (size, size))
    out = []
    for _ in range(n_blobs):
        cy, cx = rng.uniform(0, size, 2)
        r = rng.uniform(20, 120)
        ax = rng.uniform(0.6, 1.0)
        amp = rng.uniform(2000, 8000) * rng.choice([-1.0, 1.0])
        out.append((cy, cx, r, ax, amp))
    return out


def ellipse_d(blob, yy, xx):
    cy, cx, r, ax, _ = blob
    return torch.sqrt(((yy - cy) / r) ** 2 + ((xx - cx) / (r * ax)) ** 2)


def main():
    from oracle import amg_ref, saber_ref, sam2_ref
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    steps = int(os.environ.get("STEPS", "600"))
    n_slices = int(os.environ.get("SLICES", "3"))
    P = int(os.environ.get("PROMPTS", "48"))
    torch.manual_seed(0)
    cfg = get_config("large")
    Wnp = seeded_weights(cfg, 0)
    torch.set_default_device(dev)
    W = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in Wnp.items()}
    train_keys = [k for k in W if k.startswith("sam_mask_decoder.") and not k.startswith(("sam_mask_decoder.conv_s0", "sam_mask_decoder.conv_s1"))]
    train_keys += [k for k in W if k.startswith("sam_prompt_encoder.mask_downscaling.")]
    if os.environ.get("FIT_INIT"):                      # continue from an earlier result (a GPU-box call is limited to 20 minutes)
        with np.load(os.environ["FIT_INIT"]) as Z:
            for k in train_keys:
                W[k] = torch.from_numpy(np.ascontiguousarray(Z[k], dtype=np.float32)).to(dev)
        print("continuing from", os.environ["FIT_INIT"], flush=True)
    n_par = sum(W[k].numel() for k in train_keys)
    print(f"device {dev}; fitting {len(train_keys)} tensors, {n_par / 1e6:.2f} M parameters; {steps} steps of {P} prompts", flush=True)

    # ---- features of every crop of the default pyramid, frozen encoder
    crops = []        # (feats, crop box xyxy, blobs, seed)
    t0 = time.time()
    with torch.no_grad():
        for s in range(n_slices):
            img = saber_ref.prepare(saber_ref.synthetic_slice(seed=s).astype(np.float32), to_rgb=True)
            boxes, _ = amg_ref.generate_crop_boxes(img.shape[:2], 2, 512 / 1500)
            blobs = blob_params(s)
            for (x0, y0, x1, y1) in boxes[:int(os.environ.get("CROPS_MAX", "99"))]:
                with torch.device("cpu"):
                    px = sam2_ref.sam2_transforms(img[y0:y1, x0:x1], 1024)
                px = px.to(dev)
                f = sam2_ref.encode_image(W, cfg, px)
                crops.append(({k: v.detach() for k, v in f.items()}, (x0, y0, x1, y1), blobs, s))
            print(f"slice {s}: {len(boxes)} crops encoded ({time.time() - t0:.0f} s)", flush=True)
    pos = sam2_ref.dense_pe(W, 64)
    params = [W[k].requires_grad_(True) for k in train_keys]
    opt = torch.optim.Adam(params, lr=float(os.environ.get("LR", "3e-4")))
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=float(os.environ.get("LR", "3e-4")), total_steps=steps, pct_start=0.1)
    g = torch.Generator(device=dev).manual_seed(1)
    lin = (torch.arange(256, dtype=torch.float32) + 0.5) / 256.0

    def targets_for(crop, pts_crop):
        """pts_crop (P,2) crop pixels (x, y) -> (P,4,256,256) {0,1} targets on the crop's low-res grid, (P,) is-object"""
        _, (x0, y0, x1, y1), blobs, _ = crop
        yy = (y0 + lin * (y1 - y0))[:, None].expand(256, 256)
        xx = (x0 + lin * (x1 - x0))[None, :].expand(256, 256)
        px, py = pts_crop[:, 0] + x0, pts_crop[:, 1] + y0
        tgt = torch.zeros(pts_crop.shape[0], 4, 256, 256)
        best_r = torch.full((pts_crop.shape[0],), 1e9)
        for b in blobs:
            cy, cx, r, ax, _ = b
            dpt = torch.sqrt(((py - cy) / r) ** 2 + ((px - cx) / (r * ax)) ** 2)
            hit = (dpt < 0.9) & (r < best_r)             # the smallest ellipse containing the click (clicks near an edge count as background)
            if not bool(hit.any()):
                continue
            d = ellipse_d(b, yy, xx)
            m = torch.stack([(d < 1.0), (d < 1.0), (d < 1.3), (d < 0.7)]).float()
            tgt[hit] = m
            best_r = torch.where(hit, torch.full_like(best_r, r), best_r)
        return tgt, best_r < 1e8

    def sample_points(crop, n):
        """half of the clicks inside ellipses visible in the crop, half uniform"""
        _, (x0, y0, x1, y1), blobs, _ = crop
        w, h = x1 - x0, y1 - y0
        pts = torch.rand(n, 2, generator=g) * torch.tensor([w, h], dtype=torch.float32)
        vis = [b for b in blobs if x0 - 0.5 * b[2] < b[1] < x1 + 0.5 * b[2] and y0 - 0.5 * b[2] < b[0] < y1 + 0.5 * b[2]]
        if vis:
            for i in range(n // 2):
                cy, cx, r, ax, _ = vis[int(torch.randint(len(vis), (1,), generator=g))]
                a, rad = float(torch.rand(1, generator=g)) * 2 * math.pi, math.sqrt(float(torch.rand(1, generator=g))) * 0.8
                pts[i, 0] = min(max(cx + rad * r * ax * math.cos(a) - x0, 0.0), w - 1.0)
                pts[i, 1] = min(max(cy + rad * r * math.sin(a) - y0, 0.0), h - 1.0)
        return pts

    def mask_loss(logits, tgt):
        bce = F.binary_cross_entropy_with_logits(logits, tgt, reduction="none").flatten(2).mean(-1)
        p = torch.sigmoid(logits).flatten(2)
        t = tgt.flatten(2)
        dice = 1 - (2 * (p * t).sum(-1) + 1) / (p.sum(-1) + t.sum(-1) + 1)
        return 5.0 * bce + dice

    def iou_of(logits, tgt):
        m = (logits > 0).flatten(2)
        t = (tgt > 0.5).flatten(2)
        inter, union = (m & t).sum(-1).float(), (m | t).sum(-1).float()
        return torch.where(union > 0, inter / union.clamp(min=1), torch.zeros_like(union))      # empty target + empty prediction: IoU 0 (nothing to keep)

    t0 = time.time()
    for it in range(steps):
        crop = crops[int(torch.randint(len(crops), (1,), generator=g))]
        feats, (x0, y0, x1, y1), _, _ = crop
        pc = sample_points(crop, P)
        tgt, _ = targets_for(crop, pc)
        pm = pc * torch.tensor([1024.0 / (x1 - x0), 1024.0 / (y1 - y0)])
        lab = torch.ones(P, 1, dtype=torch.int64)
        sparse, dense = sam2_ref.prompt_encoder(W, pm[:, None], lab, None, 1024)
        _, _, _, allm, alli = sam2_ref.mask_decoder(W, feats, sparse, dense, True, pos)
        loss = mask_loss(allm, tgt).mean() + F.mse_loss(alli, iou_of(allm.detach(), tgt))
        # m2m refinement of one of the three planes, as the generator does it (mask_input clamped to +-32, single-mask output)
        k = 1 + int(torch.randint(3, (1,), generator=g))
        mi = torch.clamp(allm[:, k:k + 1].detach(), -32.0, 32.0)
        sparse2, dense2 = sam2_ref.prompt_encoder(W, pm[:, None], lab, mi, 1024)
        _, _, _, allm2, alli2 = sam2_ref.mask_decoder(W, feats, sparse2, dense2, False, pos)
        t2 = tgt[:, k:k + 1].expand(-1, 4, -1, -1)           # every token refines towards the plane it was given (token 0 is the one returned)
        loss = loss + mask_loss(allm2, t2).mean() + F.mse_loss(alli2, iou_of(allm2.detach(), t2))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        sched.step()
        if it % 25 == 0 or it == steps - 1:
            with torch.no_grad():
                i1 = iou_of(allm, tgt)
                obj = tgt[:, 1].flatten(1).sum(-1) > 0
                print(f"step {it:4d} loss {loss.item():.4f}  IoU(token 1 | object clicks) {i1[obj, 1].mean().item() if bool(obj.any()) else float('nan'):.3f}  "
                      f"pred-IoU mean {alli[:, 1].mean().item():.3f}  m2m IoU {iou_of(allm2, t2)[obj, 0].mean().item() if bool(obj.any()) else float('nan'):.3f}  ({time.time() - t0:.0f} s)", flush=True)
    out = {k: W[k].detach().float().cpu().numpy() for k in train_keys}
    big = max(float(np.abs(v).max()) for v in out.values())
    assert big < 6.0e4, f"a fitted tensor leaves the fp16 range ({big})"
    np.savez_compressed(OUT, **{k: v.astype(np.float16) for k, v in out.items()},
                        __meta__=np.array(f"oracle/fit_decoder_heads.py: {steps} Adam steps x {P} prompts on synthetic_slice(seed 0..{n_slices - 1}), seeded Hiera-L encoder frozen"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes", flush=True)


if __name__ == "__main__":
    main()
