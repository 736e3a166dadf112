"""GPU: parity that means something.  The engine runs matrix products on bf16 operands with fp32 accumulation;
against the fp32 oracle that costs ~6e-3 relative RMS on the encoder output.  Is that residual the sanctioned operand rounding, or
a kernel defect hiding under a loose bound?  oracle/sam2_bf16_emul.py is an independent torch restatement that rounds where the
engine rounds.  Three facts are asserted:

  1. ERROR BUDGET: the engine is no further from the fp32 oracle than the emulating oracle is (within 10 %): measured on MI355X
     6.0e-3 / 4.2e-3 / 2.8e-3 (engine) against 6.2e-3 / 4.2e-3 / 2.8e-3 (emulation) on image_embed / feat_s1 / feat_s0.  A defect
     adds error on top of the rounding budget; it cannot hide inside it.
  2. REALISATION SPREAD: two bit-different but equally valid bf16 evaluations cannot agree to 1e-3 after more than ~3 roundings in
     series: a difference d between two values that are then rounded to a grid of spacing u becomes 0 with probability 1 - d/u and u
     otherwise, i.e. RMS sqrt(d u) >> d, and the next GEMM spreads it over every output again.  The CPU test
     tests/test_oracle_bf16_emul.py measures this floor by running the emulation twice, with fp32 and with fp64 GEMM accumulation
     (identical roundings): 3.7e-3 / 3.0e-3 / 7e-4.  The engine must sit within 2x of that floor from the emulation.
  3. ONE BLOCK on identical inputs (where the 1e-3 of the north star IS reachable): a windowed and a global MultiScaleBlock
     chained from the kernel-level C-ABI against the emulated block: <= 1e-3 on the block output.
Decoder, end-to-end and AMG mask sets are treated the same way; mask-level |IoU - 1| is asserted directly.
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
def emul_feats(image, oracle_large):
    from oracle import sam2_ref, sam2_bf16_emul as E
    cfg, W = oracle_large
    pix = sam2_ref.sam2_transforms(np.repeat(image[..., None], 3, 2))
    return E.encode_image_emul(W, cfg, pix)


@pytest.fixture(scope="module")
def fp32_feats(oracle_feats):
    return oracle_feats          # (tests/conftest.py: the session's one oracle pass over `image`)


# realisation spread of the emulation itself (fp32- vs fp64-accumulated GEMMs, tests/test_oracle_bf16_emul.py)
SPREAD = {"image_embed": 3.7e-3, "feat_s1": 3.0e-3, "feat_s0": 7.3e-4}


def test_encoder_error_budget_and_realisation_spread(engine, image, emul_feats, fp32_feats):
    engine.encode(torch.from_numpy(image).cuda())
    got = engine.get_features(0)
    torch.cuda.synchronize()
    for k in ("image_embed", "feat_s1", "feat_s0"):
        e_emul = rel_rms(got[k].cpu(), emul_feats[k][0])
        e_fp32 = rel_rms(got[k].cpu(), fp32_feats[k][0])
        b_fp32 = rel_rms(emul_feats[k][0], fp32_feats[k][0])
        print(f"{k}: engine vs emulation {e_emul:.3e} (floor {SPREAD[k]:.1e}); engine vs fp32 {e_fp32:.3e}; emulation vs fp32 {b_fp32:.3e}")
        assert e_fp32 <= 1.10 * b_fp32, (k, e_fp32, b_fp32)          # 1. error budget
        assert e_emul <= 2.0 * SPREAD[k], (k, e_emul)                # 2. within the realisation spread
        assert e_emul < e_fp32                                       # the shared roundings are correlated: closer to the emulation than to fp32


def _bf16_dev(t):
    return t.to(torch.bfloat16).view(torch.uint16).cuda().contiguous()


@pytest.mark.parametrize("block", [20, 23])
def test_one_block_on_identical_inputs(gpu_lib, oracle_large, block):
    """One MultiScaleBlock of Hiera-L stage 2 (block 20: 16 windows of 256 keys; block 23: global attention over 4096 keys) chained from
    the kernel-level C-ABI (LayerNorm -> qkv GEMM -> attention -> proj GEMM + residual -> LayerNorm -> fc1 GEMM + GELU -> fc2 GEMM +
    residual) against oracle.sam2_bf16_emul.hiera_block_emul on the SAME input: here nothing is amplified through depth and the
    north star's 1e-3 holds."""
    import ctypes as C
    from oracle import sam2_bf16_emul as E
    cfg, W = oracle_large
    spec = cfg.block_specs()[block]
    din, dout, heads, win, qs = spec
    assert din == dout == 576 and qs == 1
    N = 4096
    g = torch.Generator().manual_seed(block)
    x = torch.randn(1, N, din, generator=g) * 2.0 + torch.randn(1, 1, din, generator=g)
    ref = E.hiera_block_emul(W, block, spec, x, N, cfg.ln_eps)[0]
    p = f"image_encoder.trunk.blocks.{block}."
    ptr = lambda t: C.c_void_p(t.data_ptr())
    ck = lambda st: (_ for _ in ()).throw(RuntimeError(gpu_lib.saber_k_last_error().decode())) if st != 0 else None
    keep = []                                # device operands must outlive the asynchronous launches that read them

    def f32(name):
        keep.append(W[p + name].cuda().contiguous())
        return keep[-1]

    def wbf(name):
        keep.append(_bf16_dev(W[p + name]))
        return keep[-1]
    xd = x[0].cuda().contiguous()
    xn = torch.empty(N, din, dtype=torch.uint16, device="cuda")
    ck(gpu_lib.saber_k_layernorm(ptr(xd), ptr(f32("norm1.weight")), ptr(f32("norm1.bias")), cfg.ln_eps, None, ptr(xn), N, din, 0, None))
    qkv = torch.empty(N, 3 * dout, dtype=torch.uint16, device="cuda")
    ck(gpu_lib.saber_k_gemm(ptr(xn), ptr(wbf("attn.qkv.weight")), ptr(f32("attn.qkv.bias")), None, None, ptr(qkv), N, 3 * dout, din, 0, 0, 0, 0, 0, None))
    att = torch.empty(N, dout, dtype=torch.uint16, device="cuda")
    nk = win * win if win > 0 else N
    ck(gpu_lib.saber_k_hiera_attention(ptr(qkv), ptr(att), N // nk, nk, heads, 0, None))
    x1 = torch.empty(N, dout, dtype=torch.float32, device="cuda")
    ck(gpu_lib.saber_k_gemm(ptr(att), ptr(wbf("attn.proj.weight")), ptr(f32("attn.proj.bias")), ptr(xd), ptr(x1), None, N, dout, dout, 0, 0, 0, 0, 0, None))
    ck(gpu_lib.saber_k_layernorm(ptr(x1), ptr(f32("norm2.weight")), ptr(f32("norm2.bias")), cfg.ln_eps, None, ptr(xn), N, dout, 0, None))
    hid = torch.empty(N, 4 * dout, dtype=torch.uint16, device="cuda")
    ck(gpu_lib.saber_k_gemm(ptr(xn), ptr(wbf("mlp.layers.0.weight")), ptr(f32("mlp.layers.0.bias")), None, None, ptr(hid), N, 4 * dout, dout, 1, 0, 0, 0, 0, None))
    x2 = torch.empty(N, dout, dtype=torch.float32, device="cuda")
    ck(gpu_lib.saber_k_gemm(ptr(hid), ptr(wbf("mlp.layers.1.weight")), ptr(f32("mlp.layers.1.bias")), ptr(x1), ptr(x2), None, N, dout, 4 * dout, 0, 0, 0, 0, 0, None))
    torch.cuda.synchronize()
    e_out = rel_rms(x2.cpu(), ref)
    e_branch = rel_rms(x2.cpu() - x[0], ref - x[0])                 # the block's own contribution, without the residual it rides on
    print(f"block {block}: output rel-rms {e_out:.3e}; block contribution (output - input) rel-rms {e_branch:.3e}")
    assert e_out < 2e-4 and e_branch < 1e-3                # measured 9.2e-5 / 9.9e-5 and 4.8e-4 / 5.3e-4


# decoder alone (identical features in): 2 transformer layers = ~12 roundings in series on the image-token state; measured engine
# vs emulation, bound = 2x measured
DEC_BOUND = 8e-3


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
        assert e_low < DEC_BOUND and e_iou < 1.2e-2 and e_obj < 3.6e-2 and sign > 0.998          # measured 3.2-3.9e-3 / 3.0-6.0e-3 / 0.9-1.8e-2 / 0.9990
    low, _, _ = engine.decode_points(pts.cuda(), slot=0, multimask=True)
    mi = torch.clamp(low[:, 0], -32, 32).contiguous()
    low2, iou2, _ = engine.decode_points(pts.cuda(), slot=0, multimask=False, mask_input=mi)
    torch.cuda.synchronize()
    r2, ri2, _, _, _ = E.mask_decoder_emul(W, gf, pts, None, False, mask_in=mi.cpu())
    e2, ei2 = rel_rms(low2.cpu(), r2), (iou2.cpu() - ri2).abs().max().item()
    print(f"m2m decoder vs emul: low-res rel-rms {e2:.3e}, iou abs {ei2:.3e}")
    assert e2 < DEC_BOUND and ei2 < 1.2e-2                                   # measured 3.9e-3 / 2.8e-3


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
    assert e < 1e-2 and ei < 8e-3                                            # measured 4.4-4.9e-3 / 3.2-3.7e-3
    assert float((1 - miou).median()) < 2.8e-3 and float((1 - miou).max()) < 1e-2   # measured 1.4e-3 / 4.9e-3


def test_amg_masks_vs_bf16_emulating_oracle(engine, image):
    """AMG mask sets: the emulating predictor under the oracle's AMG driver vs the engine's AMG.  Same count, and |IoU - 1| per
    matched mask within 2x of what was measured (the north star's |IoU - 1| < 1e-3 is met per decoder pass on identical inputs, not
    after encoder + two decoder passes of independent bf16 evaluations: see the module docstring, REALISATION SPREAD)."""
    from conftest import amg_case
    from saber_amd.engine import make_amg_params, unpack_bits
    # one crop layer (with more, the seeded model's image-sized masks fall to the cross-crop NMS and a single mask is left to compare);
    # box NMS off so that every mask that passes the score filters is compared
    amg = dict(npoints=6, crop_n_layers=0, box_nms_thresh=1.0, pred_iou_thresh=0.5, stability_score_thresh=0.8)
    ref = amg_case("emul_l0")          # the emulating predictor under the oracle's AMG driver: oracle/make_golden_amg_cases.py (60 s of host time)
    bits, meta = engine.amg_generate(torch.from_numpy(image).cuda(), make_amg_params(amg), max_masks=512)
    got = unpack_bits(bits, 1024)
    print(f"AMG vs emul: oracle {len(ref)} masks, engine {len(meta)} masks")
    assert len(ref) >= 10
    assert abs(len(ref) - len(meta)) <= max(1, int(0.05 * len(ref)))
    ious = []
    for r in ref:
        mb = r["segmentation"]
        best = 0.0
        for ma in got[:, 2::4, 2::4]:                 # quarter-resolution samples of both sides
            uni = np.logical_or(ma, mb).sum()
            best = max(best, np.logical_and(ma, mb).sum() / uni if uni else 1.0)
        ious.append(best)
    dev = 1.0 - np.array(ious)
    print(f"matched |IoU-1|: median {np.median(dev):.2e}, 90th pct {np.quantile(dev, 0.9):.2e}, max {dev.max():.2e}")
    # two decoder passes (grid prompt, then m2m on its logits) sit between the features and these masks; measured median 3.2e-3,
    # 90th percentile 4.4e-3, max 7.0e-3 (against the fp32 oracle: median 4.7e-3)
    assert np.median(dev) <= 6.4e-3 and np.quantile(dev, 0.9) <= 9e-3 and dev.max() <= 1.4e-2
