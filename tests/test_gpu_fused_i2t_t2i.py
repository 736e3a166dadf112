"""GPU: the opt-in fused kernel `dec_i2t_t2i_kernel` (image -> tokens of a decoder layer + the tokens -> image attention that follows it,
SABER_AMD_FUSE_I2T_T2I=1) against the two separate launches it replaces (sam2 two_way_transformer: TwoWayAttentionBlock.forward's
cross_attn_image_to_token followed by the next block's / the final cross_attn_token_to_image).

X' (the updated image tokens) is computed by the same instructions in both forms and must be bit-identical - visible in the mask logits of
prompts whose tokens did not change; the attention over X' differs only in the grouping of the online softmax (32-key blocks in one wave
instead of 64-key blocks in two), so the decode's outputs agree to the operand format's rounding."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_fused_equals_separate_launches(precision):
    from saber_amd.engine import Engine
    eng = Engine("large", device=0, seed=0, max_images=1, max_prompts=1024, precision=precision)
    try:
        g = torch.Generator(device="cpu").manual_seed(3)
        eng.encode(torch.rand(1024, 1024, generator=g).cuda())
        pts = (torch.rand(1024, 2, generator=g) * 1024).cuda()
        outs = {}
        for fuse in (False, True):
            if fuse:
                os.environ["SABER_AMD_FUSE_I2T_T2I"] = "1"
            else:
                os.environ.pop("SABER_AMD_FUSE_I2T_T2I", None)
            low, iou, _ = eng.decode_points(pts, slot=0, multimask=True)
            mi = torch.clamp(low[:, 0], -32, 32).contiguous()
            low2, iou2, _ = eng.decode_points(pts, slot=0, multimask=False, mask_input=mi)
            torch.cuda.synchronize()
            outs[fuse] = [t.float().cpu().numpy() for t in (low, iou, low2, iou2)]
        eng.check_finite()
        tol = {"bf16": 2e-2, "fp16": 3e-3}[precision]
        for name, a, b in zip(("logits", "pred_iou", "m2m logits", "m2m pred_iou"), outs[False], outs[True]):
            scale = max(1.0, float(np.abs(a).max()))
            err = float(np.abs(a - b).max()) / scale
            print(f"{precision} fused vs separate, {name}: max |diff| / max|ref| = {err:.2e} (scale {scale:.1f})")
            assert np.isfinite(b).all()
            assert err < tol, name
            if "logits" in name:
                assert ((a > 0) != (b > 0)).mean() < 2e-3
    finally:
        os.environ.pop("SABER_AMD_FUSE_I2T_T2I", None)
        eng.close()


@pytest.mark.parametrize("shared", [False, True])
def test_t2i_one_wave_per_simd_kernel_against_fp64(shared):
    """`dec_t2i_w1_kernel` (the route for whole-key-range launches; debug flag 0x20000000 / SABER_AMD_T2I_W1=0 select the 8-wave kernel): the tokens -> image attention with a wave per key quarter and all
    four query tiles per wave - against the same fp64 restatement tests/test_gpu_kernels.py::test_dec_t2i uses, and run-to-run identical
    (it orders its LDS-DMA ring by counted waits alone)."""
    import ctypes as C
    from saber_amd import _lib
    from tests.test_gpu_kernels import _dec_inputs, _blockdiag_pe_scores, kcall, ptr
    lib = _lib.load()
    assert lib.saber_k_init(0) == 0
    P = 300
    g, r, X, pe = _dec_inputs(1 if shared else P, 11)
    Qt = r(P, 64, 256, scale=0.05).to(torch.bfloat16)
    pek = r(4096, 128, scale=1.0).to(torch.bfloat16)
    tq = r(P * 8, 128, scale=1.0)
    qscale = 0.3
    Wv = (r(128, 256) / 16).to(torch.bfloat16)
    bv = r(128)
    part = torch.zeros(P * 64 * 256, device="cuda")
    ml = torch.zeros(P * 64 * 2, device="cuda")
    outs = []
    lib.saber_k_set_debug(0x10000000)
    try:
        for _ in range(3):
            out = torch.zeros(P, 8, 128, device="cuda", dtype=torch.bfloat16)
            kcall(lib, lib.saber_k_dec_t2i(ptr(X), 0 if shared else 4096 * 256, ptr(pek), ptr(Qt), ptr(tq), qscale, ptr(part), ptr(ml), P, 1, ptr(Wv), ptr(bv), ptr(out), None))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        lib.saber_k_set_debug(0)
    ref8 = torch.zeros(P, 8, 128, device="cuda", dtype=torch.bfloat16)
    lib.saber_k_set_debug(0x20000000)
    kcall(lib, lib.saber_k_dec_t2i(ptr(X), 0 if shared else 4096 * 256, ptr(pek), ptr(Qt), ptr(tq), qscale, ptr(part), ptr(ml), P, 1, ptr(Wv), ptr(bv), ptr(ref8), None))
    torch.cuda.synchronize()
    lib.saber_k_set_debug(0)
    Xd = X.double().expand(P, -1, -1)
    tqb = (tq * qscale).to(torch.bfloat16).view(P, 8, 128)
    S = Qt.double() @ Xd.transpose(1, 2) + _blockdiag_pe_scores(tqb, pek, 1.0)
    Z = (torch.softmax(S * np.log(2.0), dim=-1) @ Xd).view(P, 8, 8, 256)
    ref = torch.einsum("phtd,hid->pthi", Z, Wv.double().view(8, 16, 256)).reshape(P, 8, 128) + bv.double()
    scale = ref.abs().max().item()
    err = (outs[0].double() - ref).abs().max().item()
    err8 = (ref8.double() - ref).abs().max().item()
    print(f"dec_t2i_w1 (shared={shared}): max err {err:.3e} of scale {scale:.3f}; the 8-wave kernel: {err8:.3e}")
    assert err < 0.02 * scale
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_decode_with_t2i_w1_equals_default():
    from saber_amd.engine import Engine
    eng = Engine("large", device=0, seed=0, max_images=1, max_prompts=1024, precision="fp16")
    try:
        g = torch.Generator(device="cpu").manual_seed(4)
        eng.encode(torch.rand(1024, 1024, generator=g).cuda())
        pts = (torch.rand(1024, 2, generator=g) * 1024).cuda()
        outs = {}
        for w1 in (False, True):
            if w1:
                os.environ.pop("SABER_AMD_T2I_W1", None)       # the default route
            else:
                os.environ["SABER_AMD_T2I_W1"] = "0"
            low, iou, _ = eng.decode_points(pts, slot=0, multimask=True)
            mi = torch.clamp(low[:, 0], -32, 32).contiguous()
            low2, iou2, _ = eng.decode_points(pts, slot=0, multimask=False, mask_input=mi)
            torch.cuda.synchronize()
            outs[w1] = [t.float().cpu().numpy() for t in (low, iou, low2, iou2)]
        eng.check_finite()
        for name, a, b in zip(("logits", "pred_iou", "m2m logits", "m2m pred_iou"), outs[False], outs[True]):
            scale = max(1.0, float(np.abs(a).max()))
            err = float(np.abs(a - b).max()) / scale
            print(f"fp16, t2i one-wave-per-SIMD vs default, {name}: max |diff| / max|ref| = {err:.2e}")
            assert np.isfinite(b).all() and err < 3e-3, name
    finally:
        os.environ.pop("SABER_AMD_T2I_W1", None)
        eng.close()


@pytest.mark.parametrize("shared", [False, True])
def test_i2t_one_wave_per_simd_kernel(shared):
    """`dec_i2t_w1_kernel` (the route for whole-prompt launches: P >= 512; debug flag 0x40000000 / SABER_AMD_I2T_W1=0 select the four-wave
    kernel): a wave owns a 16-row tile - all 64 score columns, all 256 channels, softmax weights and LayerNorm statistics in registers.
    Against the fp64 formula of tests/test_gpu_kernels.py::test_dec_i2t on the first prompts, against the four-wave kernel on all of
    them (same operands, same roundings: one 16-bit ulp at most apart), in place (X = Xout: layer 1 of the decoder), run-to-run identical."""
    import torch.nn.functional as F
    from saber_amd import _lib
    from tests.test_gpu_kernels import _dec_inputs, _blockdiag_pe_scores, kcall, ptr
    lib = _lib.load()
    assert lib.saber_k_init(0) == 0
    P = 520
    g, r, X, pe = _dec_inputs(1 if shared else P, 23)
    Kt = r(P, 64, 256, scale=0.08).to(torch.bfloat16)
    peq = r(4096, 128, scale=1.0).to(torch.bfloat16)
    tk = r(P * 8, 128, scale=1.0)
    kscale = 0.3
    cb = r(P, 64)
    VtT = r(P, 256, 64, scale=0.5).to(torch.bfloat16)
    bo, gamma, beta = r(256), 1.0 + 0.1 * r(256), 0.1 * r(256)

    def run(flag, inplace=False):
        lib.saber_k_set_debug(flag)
        try:
            if inplace:
                out = X.clone()
                src = out
            else:
                out = torch.zeros(P, 4096, 256, device="cuda", dtype=torch.bfloat16)
                src = X
            kcall(lib, lib.saber_k_dec_i2t(ptr(src), 0 if shared else 4096 * 256, ptr(peq), ptr(Kt), ptr(tk), kscale, ptr(cb), ptr(VtT), ptr(bo), ptr(gamma),
                                           ptr(beta), 1e-5, ptr(out), P, None))
            torch.cuda.synchronize()
            return out
        finally:
            lib.saber_k_set_debug(0)

    w1 = run(0)
    ref4 = run(0x40000000)
    assert torch.equal(w1, run(0)), "run-to-run"
    d = (w1.float() - ref4.float()).abs()
    print(f"dec_i2t_w1 vs the four-wave kernel (shared={shared}): max |diff| {d.max().item():.3e}, differing elements {(d > 0).float().mean().item():.2e}")
    assert d.max().item() <= 0.04 and (d > 0).float().mean().item() < 0.05
    if not shared:
        assert torch.equal(w1, run(0, inplace=True)), "in place"
    n = 3
    Xd = (X.double().expand(P, -1, -1))[:n]
    tkb = (tk * kscale).to(torch.bfloat16).view(P, 8, 128)[:n]
    S = Xd @ Kt[:n].double().transpose(1, 2) + _blockdiag_pe_scores(tkb, peq, 1.0).transpose(1, 2) + cb[:n].double()[:, None, :]
    Pm = torch.softmax(S.view(n, 4096, 8, 8) * np.log(2.0), dim=-1).view(n, 4096, 64)
    Y = Pm.to(torch.bfloat16).double() @ VtT[:n].double().transpose(1, 2)
    ref = F.layer_norm(Xd + Y + bo.double(), (256,), gamma.double(), beta.double(), 1e-5)
    err = (w1[:n].double() - ref).abs().max().item()
    rms = (w1[:n].double() - ref).pow(2).mean().sqrt().item()
    print(f"dec_i2t_w1 vs fp64: max {err:.3e}, rms {rms:.3e}")
    assert err < 0.04 and rms < 4e-3
