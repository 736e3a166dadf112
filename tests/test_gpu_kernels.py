"""GPU parity tests of the individual gfx950 kernels, called through the kernel-level C-ABI
(include/saber_amd_kernels.h).  The reference for each primitive is the plain fp32 formula on the CPU
(the same formulas oracle/sam2_ref.py is built from), evaluated on the bf16-rounded operands the kernel sees.

Tolerances: bf16 operands with fp32 accumulation reproduce an fp32 evaluation of the SAME rounded operands
to ~1e-5 relative (summation order only); kernels that round an intermediate to bf16 (attention P, bf16
outputs) are allowed one bf16 ulp = 2^-8 relative to the row scale.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def bf(x):  # fp32 cpu tensor -> (bf16-rounded fp32 cpu, device uint16 storage)
    b = x.to(torch.bfloat16)
    return b.float(), b.view(torch.int16).cuda()


def from_bf(t_i16):
    return t_i16.cpu().view(torch.bfloat16).float()


def kcall(lib, st):
    assert st == 0, lib.saber_k_last_error().decode()
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,act,use_res,pool4", [
    (300, 432, 144, 0, False, 0), (4096, 576, 2304, 0, True, 0), (1000, 2304, 576, 1, False, 0),
    (5, 1, 256, 0, False, 0), (64, 4, 256, 3, False, 0), (512, 288, 144, 0, False, 1), (777, 128, 64, 2, True, 0),
    (131072, 288, 128, 0, False, 1), (65536 + 8, 200, 64, 0, True, 1),   # q-pool shortcut through the two direct-to-LDS kernels
])
def test_gemm(gpu_lib, M, N, K, act, use_res, pool4):
    g = torch.Generator().manual_seed(M * 7 + N)
    A, Ad = bf(torch.randn(M, K, generator=g))
    W, Wd = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    Mo = M // 4 if pool4 else M
    res = torch.randn(Mo, N, generator=g) if use_res else None
    ref = (A.double() @ W.double().T + bias.double())
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = F.relu(ref)
    elif act == 3:
        ref = torch.sigmoid(ref)
    if pool4:
        ref = ref.view(Mo, 4, N).max(1).values
    if use_res:
        ref = ref + res.double()
    out_f = torch.zeros(Mo, N, dtype=torch.float32, device="cuda")
    out_b = torch.zeros(Mo, N, dtype=torch.int16, device="cuda")
    bias_d, res_d = bias.cuda(), (res.cuda() if use_res else None)  # keep device operands alive across the call
    kcall(gpu_lib, gpu_lib.saber_k_gemm(ptr(Ad), ptr(Wd), ptr(bias_d), ptr(res_d), ptr(out_f), ptr(out_b),
                                        M, N, K, act, 0, pool4, 0, 0, None))
    scale = ref.abs().max().item() + 1e-6
    err = (out_f.cpu().double() - ref).abs().max().item() / scale
    assert err < 2e-5, err
    errb = (from_bf(out_b).double() - ref).abs().max().item() / scale
    assert errb < 5e-3, errb


@pytest.mark.parametrize("M,N,K,act", [(777, 1000, 192, 0), (4096, 2304, 576, 1), (300, 264, 64, 0), (2560, 1728, 576, 0), (256 * 9 + 5, 512, 1152, 1)])
def test_gemm_p256(gpu_lib, M, N, K, act):
    """persistent 256x256-tile kernel (bf16 output): forced through the debug flag for small shapes, ragged M / N, several tiles per block"""
    g = torch.Generator().manual_seed(M + N + K)
    A, Ad = bf(torch.randn(M, K, generator=g))
    W, Wd = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + bias.double()
    ref = F.gelu(ref) if act == 1 else F.relu(ref) if act == 2 else ref
    out_b = torch.zeros(M, N, dtype=torch.int16, device="cuda")
    bias_d = bias.cuda()
    gpu_lib.saber_k_set_debug(128)
    try:
        kcall(gpu_lib, gpu_lib.saber_k_gemm(ptr(Ad), ptr(Wd), ptr(bias_d), None, None, ptr(out_b), M, N, K, act, 0, 0, 0, 0, None))
    finally:
        gpu_lib.saber_k_set_debug(0)
    scale = ref.abs().max().item() + 1e-6
    errb = (from_bf(out_b).double() - ref).abs().max().item() / scale
    assert errb < 5e-3, errb


def test_gemm_act_last_and_res_mod(gpu_lib):
    g = torch.Generator().manual_seed(3)
    M, N, K = 640, 128, 64
    A, Ad = bf(torch.randn(M, K, generator=g))
    W, Wd = bf(torch.randn(N, K, generator=g) / 8)
    bias = torch.randn(N, generator=g)
    res = torch.randn(160, N, generator=g)
    ref = F.gelu(A.double() @ W.double().T + bias.double() + res.double().repeat(4, 1))
    out_f = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    bias_d, res_d = bias.cuda(), res.cuda()
    kcall(gpu_lib, gpu_lib.saber_k_gemm(ptr(Ad), ptr(Wd), ptr(bias_d), ptr(res_d), ptr(out_f), None, M, N, K, 1, 1, 0, 0, 160, None))
    assert (out_f.cpu().double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("rows,C,act", [(1000, 144, 0), (333, 576, 0), (64, 1152, 0), (4096, 64, 1), (17, 256, 0), (5001, 144, 0), (4099, 192, 1), (70003, 144, 0)])
def test_layernorm(gpu_lib, rows, C, act):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(rows, C, generator=g) * 3 + 1
    gam, bet = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-6)
    if act:
        ref = F.gelu(ref)
    of = torch.zeros(rows, C, device="cuda")
    ob = torch.zeros(rows, C, dtype=torch.int16, device="cuda")
    xd, gd, bd = x.cuda(), gam.cuda(), bet.cuda()
    kcall(gpu_lib, gpu_lib.saber_k_layernorm(ptr(xd), ptr(gd), ptr(bd), 1e-6, ptr(of), ptr(ob), rows, C, act, None))
    # act=1: the kernels' GELU is the 9-op fit of the exact erf form (common.h gelu_erf, |error| <= 2.6e-5 absolute)
    assert (of.cpu().double() - ref).abs().max().item() < (5e-5 if act else 2e-5)
    assert (from_bf(ob).double() - ref).abs().max().item() < 0.03 * ref.abs().max().item()


def ref_hiera_attention(qkv, n_windows, nk, heads, q_pool, hd=72, key_mask=None):
    t = qkv.view(n_windows, nk, 3, heads, hd).double()
    q, k, v = t[:, :, 0], t[:, :, 1], t[:, :, 2]
    if q_pool:
        q = q.view(n_windows, nk // 4, 4, heads, hd).max(2).values
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    sc = q @ k.transpose(-1, -2) * hd ** -0.5
    if key_mask is not None:
        sc = sc.masked_fill(~key_mask[None, None, None, :], float("-inf"))
    a = torch.softmax(sc, -1) @ v
    return a.transpose(1, 2).reshape(-1, heads * hd)


@pytest.mark.parametrize("n_windows,nk,heads,q_pool", [
    (9, 64, 2, 0), (5, 64, 4, 1), (33, 16, 4, 0), (7, 16, 8, 1), (3, 256, 8, 0), (2, 256, 16, 1), (1, 4096, 8, 0), (2, 128, 2, 0),
    (3, 512, 4, 0), (11, 1024, 2, 0), (21, 256, 8, 0), (2, 4096, 3, 0),   # streaming kernel: several tasks per block, chunked queries
])
def test_hiera_attention(gpu_lib, n_windows, nk, heads, q_pool):
    g = torch.Generator().manual_seed(nk + heads)
    qkv, qd = bf(torch.randn(n_windows * nk, 3 * heads * 72, generator=g) * 1.5)
    ref = ref_hiera_attention(qkv, n_windows, nk, heads, q_pool)
    out = torch.zeros(ref.shape, dtype=torch.int16, device="cuda")
    kcall(gpu_lib, gpu_lib.saber_k_hiera_attention(ptr(qd), ptr(out), n_windows, nk, heads, q_pool, None))
    got = from_bf(out).double()
    err = (got - ref).abs().max().item()
    assert err < 0.03, err  # |v| ~ 1.5: bf16 P and bf16 output rounding
    assert ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item() < 6e-3


@pytest.mark.parametrize("n_windows,nk,heads,hd,q_pool,masked", [
    # tiny/small (head_dim 96) and base+ (56): 8x8 / 4x4 windows, the 14x14 and 7x7 padded windows, global blocks whose token
    # matrix carries the window-padding rows (4900 rows, 4096 of them real keys)
    (9, 64, 1, 96, 0, 0), (5, 64, 2, 96, 1, 0), (33, 16, 2, 96, 0, 0), (7, 16, 4, 56, 1, 0), (26, 196, 4, 96, 0, 0), (11, 196, 8, 96, 1, 0),
    (25, 49, 8, 96, 0, 0), (25, 49, 16, 56, 0, 0), (9, 196, 8, 56, 0, 0), (3, 196, 16, 56, 1, 0), (1, 4900, 4, 96, 0, 1), (2, 4900, 8, 56, 0, 1),
    (3, 196, 8, 72, 0, 0), (2, 300, 2, 72, 0, 1),
])
def test_hiera_attention_all_trunks(gpu_lib, n_windows, nk, heads, hd, q_pool, masked):
    g = torch.Generator().manual_seed(nk + heads + hd)
    qkv, qd = bf(torch.randn(n_windows * nk, 3 * heads * hd, generator=g) * 1.5)
    km = kd = None
    if masked:
        km = torch.rand(nk, generator=g) > 0.2
        km[:8] = True
        pad = torch.zeros((nk + 127) // 128 * 128, dtype=torch.uint8)
        pad[:nk] = km.to(torch.uint8)
        kd = pad.cuda()
    ref = ref_hiera_attention(qkv, n_windows, nk, heads, q_pool, hd, km)
    out = torch.zeros(ref.shape, dtype=torch.int16, device="cuda")
    kcall(gpu_lib, gpu_lib.saber_k_hiera_attention_ex(ptr(qd), ptr(out), n_windows, nk, heads, hd, q_pool, ptr(kd) if masked else None, None))
    got = from_bf(out).double()
    err = (got - ref).abs().max().item()
    assert err < 0.03, err
    assert ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item() < 6e-3


@pytest.mark.parametrize("B,nq,nk,heads,hd,shared", [(3, 4096, 8, 8, 16, 0), (5, 8, 8, 8, 32, 0), (4, 8, 4096, 8, 16, 0), (4, 8, 4096, 8, 16, 1)])
def test_dec_attention(gpu_lib, B, nq, nk, heads, hd, shared):
    g = torch.Generator().manual_seed(nq + nk)
    C_ = heads * hd
    q = torch.randn(B, nq, C_, generator=g)
    k = torch.randn(1 if shared else B, nk, C_, generator=g)
    v = torch.randn(1 if shared else B, nk, C_, generator=g)
    qq = q.view(B, nq, heads, hd).transpose(1, 2).double()
    kk = k.expand(B, -1, -1).reshape(B, nk, heads, hd).transpose(1, 2).double()
    vv = v.expand(B, -1, -1).reshape(B, nk, heads, hd).transpose(1, 2).double()
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) / hd ** 0.5, -1) @ vv).transpose(1, 2).reshape(B, nq, C_)
    out = torch.zeros(B, nq, C_, dtype=torch.int16, device="cuda")
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    kcall(gpu_lib, gpu_lib.saber_k_dec_attention(ptr(qd), ptr(kd), ptr(vd), ptr(out), B, nq, nk, heads, hd, shared, None))
    assert (from_bf(out).double() - ref).abs().max().item() < 0.02


@pytest.mark.parametrize("crop", [(0, 0, 1024, 1024), (100, 50, 597, 400), (700, 724, 300, 300), (10, 20, 200, 100)])
def test_mask_post(gpu_lib, crop):
    x0, y0, cw, ch = crop
    H = W = 1024
    g = torch.Generator().manual_seed(cw)
    n = 5
    low = F.interpolate(torch.randn(n, 1, 16, 16, generator=g) * 4, size=(256, 256), mode="bicubic")[:, 0].contiguous()
    low[4] = -5.0  # empty mask
    full = F.interpolate(low[:, None], size=(ch, cw), mode="bilinear", align_corners=False)[:, 0]
    thr, off = 0.0, 0.7
    ref_mask = torch.zeros(n, H, W, dtype=torch.bool)
    ref_mask[:, y0:y0 + ch, x0:x0 + cw] = full > thr
    bits = torch.zeros(n, H, W // 32, dtype=torch.int32, device="cuda")
    stats = torch.zeros(n, 8, dtype=torch.int32, device="cuda")
    low_d = low.cuda()
    kcall(gpu_lib, gpu_lib.saber_k_mask_post(ptr(low_d), n, x0, y0, cw, ch, H, W, thr, off, ptr(bits), ptr(stats), None))
    from saber_amd.engine import unpack_bits
    got = unpack_bits(bits, W)
    st = stats.cpu().numpy()
    for i in range(n):
        diff = np.logical_xor(got[i], ref_mask[i].numpy()).sum()
        assert diff <= 3, (i, diff)  # pixels whose logit sits within fp32 rounding of the threshold
        assert abs(int(st[i, 0]) - int(ref_mask[i].sum())) <= 3
        assert abs(int(st[i, 1]) - int((full[i] > thr + off).sum())) <= 3
        assert abs(int(st[i, 2]) - int((full[i] > thr - off).sum())) <= 3
        assert int(st[i, 0]) == int(got[i].sum())
        if got[i].any():
            ys, xs = np.where(got[i])
            assert (st[i, 3], st[i, 4], st[i, 5], st[i, 6]) == (xs.min(), ys.min(), xs.max(), ys.max())
    assert st[4, 0] == 0 and st[4, 5] == -1


@pytest.mark.parametrize("dtype", ["u16", "f32"])
def test_prepare(gpu_lib, dtype):
    from oracle import saber_ref
    img = saber_ref.synthetic_slice(seed=0)
    if dtype == "u16":
        dev = torch.from_numpy(img).cuda()
        ref = saber_ref.prepare(img.astype(np.float32))
        dt = 0
    else:
        f = (img.astype(np.float32) - 30000.0) / 7.0
        dev = torch.from_numpy(f).cuda()
        ref = saber_ref.prepare(f)
        dt = 1
    out = torch.zeros(1024, 1024, device="cuda")
    ws = torch.zeros(4, 1024, 1024, device="cuda")
    mm = torch.zeros(2, dtype=torch.int32, device="cuda")
    kcall(gpu_lib, gpu_lib.saber_k_prepare(ptr(dev), dt, 1024, 1024, ptr(out), ptr(ws), ptr(mm), None))
    err = np.abs(out.cpu().numpy() - ref).max()
    assert err < 2e-4, err
    assert out.min().item() == 0.0 and abs(out.max().item() - 1.0) < 1e-6


def test_perm_index_is_window_contiguous(lib):
    # every Hiera-L window is a contiguous, aligned run of rows; 2x2 pooling groups are 4 consecutive rows
    for stage, win in ((0, 8), (1, 4), (2, 16), (3, 8)):
        g = 256 >> stage
        idx = np.array([[lib.saber_k_perm_index(y, x, stage) for x in range(g)] for y in range(g)])
        assert sorted(idx.ravel().tolist()) == list(range(g * g))
        for wy in range(0, g, win):
            for wx in range(0, g, win):
                w = idx[wy:wy + win, wx:wx + win].ravel()
                assert w.max() - w.min() == win * win - 1 and w.min() % (win * win) == 0
        for y in range(0, g if stage < 3 else 0, 2):  # no pooling happens out of the last stage
            for x in range(0, g, 2):
                q = idx[y:y + 2, x:x + 2].ravel()
                assert q.tolist() == list(range(q[0], q[0] + 4)) and q[0] % 4 == 0


@pytest.mark.parametrize("M,N,K,act,use_res", [
    (32768, 576, 576, 0, True),      # stage-2 proj: fp32 + residual fast epilogue of the persistent direct-to-LDS kernel
    (32768, 2304, 576, 1, False),    # stage-2 fc1: bf16 + GELU epilogue through the LDS transposition
    (65536, 432, 144, 0, False),     # stage-0 qkv: K = 144 on zero-padded weight rows (engine upload layout)
    (70000, 288, 1152, 0, True),     # ragged M, N not a multiple of the 128 tile
])
def test_gemm_direct_to_lds(gpu_lib, M, N, K, act, use_res):
    """Shapes large enough for gemm_bf16_glds_kernel<4> (the kernel every Hiera block runs); torch fp64 on the GPU is the reference."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    Kp = (K + 63) // 64 * 64
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = torch.zeros(N, Kp, device="cuda", dtype=torch.bfloat16)
    W[:, :K] = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g) if use_res else None
    ref = A.double() @ W[:, :K].double().T + bias.double()
    if act == 1:
        ref = F.gelu(ref)
    if use_res:
        ref = ref + res.double()
    if use_res:
        out = torch.zeros(M, N, device="cuda")
        kcall(gpu_lib, gpu_lib.saber_k_gemm_ld(ptr(A), K, ptr(W), Kp, 1, ptr(bias), ptr(res), ptr(out), None, M, N, K, act, None))
        err = (out.double() - ref).abs().max().item()
        assert err < 2e-4, err
    else:
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        kcall(gpu_lib, gpu_lib.saber_k_gemm_ld(ptr(A), K, ptr(W), Kp, 1, ptr(bias), None, None, ptr(out), M, N, K, act, None))
        err = ((out.double() - ref).abs() / (ref.abs() + 1.0)).max().item()
        assert err < 2.0 ** -8, err      # one bf16 rounding of the output (+ the 2.6e-5 GELU fit)


def _dec_inputs(P, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s, scale=1.0: (torch.randn(*s, device="cuda", generator=g) * scale)
    X = r(P, 4096, 256).to(torch.bfloat16)
    pe = r(4096, 256).to(torch.bfloat16)
    return g, r, X, pe


def _blockdiag_pe_scores(tproj, pproj, scale):
    """scale * sum_i tproj[p, t, 16h + i] * pproj[n, 16h + i] -> [P, 64 = 8h + t, 4096] (the positional term of a folded attention)"""
    P = tproj.shape[0]
    tp = tproj.double().view(P, 8, 8, 16).permute(0, 2, 1, 3)          # [p][h][t][i]
    pp = pproj.double().view(4096, 8, 16)                              # [n][h][i]
    return scale * torch.einsum("phti,nhi->phtn", tp, pp).reshape(P, 64, 4096)


@pytest.mark.parametrize("P,shared", [(3, False), (2, True), (70, False)])
def test_dec_i2t(gpu_lib, P, shared):
    """Folded image->token attention + residual + LayerNorm (dec_i2t_kernel) against the plain formula in fp64:
    out = LN(x + softmax_per_head(x Kt^T + kscale blockdiag(tk) peq^T + cb) Vt + bo).  Scores live in the exp2 domain."""
    g, r, X, pe = _dec_inputs(1 if shared else P, 11 + P)
    Kt = r(P, 64, 256, scale=0.08).to(torch.bfloat16)
    peq = r(4096, 128, scale=1.0).to(torch.bfloat16)
    tk = r(P * 8, 128, scale=1.0)
    kscale = 0.3
    cb = r(P, 64)
    VtT = r(P, 256, 64, scale=0.5).to(torch.bfloat16)
    bo, gamma, beta = r(256), 1.0 + 0.1 * r(256), 0.1 * r(256)
    out = torch.zeros(P, 4096, 256, device="cuda", dtype=torch.bfloat16)
    kcall(gpu_lib, gpu_lib.saber_k_dec_i2t(ptr(X), 0 if shared else 4096 * 256, ptr(peq), ptr(Kt), ptr(tk), kscale, ptr(cb), ptr(VtT), ptr(bo), ptr(gamma),
                                          ptr(beta), 1e-5, ptr(out), P, None))
    Xd = X.double().expand(P, -1, -1)
    tkb = (tk * kscale).to(torch.bfloat16).view(P, 8, 128)              # the kernel rounds the scaled projection to bf16 for the MFMA
    S = Xd @ Kt.double().transpose(1, 2) + _blockdiag_pe_scores(tkb, peq, 1.0).transpose(1, 2) + cb.double()[:, None, :]
    Pm = torch.softmax(S.view(P, 4096, 8, 8) * np.log(2.0), dim=-1).view(P, 4096, 64)
    Y = Pm.to(torch.bfloat16).double() @ VtT.double().transpose(1, 2)        # the kernel rounds P to bf16 for the second MFMA
    ref = F.layer_norm(Xd + Y + bo.double(), (256,), gamma.double(), beta.double(), 1e-5)
    err = (out.double() - ref).abs().max().item()
    assert err < 0.04, err          # bf16 output of O(1..4) LayerNorm values: one ulp is 2^-7 at magnitude 2..4
    assert (out.double() - ref).pow(2).mean().sqrt().item() < 4e-3


@pytest.mark.parametrize("P,split,shared", [(3, 1, False), (2, 4, True), (5, 8, False)])
def test_dec_t2i(gpu_lib, P, split, shared):
    """Folded token->image attention (dec_t2i_kernel + finish):
    out[p][t][16h+i] = Wv[16h+i] . (sum_n softmax_n(Qt[8h+t] . x_n + qscale tq[h,t] . pek_n[h]) x_n) + bv."""
    g, r, X, pe = _dec_inputs(1 if shared else P, 5 + P)
    Qt = r(P, 64, 256, scale=0.05).to(torch.bfloat16)
    pek = r(4096, 128, scale=1.0).to(torch.bfloat16)
    tq = r(P * 8, 128, scale=1.0)
    qscale = 0.3
    Wv = (r(128, 256) / 16).to(torch.bfloat16)
    bv = r(128)
    part = torch.zeros(P * split * 64 * 256, device="cuda")
    ml = torch.zeros(P * split * 64 * 2, device="cuda")
    out = torch.zeros(P, 8, 128, device="cuda", dtype=torch.bfloat16)
    kcall(gpu_lib, gpu_lib.saber_k_dec_t2i(ptr(X), 0 if shared else 4096 * 256, ptr(pek), ptr(Qt), ptr(tq), qscale, ptr(part), ptr(ml), P, split, ptr(Wv),
                                          ptr(bv), ptr(out), None))
    Xd = X.double().expand(P, -1, -1)
    tqb = (tq * qscale).to(torch.bfloat16).view(P, 8, 128)
    S = Qt.double() @ Xd.transpose(1, 2) + _blockdiag_pe_scores(tqb, pek, 1.0)   # [P, 64, 4096], exp2 domain
    Pm = torch.softmax(S * np.log(2.0), dim=-1)
    Z = Pm @ Xd                                                              # [P, 64 = 8h + t, 256]
    Z = Z.view(P, 8, 8, 256)                                                 # [p][h][t][256]
    ref = torch.einsum("phtd,hid->pthi", Z, Wv.double().view(8, 16, 256)).reshape(P, 8, 128) + bv.double()
    err = (out.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err < 0.02 * scale, (err, scale)      # P is rounded to bf16 before the PV MFMA, output stored as bf16


def test_hand_synchronised_kernels_are_run_to_run_identical(gpu_lib):
    """Race screen for the kernels that order LDS-DMA traffic by hand (counted vmcnt + raw barriers): the same launch repeated must be
    bit-identical every time (a read that overtakes its DMA shows up as run-to-run differences long before it fails a tolerance)."""
    g = torch.Generator().manual_seed(99)
    # staggered 256x256 GEMM (its normal route: >= 1024 tiles, N >= 1024)
    M, N, K = 256 * 130, 2304, 576
    _, Ad = bf(torch.randn(M, K, generator=g))
    _, Wd = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g).cuda()
    outs = []
    for _ in range(6):
        o = torch.zeros(M, N, dtype=torch.int16, device="cuda")
        kcall(gpu_lib, gpu_lib.saber_k_gemm(ptr(Ad), ptr(Wd), ptr(bias), None, None, ptr(o), M, N, K, 1, 0, 0, 0, 0, None))
        outs.append(o)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    # streaming global attention and the 256-key window kernel
    for nw, nk, heads in ((5, 4096, 8), (70, 256, 8)):
        _, qd = bf(torch.randn(nw * nk, 3 * heads * 72, generator=g) * 1.5)
        outs = []
        for _ in range(6):
            o = torch.zeros(nw * nk, heads * 72, dtype=torch.int16, device="cuda")
            kcall(gpu_lib, gpu_lib.saber_k_hiera_attention(ptr(qd), ptr(o), nw, nk, heads, 0, None))
            outs.append(o)
        assert all(torch.equal(outs[0], o) for o in outs[1:])
    # decoder streaming kernels
    P = 300
    gg, r, X, pe = _dec_inputs(P, 123)
    Kt = r(P, 64, 256, scale=0.08).to(torch.bfloat16); peq = r(4096, 128).to(torch.bfloat16); tk = r(P * 8, 128); cb = r(P, 64)
    VtT = r(P, 256, 64, scale=0.5).to(torch.bfloat16); bo, gamma, beta = r(256), 1.0 + 0.1 * r(256), 0.1 * r(256)
    Wv = (r(128, 256) / 16).to(torch.bfloat16); bv = r(128)
    part = torch.zeros(P * 64 * 256, device="cuda"); ml = torch.zeros(P * 64 * 2, device="cuda")
    o1, o2 = [], []
    for _ in range(4):
        a = torch.zeros(P, 4096, 256, device="cuda", dtype=torch.bfloat16)
        kcall(gpu_lib, gpu_lib.saber_k_dec_i2t(ptr(X), 4096 * 256, ptr(peq), ptr(Kt), ptr(tk), 0.3, ptr(cb), ptr(VtT), ptr(bo), ptr(gamma), ptr(beta), 1e-5, ptr(a), P, None))
        b = torch.zeros(P, 8, 128, device="cuda", dtype=torch.bfloat16)
        kcall(gpu_lib, gpu_lib.saber_k_dec_t2i(ptr(X), 4096 * 256, ptr(peq), ptr(Kt), ptr(tk), 0.3, ptr(part), ptr(ml), P, 1, ptr(Wv), ptr(bv), ptr(b), None))
        o1.append(a); o2.append(b)
    assert all(torch.equal(o1[0], o) for o in o1[1:]) and all(torch.equal(o2[0], o) for o in o2[1:])


@pytest.mark.parametrize("M,N,K,with_res,with_bf", [(4096, 576, 576, True, False), (4096 + 70, 576, 2304, True, True), (16384, 288, 1152, True, False),
                                                    (2048 + 300, 288, 288, False, True), (65536, 144, 144, True, False), (4096 + 33, 144, 576, True, True),
                                                    (128 * 356 + 17, 576, 576, True, False)])      # 357 tiles on 256 workgroups: uneven walks, the short ones start late
def test_gemm_rowln(gpu_lib, M, N, K, with_res, with_bf):
    """residual GEMM + the LayerNorm that follows it in one kernel (gemm_rowln.hip): y against fp64 on the same bf16 operands,
    the normalised bf16 rows against LayerNorm of the kernel's own y (<= 1 bf16 ulp, almost all exact) and of the fp64 y"""
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.7).to(torch.bfloat16)
    Kp = (K + 63) // 64 * 64
    Wf = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    Wp = torch.zeros(N, Kp, dtype=torch.bfloat16)
    Wp[:, :K] = Wf
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) * 2 + 0.5 if with_res else None
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    Ad, Wd = A.view(torch.uint16).cuda(), Wp.view(torch.uint16).cuda()
    out = (res.clone() if with_res else torch.zeros(M, N)).cuda()        # in place: res aliases out_f32, as the engine's residual stream does
    out_bf = torch.zeros(M, N, dtype=torch.uint16, device="cuda") if with_bf else None
    ln_out = torch.zeros(M, N, dtype=torch.uint16, device="cuda")
    bd, gd, bed = bias.cuda(), gamma.cuda(), beta.cuda()
    kcall(gpu_lib, gpu_lib.saber_k_gemm_rowln(ptr(Ad), K, ptr(Wd), Kp, ptr(bd), ptr(out) if with_res else None, ptr(out), ptr(out_bf) if with_bf else None,
                                              ptr(gd), ptr(bed), 1e-6, ptr(ln_out), M, N, K, None))
    torch.cuda.synchronize()
    ref = A.double() @ Wf.double().T + bias.double() + (res.double() if with_res else 0.0)
    y = out.cpu()
    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, err
    if with_bf:
        assert torch.equal(from_bf(out_bf), y.to(torch.bfloat16).float())
    ln_self = torch.nn.functional.layer_norm(y, (N,), gamma, beta, 1e-6)
    got = from_bf(ln_out)
    d = (got - ln_self.to(torch.bfloat16).float()).abs()
    ulp = ln_self.abs().clamp(min=1e-3) * 2.0 ** -7
    assert (d <= ulp).all(), float((d / ulp).max())
    assert (d > 0).float().mean().item() < 2e-3         # rounding flips only
    ln_ref = torch.nn.functional.layer_norm(ref, (N,), gamma.double(), beta.double(), 1e-6)
    assert (got.double() - ln_ref).abs().max().item() < 0.05


def test_gemm_rowln_late_start_changes_nothing(gpu_lib):
    """gemm_rowln_kernel starts the workgroups with the shorter tile walk late (timing only): same bits with the late start switched off
    (saber_k_set_debug(32768)), on a problem whose tile count (357) is not a multiple of the grid (256)"""
    M, N, K = 128 * 356 + 17, 576, 2304
    g = torch.Generator().manual_seed(5)
    A = (torch.randn(M, K, generator=g) * 0.7).to(torch.bfloat16).view(torch.uint16).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).view(torch.uint16).cuda()
    bias, gamma, beta = torch.randn(N, generator=g).cuda(), (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    res = (torch.randn(M, N, generator=g) * 2).cuda()
    outs = []
    for flag in (0, 32768, 0):
        gpu_lib.saber_k_set_debug(flag)
        try:
            y = res.clone()
            ln = torch.zeros(M, N, dtype=torch.uint16, device="cuda")
            kcall(gpu_lib, gpu_lib.saber_k_gemm_rowln(ptr(A), K, ptr(W), K, ptr(bias), ptr(y), ptr(y), None, ptr(gamma), ptr(beta), 1e-6, ptr(ln), M, N, K, None))
            torch.cuda.synchronize()
        finally:
            gpu_lib.saber_k_set_debug(0)
        outs.append((y, ln))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])

