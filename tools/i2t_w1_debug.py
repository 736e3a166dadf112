"""Stage-by-stage comparison of dec_i2t_w1_kernel with the four-wave kernel (development): python tools/i2t_w1_debug.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from saber_amd import _lib
from tests.test_gpu_kernels import _dec_inputs, kcall, ptr
lib = _lib.load(); assert lib.saber_k_init(0) == 0
P = 520
g, r, X, pe = _dec_inputs(P, 23)
Kt0 = r(P, 64, 256, scale=0.08).to(torch.bfloat16); peq = r(4096, 128, scale=1.0).to(torch.bfloat16); tk0 = r(P * 8, 128, scale=1.0)
cb0 = r(P, 64); VtT0 = r(P, 256, 64, scale=0.5).to(torch.bfloat16); bo0, gamma, beta = r(256), 1.0 + 0.1 * r(256), 0.1 * r(256)
def run(flag, Kt, tk, cb, VtT, bo):
    lib.saber_k_set_debug(flag)
    out = torch.zeros(P, 4096, 256, device="cuda", dtype=torch.bfloat16)
    kcall(lib, lib.saber_k_dec_i2t(ptr(X), 4096 * 256, ptr(peq), ptr(Kt), ptr(tk), 0.3, ptr(cb), ptr(VtT), ptr(bo), ptr(gamma), ptr(beta), 1e-5, ptr(out), P, None))
    torch.cuda.synchronize(); lib.saber_k_set_debug(0)
    return out.float()
Z = torch.zeros_like
cases = {"V = 0, bo = 0: LN(X)": (Kt0, tk0, cb0, Z(VtT0), Z(bo0)),
         "V = 0: LN(X + bo)": (Kt0, tk0, cb0, Z(VtT0), bo0),
         "K = 0, tk = 0, cb = 0: uniform softmax": (Z(Kt0), Z(tk0), Z(cb0), VtT0, bo0),
         "tk = 0, cb = 0: X.Kt only": (Kt0, Z(tk0), Z(cb0), VtT0, bo0),
         "K = 0, tk = 0: cb only": (Z(Kt0), Z(tk0), cb0, VtT0, bo0),
         "K = 0, cb = 0: positional term only": (Z(Kt0), tk0, Z(cb0), VtT0, bo0),
         "everything": (Kt0, tk0, cb0, VtT0, bo0)}
for name, a in cases.items():
    ref = run(0x40000000, *a); o = run(0, *a)
    d = (o - ref).abs()
    print(f"{name:42s} max diff {d.max().item():.4f}  frac > 0.05: {(d > 0.05).float().mean().item():.2e}  nan {torch.isnan(o).sum().item()}")
print("---- where")
a = cases["V = 0, bo = 0: LN(X)"]
o = run(0, *a)
nanrow = torch.isnan(o).any(-1)                     # [P, 4096]
tile = torch.arange(4096, device="cuda") // 16
print("NaN rows per prompt (first 4 prompts):", nanrow.sum(-1)[:4].tolist(), " tiles with NaN rows, prompt 0:", tile[nanrow[0]].unique().tolist()[:40])
print("NaN rows by row-in-tile:", torch.bincount((torch.arange(4096, device='cuda') % 16).repeat(P, 1)[nanrow], minlength=16).tolist())
a = cases["K = 0, tk = 0, cb = 0: uniform softmax"]
ref = run(0x40000000, *a); o = run(0, *a)
bad = (o - ref).abs() > 0.05
print("uniform: bad by tile (prompt 0, first 24 tiles):", bad[0].view(256, 16, 256).any(-1).any(-1)[:24].int().tolist())
print("uniform: bad fraction by channel tile:", [round(x, 3) for x in bad.view(P, 4096, 16, 16).float().mean((0, 1, 3)).tolist()])
print("uniform: bad fraction by wave (tile % 4):", [round(bad[:, (tile % 4) == w].float().mean().item(), 3) for w in range(4)])
print("uniform: bad fraction by n parity:", [round(bad[:, ((tile // 4) % 2) == k].float().mean().item(), 3) for k in range(2)])
