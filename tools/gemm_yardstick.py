"""Yardstick, not product: the encoder's GEMM shapes (Hiera-L, 21 crops per slice) through this repository's kernels (saber_k_gemm: bias,
bf16 out; GELU for fc1) and through the vendor library as PyTorch calls it (torch.nn.functional.linear on bf16 = hipBLASLt / rocBLAS),
same operands, TFLOP/s each.  Says how far the hand-written kernels are from what the library reaches on the same shapes."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from saber_amd import _lib
    lib = _lib.load()
    assert lib.saber_k_init(0) == 0
    imgs = 21
    shapes = []
    for stage, (dim, toks) in enumerate(((144, 65536), (288, 16384), (576, 4096), (1152, 1024))):
        M = imgs * toks
        shapes += [(f"s{stage + 1} qkv", M, 3 * dim, dim, 0), (f"s{stage + 1} fc1", M, 4 * dim, dim, 1), (f"s{stage + 1} fc2", M, dim, 4 * dim, 0),
                   (f"s{stage + 1} proj", M, dim, dim, 0)]
    p = lambda t: C.c_void_p(t.data_ptr())
    warm = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
    for _ in range(50):
        warm @ warm
    torch.cuda.synchronize()
    print(f"{'shape':10s} {'M':>8s} {'N':>5s} {'K':>5s} | {'ours us':>9s} {'TF/s':>7s} | {'library us':>10s} {'TF/s':>7s}")
    for name, M, N, K, act in shapes:
        A = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")

        Kp = (K + 63) // 64 * 64                      # the engine keeps W rows zero-padded to a multiple of 64 in K (w_kpad)
        Wp = torch.zeros(N, Kp, dtype=torch.bfloat16, device="cuda")
        Wp[:, :K] = W

        def ours():
            st = lib.saber_k_gemm_ld(p(A.view(torch.uint16)), K, p(Wp.view(torch.uint16)), Kp, 1, p(b), None, None, p(out.view(torch.uint16)), M, N, K, act,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert st == 0, lib.saber_k_last_error()
        bb = b.to(torch.bfloat16)

        def vendor():
            y = torch.nn.functional.linear(A, W, bb)
            return torch.nn.functional.gelu(y) if act else y

        def vendor_mm():            # the GEMM alone (GELU is a second kernel in the library path)
            return torch.nn.functional.linear(A, W, bb)

        def t(f, reps=40):
            for _ in range(10):                       # clocks ramp over tens of milliseconds: an unwarmed 5-launch sample reads up to 2x slow
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        to, tv = t(ours), t(vendor_mm)
        to, tv = min(to, t(ours)), min(tv, t(vendor_mm))
        fl = 2.0 * M * N * K
        ref = vendor().float()
        err = ((out.float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        print(f"{name:10s} {M:8d} {N:5d} {K:5d} | {to:9.1f} {fl / to * 1e-6:7.1f} | {tv:10.1f} {fl / tv * 1e-6:7.1f}   rel diff {err:.1e}", flush=True)
        del A, W, out, ref


if __name__ == "__main__":
    main()
