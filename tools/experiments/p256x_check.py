"""Development: the 32x32x16 persistent GEMM (SABER_AMD_P256X=1) against torch on a few shapes, then its time next to the default kernel's.
SABER_AMD_P256X=1 python tools/p256x_check.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
lib.saber_k_set_debug(128)          # force the persistent 256 x 256 path
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K, act in ((2048, 1024, 576, 1), (700, 520, 288, 0), (86016, 2304, 576, 1), (86016, 1728, 576, 0), (21504, 4608, 1152, 1)):
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16); ldw = (K + 63) // 64 * 64
    W = torch.zeros(N, ldw, device="cuda", dtype=torch.bfloat16); W[:, :K] = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g); out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    call = lambda: lib.saber_k_gemm_ld(p(A), K, p(W), ldw, 1, p(bias), None, None, p(out), M, N, K, act, None)
    assert call() == 0, lib.saber_k_last_error()
    torch.cuda.synchronize()
    rows = slice(0, min(M, 4096))
    ref = A[rows].float() @ W[:, :K].float().T + bias
    if act: ref = torch.nn.functional.gelu(ref)
    err = ((out[rows].float() - ref).abs().max() / ref.abs().max()).item()
    tail = slice(max(0, M - 300), M)
    ref2 = A[tail].float() @ W[:, :K].float().T + bias
    if act: ref2 = torch.nn.functional.gelu(ref2)
    err2 = ((out[tail].float() - ref2).abs().max() / ref2.abs().max()).item()
    for _ in range(3): call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): call()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    print(f"M={M} N={N} K={K} act={act}: max err / max |ref| first rows {err:.2e}, last rows {err2:.2e}; {us:.1f} us = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s", flush=True)
