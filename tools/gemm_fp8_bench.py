"""MXFP8 GEMM (csrc/gemm_fp8.hip) against the engine's tuned bf16 GEMMs on the Hiera-L stage-2 / stage-3 shapes of a 21-crop pass:
python tools/gemm_fp8_bench.py   ->   microseconds and TFLOP/s per shape, both paths, same box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
def timeit(fn, n=20):
    for _ in range(5): assert fn() == 0, lib.saber_k_last_error()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, form in (("s2 qkv", 86016, 1728, 576, "bf16"), ("s2 fc1", 86016, 2304, 576, "mx"), ("s2 fc2", 86016, 576, 2304, "f32"), ("s3 qkv", 21504, 3456, 1152, "bf16"),
                            ("s3 fc1", 21504, 4608, 1152, "mx"), ("s3 fc2", 21504, 1152, 4608, "f32")):
    Kp = (K + 127) // 128 * 128; Mp = (M + 767) // 768 * 768; Np = (N + 191) // 192 * 192
    A8 = torch.randint(0, 120, (M, Kp), dtype=torch.uint8, device="cuda", generator=g); W8 = torch.randint(0, 120, (N, Kp), dtype=torch.uint8, device="cuda", generator=g)
    sa = torch.full((Kp // 128, Mp, 4), 120, dtype=torch.uint8, device="cuda"); sw = torch.full((Kp // 128, Np, 4), 120, dtype=torch.uint8, device="cuda")
    bias = torch.zeros(N, device="cuda")
    outb = torch.empty(M, N, dtype=torch.uint16, device="cuda")
    outf = torch.empty(M, N, device="cuda") if form == "f32" else None
    res = torch.zeros(M, N, device="cuda") if form == "f32" else None
    o8 = torch.empty(M, N, dtype=torch.uint8, device="cuda"); os_ = torch.empty((N + 127) // 128, Mp, 4, dtype=torch.uint8, device="cuda")
    Ab = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16).view(torch.uint16); ldw = (K + 63) // 64 * 64
    Wb = torch.randn(N, ldw, device="cuda", generator=g).to(torch.bfloat16).view(torch.uint16)
    xf = torch.randn(M, K, device="cuda", generator=g); gam = torch.ones(K, device="cuda")
    args = (p(A8), Kp, p(sa), Mp, p(W8), Kp, p(sw), Np, p(bias))
    if form == "bf16": f8 = lambda: lib.saber_k_gemm_mx(*args, None, None, p(outb), None, None, 0, N, M, N, Kp, 0, None)
    elif form == "mx": f8 = lambda: lib.saber_k_gemm_mx(*args, None, None, None, p(o8), p(os_), Mp, N, M, N, Kp, 1, None)
    else: f8 = lambda: lib.saber_k_gemm_mx(*args, p(res), p(outf), None, None, None, 0, N, M, N, Kp, 0, None)
    t8 = timeit(f8)
    if form == "f32": f16 = lambda: lib.saber_k_gemm_ld(p(Ab), K, p(Wb), ldw, 1, p(bias), p(res), p(outf), None, M, N, K, 0, None)
    else: f16 = lambda: lib.saber_k_gemm_ld(p(Ab), K, p(Wb), ldw, 1, p(bias), None, None, p(outb), M, N, K, 1 if form == "mx" else 0, None)
    t16 = timeit(f16)
    extra = ""
    if K <= 1152:
        tq = timeit(lambda: lib.saber_k_ln_mx(p(xf), K, p(gam), p(bias[:K] if N >= K else gam), 1e-6, K, p(A8), Kp, Kp, p(sa), Mp, M, None))
        extra = f" | LayerNorm -> MX of A {tq:6.1f} us"
    fl = 2.0 * M * N * K
    print(f"{name}: M={M} N={N} K={K} out={form}: mxfp8 {t8:7.1f} us = {fl / t8 / 1e6:7.0f} TFLOP/s | bf16 kernel {t16:7.1f} us = {fl / t16 / 1e6:7.0f} TFLOP/s | x{t16 / t8:.2f}{extra}", flush=True)
