"""Microbenchmark of the fused decoder kernels through the kernel C-ABI: python tools/dec_bench.py i2t [P ...]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0, lib.saber_k_last_error()
def ptr(t): return C.c_void_p(t.data_ptr())
def bf(t): return t.to(torch.bfloat16).contiguous()
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
which = sys.argv[1]
Ps = [int(x) for x in sys.argv[2:]] or [64, 256, 1024]
g = torch.Generator(device="cuda").manual_seed(0)
pe = bf(torch.randn(4096, 256, device="cuda", generator=g))
for P in Ps:
    X = bf(torch.randn(P, 4096, 256, device="cuda", generator=g))
    pp = bf(torch.randn(4096, 128, device="cuda", generator=g)); tproj = torch.randn(P * 8, 128, device="cuda", generator=g)
    if which == "i2t":
        Kt = bf(torch.randn(P, 64, 256, device="cuda", generator=g) * 0.05); cb = torch.randn(P, 64, device="cuda", generator=g)
        VtT = bf(torch.randn(P, 256, 64, device="cuda", generator=g)); bo = torch.randn(256, device="cuda", generator=g)
        ga = torch.ones(256, device="cuda"); be = torch.zeros(256, device="cuda"); out = torch.empty_like(X)
        for flags in [0, 0x100, 0x200, 0x400, 0x800]:
            lib.saber_k_set_debug(flags)
            us = timeit(lambda: lib.saber_k_dec_i2t(ptr(X), 4096 * 256, ptr(pp), ptr(Kt), ptr(tproj), 0.3, ptr(cb), ptr(VtT), ptr(bo), ptr(ga), ptr(be), 1e-5, ptr(out), P, None))
            print(f"i2t P={P:5d} flags={flags:#06x}: {us:8.1f} us   {P * 4096 * 256 * 4 / us / 1e6:6.2f} TB/s", flush=True)
        lib.saber_k_set_debug(0)
    elif which == "t2i":
        Qt = bf(torch.randn(P, 64, 256, device="cuda", generator=g) * 0.05)
        Wv = bf(torch.randn(128, 256, device="cuda", generator=g) / 16); bv = torch.randn(128, device="cuda", generator=g)
        out = torch.empty(P, 8, 128, device="cuda", dtype=torch.bfloat16)
        for split in (1, 2, 4, 8):
            part = torch.empty(P * split * 64 * 256, device="cuda"); ml = torch.empty(P * split * 64 * 2, device="cuda")
            us = timeit(lambda: lib.saber_k_dec_t2i(ptr(X), 4096 * 256, ptr(pp), ptr(Qt), ptr(tproj), 0.3, ptr(part), ptr(ml), P, split, ptr(Wv), ptr(bv), ptr(out), None))
            print(f"t2i P={P:5d} split={split}: {us:8.1f} us   {P * 4096 * 256 * 2 / us / 1e6:6.2f} TB/s", flush=True)
    elif which == "copy":
        out = torch.empty_like(X)
        us = timeit(lambda: out.copy_(X))
        print(f"copy P={P:5d}: {us:8.1f} us   {P * 4096 * 256 * 4 / us / 1e6:6.2f} TB/s (read+write)", flush=True)
        us = timeit(lambda: X.sum())
        print(f"sum  P={P:5d}: {us:8.1f} us   {P * 4096 * 256 * 2 / us / 1e6:6.2f} TB/s (read)", flush=True)
        us = timeit(lambda: out.zero_())
        print(f"zero P={P:5d}: {us:8.1f} us   {P * 4096 * 256 * 2 / us / 1e6:6.2f} TB/s (write)", flush=True)
