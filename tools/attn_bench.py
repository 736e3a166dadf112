import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
def ptr(t): return C.c_void_p(t.data_ptr())
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (nw, nk, heads) in [(336, 256, 8), (21, 4096, 8)]:
    qkv = (torch.randn(nw * nk, 3 * heads * 72, device="cuda") * 1.5).to(torch.bfloat16)
    out = torch.empty(nw * nk, heads * 72, device="cuda", dtype=torch.bfloat16)
    for flags in (0, 2, 32):
        lib.saber_k_set_debug(flags)
        us = t(lambda: lib.saber_k_hiera_attention(ptr(qkv), ptr(out), nw, nk, heads, 0, None))
        fl = 4.0 * 72 * nw * nk * nk * heads
        print(f"nw={nw} nk={nk} flags={flags}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s", flush=True)
lib.saber_k_set_debug(0)
