"""Micro-benchmark of K8 (mask_post_w1k_kernel) on the three crop sizes of the default AMG pyramid (GPU box only):
python tools/mask_post_bench.py [n_masks]   (SABER_AMD_LIB=<path> selects a differently built library: A/B of compile-time variants)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
def ptr(t): return C.c_void_p(t.data_ptr())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
g = torch.Generator(device="cuda").manual_seed(0)
low = F.interpolate(torch.randn(n, 1, 16, 16, device="cuda", generator=g) * 4, size=(256, 256), mode="bicubic")[:, 0].contiguous()
bits = torch.zeros(n, 1024, 32, dtype=torch.int32, device="cuda"); stats = torch.zeros(n, 8, dtype=torch.int32, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (x0, y0, cw, ch) in ((0, 0, 1024, 1024), (427, 0, 597, 597), (331, 662, 362, 362)):
    run = lambda: lib.saber_k_mask_post(ptr(low), n, x0, y0, cw, ch, 1024, 1024, 0.0, 0.7, ptr(bits), ptr(stats), st)
    for _ in range(3): assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{os.path.basename(os.environ.get('SABER_AMD_LIB', 'libsaber_amd.so')):28s} crop {cw:4d}x{ch:<4d} {n} masks: {us:8.1f} us = {us / n:6.3f} us per mask, {n * cw * ch / us / 1e6:6.2f} Tpx/s, "
          f"{n * (262144 + 131072) / us / 1e3:7.1f} GB/s", flush=True)
