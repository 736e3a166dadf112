"""LayerNorm kernel timing on the encoder's shapes (rows = 21 crops x tokens): python tools/ln_bench.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
if os.environ.get("OLD"): _lib.LIB_PATH = _lib.LIB_PATH.replace("libsaber_amd.so", "libsaber_amd_old.so")
lib = _lib.load()
ptr = lambda t: C.c_void_p(t.data_ptr())
for rows, Cc in ((21 * 65536, 144), (21 * 16384, 288), (21 * 4096, 576), (21 * 1024, 1152)):
    x = torch.randn(rows, Cc, device="cuda"); g = torch.randn(Cc, device="cuda"); b = torch.randn(Cc, device="cuda")
    ob = torch.zeros(rows, Cc, dtype=torch.int16, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): lib.saber_k_layernorm(ptr(x), ptr(g), ptr(b), 1e-6, None, ptr(ob), rows, Cc, 0, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.saber_k_layernorm(ptr(x), ptr(g), ptr(b), 1e-6, None, ptr(ob), rows, Cc, 0, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"rows={rows:8d} C={Cc:5d}  {us:8.1f} us  {rows * Cc * 6 / us / 1e6:6.2f} TB/s")
