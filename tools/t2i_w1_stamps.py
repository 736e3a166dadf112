"""In-kernel cycle stamps of dec_t2i_w1_kernel (development): python tools/t2i_w1_stamps.py [P]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
def ptr(t): return C.c_void_p(t.data_ptr())
def bf(t): return t.to(torch.bfloat16).contiguous()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator(device="cuda").manual_seed(0)
X = bf(torch.randn(P, 4096, 256, device="cuda", generator=g))
pp = bf(torch.randn(4096, 128, device="cuda", generator=g)); tproj = torch.randn(P * 8, 128, device="cuda", generator=g)
Qt = bf(torch.randn(P, 64, 256, device="cuda", generator=g) * 0.05)
Wv = bf(torch.randn(128, 256, device="cuda", generator=g) / 16); bv = torch.randn(128, device="cuda", generator=g)
o2 = torch.empty(P, 8, 128, device="cuda", dtype=torch.bfloat16)
part = torch.empty(P * 64 * 256, device="cuda"); ml = torch.empty(P * 64 * 2, device="cuda")
call = lambda: lib.saber_k_dec_t2i(ptr(X), 4096 * 256, ptr(pp), ptr(Qt), ptr(tproj), 0.3, ptr(part), ptr(ml), P, 1, ptr(Wv), ptr(bv), ptr(o2), None)
for flag, name in ((0x20000000, "dec_t2i_kernel<8>"), (0x10000000, "dec_t2i_w1_kernel")):
    lib.saber_k_set_debug(flag)
    for _ in range(3): call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): call()
    e.record(); torch.cuda.synchronize()
    print(f"{name}: {s.elapsed_time(e) / 10 * 1e3:.1f} us per launch of {P} prompts")
st = torch.zeros(P * 8 * 6, dtype=torch.int64, device="cuda")
lib.saber_k_set_stamp_buffer(ptr(st)); call(); torch.cuda.synchronize(); lib.saber_k_set_stamp_buffer(None)
lib.saber_k_set_debug(0)
s = st[:P * 4 * 4].view(P, 4, 4).double().cpu() / 32.0
names = ["wait + scores (72 MFMA, 48 ds_read_b128)", "softmax x 4 (+ rescale)", "PV (64 MFMA, 32 tr reads)", "refill (16 LDS-DMA pieces)"]
print("s_memtime ticks per 32-key step of a wave (4 query tiles), mean over workgroups; per wave:")
for k, n in enumerate(names):
    print(f"  {n:44s} " + " ".join(f"{s[:, w, k].mean():7.0f}" for w in range(4)) + f"   | all {s[:, :, k].mean():7.0f}")
print("  total per step", s.sum(-1).mean().item())
