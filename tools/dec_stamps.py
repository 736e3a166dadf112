"""In-kernel cycle stamps of dec_i2t (development): python tools/dec_stamps.py [P]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
def ptr(t): return C.c_void_p(t.data_ptr())
def bf(t): return t.to(torch.bfloat16).contiguous()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator(device="cuda").manual_seed(0)
pe = bf(torch.randn(4096, 256, device="cuda", generator=g)); X = bf(torch.randn(P, 4096, 256, device="cuda", generator=g))
pp = bf(torch.randn(4096, 128, device="cuda", generator=g)); tproj = torch.randn(P * 8, 128, device="cuda", generator=g)
Kt = bf(torch.randn(P, 64, 256, device="cuda", generator=g) * 0.05); cb = torch.randn(P, 64, device="cuda", generator=g)
VtT = bf(torch.randn(P, 256, 64, device="cuda", generator=g)); bo = torch.randn(256, device="cuda", generator=g)
ga = torch.ones(256, device="cuda"); be = torch.zeros(256, device="cuda"); out = torch.empty_like(X)
RT = 2 if int(os.environ.get("DBG", "0"), 0) & 1 else 1
lib.saber_k_set_debug(int(os.environ.get("DBG", "0"), 0))
st = torch.zeros(P * 8 * 6, dtype=torch.int64, device="cuda")
call = lambda: lib.saber_k_dec_i2t(ptr(X), 4096 * 256, ptr(pp), ptr(Kt), ptr(tproj), 0.3, ptr(cb), ptr(VtT), ptr(bo), ptr(ga), ptr(be), 1e-5, ptr(out), P, None)
for _ in range(3): call()
lib.saber_k_set_stamp_buffer(ptr(st)); call(); torch.cuda.synchronize(); lib.saber_k_set_stamp_buffer(None)
s = st[:P * 4 * RT * 6].view(P, 4 * RT, 6).double().cpu() / (4096.0 / (16 * RT))   # per tile
names = ["gemm1+softmax+Pwrite", "vmcnt wait", "barrier", "issue DMA", "finish_tile(t-1)+stores", "gemm2+residual+stats"]
print("cycles per tile (s_memtime ticks), mean over blocks; per wave:")
for k, n in enumerate(names):
    print(f"  {n:28s} " + " ".join(f"{s[:, w, k].mean():7.0f}" for w in range(4 * RT)) + f"   | all {s[:, :, k].mean():7.0f}")
print("  total per tile", s.sum(-1).mean().item(), f"(tiles of {16 * RT} rows)")
# ---- t2i (64-key blocks, 64 per prompt)
Qt = bf(torch.randn(P, 64, 256, device="cuda", generator=g) * 0.05)
Wv = bf(torch.randn(128, 256, device="cuda", generator=g) / 16); bv = torch.randn(128, device="cuda", generator=g)
o2 = torch.empty(P, 8, 128, device="cuda", dtype=torch.bfloat16)
part = torch.empty(P * 64 * 256, device="cuda"); ml = torch.empty(P * 64 * 2, device="cuda")
call2 = lambda: lib.saber_k_dec_t2i(ptr(X), 4096 * 256, ptr(pp), ptr(Qt), ptr(tproj), 0.3, ptr(part), ptr(ml), P, 1, ptr(Wv), ptr(bv), ptr(o2), None)
for _ in range(3): call2()
st.zero_(); lib.saber_k_set_stamp_buffer(ptr(st)); call2(); torch.cuda.synchronize(); lib.saber_k_set_stamp_buffer(None)
NW = 4 if int(os.environ.get("DBG", "0"), 0) & 4 else 8
s = st[:P * NW * 6].view(P, NW, 6).double().cpu() / (4096.0 / (8 * NW))
names = ["issue DMA", "QK^T (18 ds_read + 18 mfma)", "softmax", "PV (32 tr_read + 16 mfma)", "vmcnt wait", "barrier"]
print(f"t2i cycles per {8 * NW}-key block; per wave:")
for k, n in enumerate(names):
    print(f"  {n:28s} " + " ".join(f"{s[:, w, k].mean():7.0f}" for w in range(NW)) + f"   | all {s[:, :, k].mean():7.0f}")
print("  total per block", s.sum(-1).mean().item())
