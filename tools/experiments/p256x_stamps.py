"""Development: cycle stamps of the 32x32x16 persistent GEMM.  SABER_AMD_P256X=1 python tools/p256x_stamps.py M N K [act]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
def ptr(t): return C.c_void_p(t.data_ptr())
M, N, K = [int(x) for x in sys.argv[1:4]]; act = int(sys.argv[4]) if len(sys.argv) > 4 else 0
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
lib.saber_k_set_debug(int(os.environ.get("DBG", "128"), 0))
call = lambda: lib.saber_k_gemm_ld(ptr(A), K, ptr(W), K, 1, ptr(bias), None, None, ptr(out), M, N, K, act, None)
for _ in range(3): call()
st = torch.zeros(256 * 8 * 6, dtype=torch.int64, device="cuda")
lib.saber_k_set_stamp_buffer(ptr(st)); call(); torch.cuda.synchronize(); lib.saber_k_set_stamp_buffer(None)
tiles = ((M + 255) // 256) * ((N + 255) // 256); nk = (K + 31) // 32
s = st.view(256, 8, 6).double().cpu()
names = ["sub-step 0 (8 MFMA + 6 reads + 2 DMA)", "vmcnt wait", "barrier", "sub-step 1", "epilogue + bookkeeping (per tile)"]
iters = tiles / 256.0 * nk
print(f"tiles {tiles} ({tiles / 256:.2f} per block), {nk} K-tiles each; cycles per K-tile (epilogue: per tile), per wave:")
for k, n in enumerate(names):
    d = tiles / 256.0 if k == 4 else iters
    print(f"  {n:40s} " + " ".join(f"{s[:, w, k].mean() / d:7.0f}" for w in range(8)))
print("  per K-tile incl. epilogue share", s.sum(-1).mean().item() / iters)
