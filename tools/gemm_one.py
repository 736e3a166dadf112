import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
M, N, K = [int(x) for x in sys.argv[1:4]]
def ptr(t): return C.c_void_p(t.data_ptr())
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5): lib.saber_k_gemm(ptr(A), ptr(W), ptr(bias), None, None, ptr(out), M, N, K, 0, 0, 0, 0, 0, None)
torch.cuda.synchronize()
