import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
def ptr(t): return C.c_void_p(t.data_ptr())
M, N, K = 86016, 2304, 576
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, lda, rows in (("A streamed (lda = K)", K, M), ("A resident (lda = 0: every row reads row 0)", 0, M), ("A streamed, M = 8192", K, 8192), ("A resident, M = 8192", 0, 8192)):
    def run(): lib.saber_k_gemm_ld(ptr(A), lda, ptr(W), K, 1, ptr(bias), None, None, ptr(out), rows, N, K, 0, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:50s} M={rows:6d} {ms*1e3:8.1f} us  {2.0*rows*N*K/ms/1e9:7.1f} TF/s", flush=True)
