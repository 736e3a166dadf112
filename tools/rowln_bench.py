"""Micro-benchmark (GPU box only): residual GEMM + LayerNorm as ONE kernel (gemm_rowln.hip) against the pair it replaces
(fp32 + residual GEMM, then LayerNorm) on the Hiera-L residual shapes at the 21-crop batch of the AMG default."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
def ptr(t): return C.c_void_p(t.data_ptr())
B = int(os.environ.get("B", "21"))
OP = os.environ.get('OP', 'bf16')          # 16-bit operand type: bf16 | f16 (saber_k_set_operand_type)
DT = {'bf16': torch.bfloat16, 'f16': torch.float16}[OP]
lib.saber_k_set_operand_type(1 if OP == 'f16' else 0)
if os.environ.get('DBG'): lib.saber_k_set_debug(int(os.environ['DBG'], 0))
shapes = [(4096 * B, 576, 576), (4096 * B, 576, 2304)] if os.environ.get("SHORT") else [(4096 * B, 576, 576), (4096 * B, 576, 2304), (16384 * B, 288, 288), (16384 * B, 288, 1152), (65536 * B, 144, 144), (65536 * B, 144, 576)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
dbgs = [int(v, 0) for v in os.environ['DBGS'].split(',')] if os.environ.get('DBGS') else [None]
for M, N, K in shapes:
    Kp = (K + 63) // 64 * 64
    A = torch.randn(M, K, device="cuda").to(DT); W = torch.zeros(N, Kp, device="cuda", dtype=DT)
    W[:, :K] = (torch.randn(N, K, device="cuda") / K ** 0.5).to(DT)
    bias, g, b = torch.randn(N, device="cuda"), torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
    x = torch.randn(M, N, device="cuda"); xn = torch.empty(M, N, device="cuda", dtype=DT)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if dbgs != [None]:          # sweep of development flags in one process (DBGS=a,b,c): fused kernel only
        for d in dbgs:
            lib.saber_k_set_debug(d)
            t = timeit(lambda: lib.saber_k_gemm_rowln(ptr(A), K, ptr(W), Kp, ptr(bias), ptr(x), ptr(x), None, ptr(g), ptr(b), 1e-6, ptr(xn), M, N, K, s))
            print(f"M={M:8d} N={N:4d} K={K:5d}  DBG={d:#x}  fused {t:8.1f} us", flush=True)
        lib.saber_k_set_debug(0)
        continue
    t_f = timeit(lambda: lib.saber_k_gemm_rowln(ptr(A), K, ptr(W), Kp, ptr(bias), ptr(x), ptr(x), None, ptr(g), ptr(b), 1e-6, ptr(xn), M, N, K, s))
    t_g = timeit(lambda: lib.saber_k_gemm_ld(ptr(A), K, ptr(W), Kp, 1, ptr(bias), ptr(x), ptr(x), None, M, N, K, 0, s))
    t_l = timeit(lambda: lib.saber_k_layernorm(ptr(x), ptr(g), ptr(b), 1e-6, None, ptr(xn), M, N, 0, s))
    hbm = M * (K * 2 + N * 4 * 2 + N * 2) / 1e9
    print(f"M={M:8d} N={N:4d} K={K:5d}  fused {t_f:8.1f} us ({2.0*M*N*K/t_f/1e6:7.1f} TF/s, {hbm/t_f*1e3:5.2f} TB/s algorithmic)   gemm {t_g:8.1f} + ln {t_l:7.1f} = {t_g+t_l:8.1f} us  [{OP}]", flush=True)
