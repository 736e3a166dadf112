"""Micro-benchmark of the bf16 GEMM kernel on the Hiera-L shapes (GPU box only)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
def ptr(t): return C.c_void_p(t.data_ptr())
ACT = int(os.environ.get('ACT', '0'))
OP = os.environ.get('OP', 'bf16')          # 16-bit operand type: bf16 | f16 (the same kernels compiled for the other type, saber_k_set_operand_type)
DT = {'bf16': torch.bfloat16, 'f16': torch.float16}[OP]
lib.saber_k_set_operand_type(1 if OP == 'f16' else 0)
DATA = os.environ.get('DATA', 'randn')     # randn | zeros | gelu | tiny: operand bit activity / fp16 subnormals (MI355X_MICROARCH.md "DVFS give-back")
def rnd(*s):
    if DATA == 'zeros': return torch.zeros(*s, device="cuda")
    x = torch.randn(*s, device="cuda")
    if DATA == 'gelu': return torch.nn.functional.gelu(2.0 * x)      # what fc2 reads: half of the values tiny (fp16 subnormals below 6.1e-5)
    if DATA == 'tiny': return x * 1e-6                               # every value an fp16 subnormal (a normal bf16)
    return x
RES = int(os.environ.get('RES', '0'))   # 1: fp32 output with fp32 residual (proj / fc2 of a Hiera block)
if os.environ.get('DBG'): lib.saber_k_set_debug(int(os.environ['DBG'], 0))
M0 = int(os.environ.get("M0", "8"))
shapes = [(4096 * M0, 1728, 576), (4096 * M0, 576, 576), (4096 * M0, 2304, 576), (4096 * M0, 576, 2304), (16384 * M0, 864, 288), (16384 * M0, 1152, 288), (16384 * M0, 288, 1152),
          (65536 * M0, 432, 144), (65536 * M0, 576, 144), (65536 * M0, 144, 576), (1024 * M0, 3456, 1152), (1024 * M0, 4608, 1152), (1024 * M0, 1152, 4608), (8192, 256, 256), (4096, 4096, 4096), (8192, 8192, 8192)]
dbgs = [int(v, 0) for v in os.environ['DBGS'].split(',')] if os.environ.get('DBGS') else [None]     # sweep of development flags in one process
if os.environ.get('SHAPES'): shapes = [shapes[int(i)] for i in os.environ['SHAPES'].split(',')]
if os.environ.get('SHAPE_LIST'): shapes = [tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPE_LIST'].split(';')]     # "M,N,K;M,N,K"
for M, N, K in shapes:
    Kp = (K + 63) // 64 * 64   # weights as the engine uploads them: rows zero-padded to a multiple of 64
    A = rnd(M, K).to(DT); W = torch.zeros(N, Kp, device="cuda", dtype=DT)
    W[:, :K] = (rnd(N, K) / K ** 0.5).to(DT)
    bias = rnd(N); out = torch.empty(M, N, device="cuda", dtype=DT)
    outf = torch.empty(M, N, device="cuda") if RES else None; res = torch.randn(M, N, device="cuda") if RES else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run(): lib.saber_k_gemm_ld(ptr(A), K, ptr(W), Kp, 1, ptr(bias), ptr(res) if RES else None, ptr(outf) if RES else None, None if RES else ptr(out), M, N, K, ACT, st)
    for d in dbgs:
        if d is not None: lib.saber_k_set_debug(d)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"M={M:7d} N={N:5d} K={K:5d}  {'' if d is None else f'DBG={d:#x}  '}{ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF/s  [{OP}, {DATA}]", flush=True)
