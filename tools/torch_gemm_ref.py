"""Library yardstick for the Hiera-L GEMM shapes (torch.mm bf16 -> rocBLAS/hipBLASLt), NOT used by the engine: python tools/torch_gemm_ref.py"""
import torch, sys
M0 = int(sys.argv[1]) if len(sys.argv) > 1 else 21
shapes = [(4096 * M0, 1728, 576), (4096 * M0, 576, 576), (4096 * M0, 2304, 576), (4096 * M0, 576, 2304), (16384 * M0, 864, 288), (16384 * M0, 1152, 288),
          (16384 * M0, 288, 1152), (65536 * M0, 432, 144), (65536 * M0, 576, 144), (65536 * M0, 144, 576), (1024 * M0, 3456, 1152), (1024 * M0, 4608, 1152), (1024 * M0, 1152, 4608), (8192, 8192, 8192)]
for M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    for _ in range(3): C = A @ W.T
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): C = A @ W.T
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"torch.mm M={M:8d} N={N:5d} K={K:5d}  {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF/s", flush=True)
