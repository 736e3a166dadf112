"""Which kernel the vendor library runs for the stage-3 fc1 / fc2 shapes (run under rocprofv3 --kernel-trace: the trace carries the
kernel name, workgroup size, LDS bytes, VGPR / AccVGPR counts)."""
import torch

for (M, N, K) in ((86016, 2304, 576), (86016, 576, 2304), (86016, 1728, 576)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        torch.nn.functional.linear(A, W, b)
    torch.cuda.synchronize()
