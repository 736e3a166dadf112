"""HBM yardstick on this box: what a plain streaming kernel reaches (library copy / fill / reduction through PyTorch-ROCm), to read
the PMC-derived GB/s of the HBM-bound kernels (dec_i2t, dec_t2i, mask_embed_src, layernorm) against something measured."""
import torch
n = 1 << 30     # 4 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
s = t(lambda: b.copy_(a)); print(f"copy   (4 GiB R + 4 GiB W): {8 * n / s / 1e12:.2f} TB/s")
s = t(lambda: a.sum());    print(f"sum    (4 GiB R)          : {4 * n / s / 1e12:.2f} TB/s")
s = t(lambda: b.fill_(1.0)); print(f"fill   (4 GiB W)          : {4 * n / s / 1e12:.2f} TB/s")
h = a.view(torch.int16)[: n]  # bf16-sized elements
s = t(lambda: torch.add(a, 1.0, out=b)); print(f"add    (4 GiB R + 4 GiB W): {8 * n / s / 1e12:.2f} TB/s")
