"""In-kernel cycle stamps of the MXFP8 GEMM (development): python tools/gemm_fp8_stamps.py M N K form(bf16|mx|f32)
Cycles (s_memtime) per K-step and phase, mean over workgroups, one column per wave."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
M, N, K = [int(x) for x in sys.argv[1:4]]; form = sys.argv[4] if len(sys.argv) > 4 else "bf16"
Kp = (K + 127) // 128 * 128; Mp = (M + 767) // 768 * 768; Np = (N + 191) // 192 * 192
A8 = torch.randint(0, 120, (M, Kp), dtype=torch.uint8, device="cuda"); W8 = torch.randint(0, 120, (N, Kp), dtype=torch.uint8, device="cuda")
sa = torch.full((Kp // 128, Mp, 4), 120, dtype=torch.uint8, device="cuda"); sw = torch.full((Kp // 128, Np, 4), 120, dtype=torch.uint8, device="cuda")
bias = torch.zeros(N, device="cuda"); outb = torch.empty(M, N, dtype=torch.uint16, device="cuda")
outf = torch.empty(M, N, device="cuda"); res = torch.zeros(M, N, device="cuda")
o8 = torch.empty(M, N, dtype=torch.uint8, device="cuda"); os_ = torch.empty((N + 127) // 128, Mp, 4, dtype=torch.uint8, device="cuda")
args = (p(A8), Kp, p(sa), Mp, p(W8), Kp, p(sw), Np, p(bias))
if form == "bf16": call = lambda: lib.saber_k_gemm_mx(*args, None, None, p(outb), None, None, 0, N, M, N, Kp, 0, None)
elif form == "mx": call = lambda: lib.saber_k_gemm_mx(*args, None, None, None, p(o8), p(os_), Mp, N, M, N, Kp, 1, None)
else: call = lambda: lib.saber_k_gemm_mx(*args, p(res), p(outf), None, None, None, 0, N, M, N, Kp, 0, None)
for _ in range(3): assert call() == 0, lib.saber_k_last_error()
ver = int(os.environ.get("SABER_AMD_MX_KERNEL", "4"))
TM, TN, NW, NWG = {2: (256, 192, 8, 256), 3: (192, 192, 8, 256), 4: (192, 96, 4, 512)}[ver]
st = torch.zeros(NWG * NW * 8, dtype=torch.int64, device="cuda")
lib.saber_k_set_stamp_buffer(p(st)); call(); torch.cuda.synchronize(); lib.saber_k_set_stamp_buffer(None)
tiles = ((M + TM - 1) // TM) * ((N + TN - 1) // TN); nk = Kp // 128
s = st.view(NWG, NW, 8).double().cpu()
steps = tiles / float(NWG) * nk
names = {2: ["barrier", "LDS-DMA issue", "fragment read issue", "first fragments back", "products + later fragments", "transfer wait", "epilogue (per tile)"],
         3: ["barrier", "fragment read issue", "first fragments back", "products + LDS-DMA issue", "transfer wait", "-", "epilogue (per tile)"],
         4: ["barrier", "LDS-DMA issue", "fragment read issue", "first fragments back", "products + later fragments", "transfer wait", "epilogue (per tile)"]}[ver]
print(f"kernel {ver}: {M}x{N}x{K} out={form}: {tiles} tiles of {TM}x{TN} ({tiles / NWG:.2f} per workgroup), {nk} K-steps each; s_memtime ticks per K-step (epilogue: per tile), per wave:")
for k, n in enumerate(names):
    d = tiles / float(NWG) if k == 6 else steps
    print(f"  {n:28s} " + " ".join(f"{s[:, w, k].mean() / d:7.0f}" for w in range(NW)))
print("  total per workgroup", s.sum(-1).mean().item(), " per K-step incl. epilogue share", s.sum(-1).mean().item() / steps)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): call()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 100
print(f"  {us:.1f} us per launch = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s; us per K-step {us / steps:.3f}")
