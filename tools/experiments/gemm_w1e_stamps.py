"""In-kernel cycle stamps of gemm_w1e_kernel (development): python tools/gemm_w1e_stamps.py M N K [act]
Per wave, summed over a workgroup's tiles and divided by its tile count: K-step 0 (conversion of the finished tile + direct stores), the
K-steps that carry the parked stores (+ the padding step), the plain K-steps, the drain of the last tile."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
lib = _lib.load(); assert lib.saber_k_init(0) == 0
def ptr(t): return C.c_void_p(t.data_ptr())
M, N, K = [int(x) for x in sys.argv[1:4]]; act = int(sys.argv[4]) if len(sys.argv) > 4 else 0
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
lib.saber_k_set_debug(128)
call = lambda: lib.saber_k_gemm_ld(ptr(A), K, ptr(W), K, 1, ptr(bias), None, None, ptr(out), M, N, K, act, None)
for _ in range(3): call()
st = torch.zeros(256 * 4 * 4, dtype=torch.int64, device="cuda")
lib.saber_k_set_stamp_buffer(ptr(st)); call(); torch.cuda.synchronize(); lib.saber_k_set_stamp_buffer(None)
tiles = ((M + 255) // 256) * ((N + 255) // 256); nk = K // 32
s = st.view(256, 4, 4).double().cpu()
per = tiles / 256.0
print(f"{os.environ.get('SABER_AMD_LIB', 'default lib')}: tiles {tiles} ({per:.2f} per workgroup), {nk} K-steps each; s_memtime ticks (100 MHz: 10 ns) per tile, mean over workgroups, per wave")
for k, n in enumerate(["K-step 0 (conversion)", "head steps (parked stores)", "plain steps", "drain (per workgroup)"]):
    d = 1.0 if k == 3 else per
    print(f"  {n:28s} " + " ".join(f"{s[:, w, k].mean() / d:8.1f}" for w in range(4)))
print(f"  total per workgroup {s.sum(-1).mean().item():.0f} ticks = {s.sum(-1).mean().item() * 0.01:.1f} us")
