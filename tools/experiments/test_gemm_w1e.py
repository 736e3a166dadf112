"""Parity test of the parked experiment tools/experiments/gemm_w1e.hip (not collected by the suites: the kernel is not in the library).
To run it: revive the kernel as tools/experiments/README.md says, copy this file next to tests/test_gpu_kernels.py."""
import torch
import torch.nn.functional as F
import pytest

from test_gpu_kernels import bf, from_bf, kcall, ptr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("op", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K,act", [(256 * 9 + 16, 1728, 576, 0), (4096, 2304, 576, 1), (256 * 5 + 48, 3456, 1152, 0), (256 * 3, 4608, 1152, 1), (528, 64, 256, 1)])
def test_gemm_w1e_deferred_epilogue(gpu_lib, M, N, K, act, op):
    """csrc/gemm_w1e.hip (round 5): four waves of 128 x 128 whose epilogue runs under the NEXT tile's K loop - the kernel of attn.qkv and
    mlp.layers.0 in stages 2-3.  Forced for small shapes through the debug flag; ragged M (a last tile of 16 / 48 rows), N that is not a
    multiple of 256 (1728 = 6.75 tiles: suppressed column stores), several tiles per workgroup (the deferred stores of tile T leave during
    tile T + 1, the last tile is drained after the loop), one tile per workgroup (only the drain), both operand types.  Reference: fp64."""
    import numpy as np
    g = torch.Generator().manual_seed(M + N + K + act)
    A32 = torch.randn(M, K, generator=g)
    W32 = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    if op == "bf16":
        (A, Ad), (W, Wd) = bf(A32), bf(W32)
    else:
        A, W = A32.half().float(), W32.half().float()
        Ad, Wd = A32.half().view(torch.int16).cuda(), W32.half().view(torch.int16).cuda()
    ref = A.double() @ W.double().T + bias.double()
    ref = F.gelu(ref) if act == 1 else ref
    out_b = torch.full((M + 8, N), 0x7fff, dtype=torch.int16, device="cuda")      # 8 guard rows behind the matrix: nothing may be written there
    bias_d = bias.cuda()
    prev = gpu_lib.saber_k_set_operand_type(1 if op == "f16" else 0)
    gpu_lib.saber_k_set_debug(128)
    try:
        kcall(gpu_lib, gpu_lib.saber_k_gemm_ld(ptr(Ad), K, ptr(Wd), K, 1, ptr(bias_d), None, None, ptr(out_b), M, N, K, act, None))
        torch.cuda.synchronize()
        # the SAME problem on the staggered 8-wave kernel (flag 131072): identical 16-bit results (same MFMA, same K order, same epilogue arithmetic)
        out_o = torch.zeros(M, N, dtype=torch.int16, device="cuda")
        gpu_lib.saber_k_set_debug(128 | 131072)
        kcall(gpu_lib, gpu_lib.saber_k_gemm_ld(ptr(Ad), K, ptr(Wd), K, 1, ptr(bias_d), None, None, ptr(out_o), M, N, K, act, None))
        torch.cuda.synchronize()
    finally:
        gpu_lib.saber_k_set_debug(0)
        gpu_lib.saber_k_set_operand_type(prev)
    assert (out_b[M:] == 0x7fff).all()
    got = from_bf(out_b[:M]) if op == "bf16" else out_b[:M].cpu().view(torch.float16).float()
    scale = ref.abs().max().item() + 1e-6
    errb = (got.double() - ref).abs().max().item() / scale
    assert errb < (5e-3 if op == "bf16" else 7e-4), errb
    assert torch.equal(out_b[:M], out_o)




# ---- compile-time check (CPU): was tests/test_abi.py::test_no_spills_in_gemm_w1e while the kernel was in the build
import os
import re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_no_spills_in_gemm_w1e():
    """csrc/gemm_w1e.hip is a one-wave-per-SIMD kernel whose 256 accumulators are PHYSICAL AccVGPRs named in its instruction strings (the
    compiler only learns from one clobber list that the AccVGPR file is taken).  Two things must hold for that to be sound, and both are
    properties of the compiler's output, so they are checked here on the generated assembly of both operand types: (1) nothing is
    spilled (a reload inside the K loop is a vector-memory load whose wait drains the ring of LDS-DMA transfers; a spill INTO an AccVGPR
    would overwrite an accumulator), (2) no AccVGPR instruction exists outside the kernel's own inline-asm blocks."""
    import shutil
    import subprocess
    import tempfile
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "saber_amd", "csrc")
    for flags in (["-DSABER_OP_NS=op_bf16"], ["-DSABER_OP_NS=op_f16", "-DSABER_OP_F16=1"]):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "w1e.s")
            r = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-unused-value", *flags,
                                '-DSABER_OP_SRC="gemm_w1e.hip"', "-S", "--cuda-device-only", "op_wrap.hip", "-o", out,
                                "-Rpass-analysis=kernel-resource-usage"], cwd=csrc  # (with gemm_w1e.hip copied back into csrc/), capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            remarks = [ln for ln in r.stderr.splitlines() if "VGPRs Spill" in ln or "ScratchSize" in ln]
            assert len(remarks) == 8, r.stderr[-2000:]          # four kernels (GELU x stamps) x two remarks
            assert all(ln.rstrip().split(":")[-1].split()[0] == "0" for ln in remarks), remarks
            assert sum("AGPRs: 256" in ln for ln in r.stderr.splitlines()) == 4
            inside, stray = False, []
            for ln in open(out):
                if "ASMSTART" in ln:
                    inside = True
                elif "ASMEND" in ln:
                    inside = False
                elif not inside and ("v_accvgpr" in ln or re.search(r"\ba\[?\d", ln.split(";")[0])) and not ln.lstrip().startswith((".", ";")):
                    stray.append(ln.strip())
            assert not stray, stray[:5]
