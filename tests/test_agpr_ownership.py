"""CPU (hipcc cross-compiles): the kernels that name their AccVGPRs in the instruction strings own the whole AccVGPR file.

`dec_t2i_w1_kernel` keeps its 64 x 256 partial sums in a[0..255], `dec_i2t_w1_kernel` the prompt's folded K / V operands; the compiler is
told with a clobber list, but that is a statement about ONE point of the program: nothing stops the register allocator from parking a
VGPR it has no room for in a "free" AccVGPR (it prefers that to scratch), or from placing the result of an MFMA of its own there - both
were seen while these kernels were written (wrong results, no diagnostics).  So: compile the translation unit to assembly and require that
every AccVGPR reference of these kernels sits inside an inline-asm block (;;#ASMSTART .. ;;#ASMEND), in both operand-type namespaces' shared
source (the bf16 instantiation is checked; the fp16 one differs in the MFMA mnemonic only)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "saber_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_compiler_stays_out_of_the_accvgprs(tmp_path):
    out = tmp_path / "decoder_fused.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-unused-value", "-DSABER_OP_NS=op_bf16",
           "-DSABER_OP_SRC=\"decoder_fused.hip\"", "-S", "--cuda-device-only", "op_wrap.hip", "-o", str(out)]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    text = out.read_text()
    checked = 0
    for m in re.finditer(r"^(_ZN7op_bf16\d+(dec_t2i_w1_kernel|dec_i2t_w1_kernel)\w*):", text, re.M):
        body = text[m.start():text.index(".Lfunc_end", m.start())]
        inside, stray = False, []
        for line in body.split("\n"):
            if ";;#ASMSTART" in line:
                inside = True
            elif ";;#ASMEND" in line:
                inside = False
            elif not inside:
                code = line.split(";")[0]
                if "v_accvgpr" in code or re.search(r"\ba\[?\d", code):
                    stray.append(line.strip())
        assert not stray, (m.group(1), stray[:5])
        assert "scratch_" not in body or body.count("scratch_") <= 8, "spills inside a one-wave-per-SIMD kernel"
        checked += 1
    assert checked == 6, checked            # t2i_w1 <STAMPS, SHARED> x 4, i2t_w1 <SHARED> x 2
