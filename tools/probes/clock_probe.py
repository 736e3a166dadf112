import ctypes as C, subprocess, torch
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "tools/probes/clock_probe.hip", "-o", "/tmp/clock_probe.so"])
lib = C.CDLL("/tmp/clock_probe.so")
p = lambda t: C.c_void_p(t.data_ptr())
for name, mode, blocks, iters in (("one workgroup, VALU only", 0, 1, 2000000), ("VALU on every CU (2 workgroups per CU)", 0, 512, 2000000),
                                 ("bf16 MFMA back to back on every SIMD (2 waves per SIMD)", 1, 512, 400000), ("the same, one wave per SIMD", 1, 256, 400000)):
    out = torch.zeros(2 * blocks, dtype=torch.int64, device="cuda"); sink = torch.zeros(blocks * 256, device="cuda")
    for rep in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); lib.clock_run(mode, blocks, iters, p(out), p(sink)); b.record(); torch.cuda.synchronize()
    o = out.view(blocks, 2).double().cpu()
    ghz = (o[:, 0] / o[:, 1] * 0.1)
    ms = a.elapsed_time(b)
    extra = ""
    if mode == 1:
        fl = blocks * 4 * iters * 16 * 16384.0
        extra = f", {fl / ms / 1e9:.0f} TFLOP/s achieved ({fl / ms / 1e9 / 2500 * 100:.0f} % of the 2.5 PF nominal peak)"
    print(f"{name}: shader clock {ghz.mean():.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f}) over {ms:.1f} ms{extra}")
