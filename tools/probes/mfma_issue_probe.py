import ctypes as C, subprocess, torch
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "tools/probes/mfma_issue_probe.hip", "-o", "/tmp/mfma_issue_probe.so"])
lib = C.CDLL("/tmp/mfma_issue_probe.so")
p = lambda t: C.c_void_p(t.data_ptr())
iters = 200000
for kind, per_iter_flop in ((16, 32 * 16384.0), (32, 16 * 32768.0)):
    for waves_per_simd in (1, 2, 4):
        blocks = 256 * waves_per_simd
        sink = torch.zeros(blocks * 256, device="cuda")
        row = []
        for nacc in ((1, 2, 4, 8, 16) if kind == 16 else (1, 2, 4, 8)):
            for rep in range(2):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); lib.run(kind, nacc, blocks, iters, p(sink)); b.record(); torch.cuda.synchronize()
            tf = blocks * 4 * iters * per_iter_flop / a.elapsed_time(b) / 1e9
            row.append(f"{nacc} acc: {tf:5.0f} ({tf / 25:3.0f} %)")
        print(f"v_mfma_f32_{kind}x{kind}x{512 // kind}_bf16, {waves_per_simd} wave(s) per SIMD, TFLOP/s by independent accumulators per wave: " + " | ".join(row))
