"""Co-residency experiment, cheap form: how much of their speed do the decoder's HBM-bound kernels and the encoder keep
on a stream restricted to N of the 256 CUs (hipExtStreamCreateWithCUMask)?  If i2t / t2i / mask_embed_src held >= 85 % of their bandwidth on
half the chip, one slice's decoder could run beside another slice's encoder on disjoint CUs.  python tools/cu_mask_bench.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from saber_amd.engine import Engine
P = 1024
eng = Engine("large", device=0, seed=0, max_images=21, max_prompts=1024)
lib = eng.lib
eng.set_graphs(False)
img = torch.rand(1024, 1024, device="cuda")
pts = torch.rand(P, 2, device="cuda") * 1024
crops = [[0, 0, 1024, 1024]] * 21


def measure(stream, label):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        eng.encode(img, crops)
        low, iou, _ = eng.decode_points(pts, slot=0, multimask=True)
        mi = torch.clamp(low[:, 0], -32, 32).contiguous()
        for _ in range(2):
            eng.decode_points(pts, slot=0, multimask=False, mask_input=mi)
        torch.cuda.synchronize()
        eng.profile_begin()
        for _ in range(3):
            eng.decode_points(pts, slot=0, multimask=False, mask_input=mi)
        prof = eng.profile_end()
        dec = {k: v["ms"] / 3 for k, v in prof.items() if v["ms"] > 0}
        torch.cuda.synchronize()
        eng.profile_begin()
        eng.encode(img, crops)
        prof = eng.profile_end()
        enc = {k: v["ms"] for k, v in prof.items() if v["ms"] > 0}
    print(f"{label:28s} m2m decode of {P} prompts: " + " ".join(f"{k} {v:.3f}" for k, v in dec.items()) + f" | sum {sum(dec.values()):.2f} ms")
    print(f"{'':28s} 21-crop encoder pass:      " + " ".join(f"{k} {v:.2f}" for k, v in enc.items()) + f" | sum {sum(enc.values()):.2f} ms")
    return dec, enc


base = measure(None, "all 256 CUs (plain stream)")
for n in (192, 128, 96, 64):
    h = C.c_void_p()
    assert lib.saber_k_stream_create_cu_range(0, n, C.byref(h)) == 0, lib.saber_k_last_error()
    s = torch.cuda.ExternalStream(h.value)
    d, e = measure(s, f"{n} CUs (mask 0..{n - 1})")
    print(f"{'':28s} kept of full-chip speed: " + " ".join(f"{k} {base[0][k] / d[k] * 100:.0f}%" for k in d if k in base[0]) +
          " | encoder " + " ".join(f"{k} {base[1][k] / e[k] * 100:.0f}%" for k in ("gemm_bf16", "hiera_attention") if k in e))
    torch.cuda.synchronize()
    lib.saber_k_stream_destroy(h)
eng.close()
