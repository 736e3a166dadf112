"""Time engine.decode_points for a batch of prompts (first pass and m2m pass), with the per-class profile: python tools/decode_bench.py [P]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from saber_amd.engine import Engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = Engine("large", device=0, seed=0, max_images=1, max_prompts=1024)
img = torch.rand(1024, 1024, device="cuda")
eng.encode(img)
pts = torch.rand(P, 2, device="cuda") * 1024
low, iou, _ = eng.decode_points(pts, slot=0, multimask=True)
mi = torch.clamp(low[:, 0], -32, 32).contiguous()
for name, kw in (("first", dict(multimask=True)), ("m2m", dict(multimask=False, mask_input=mi))):
    for _ in range(2): eng.decode_points(pts, slot=0, **kw)
    torch.cuda.synchronize()
    eng.profile_begin()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): eng.decode_points(pts, slot=0, **kw)
    e.record(); torch.cuda.synchronize()
    prof = eng.profile_end()
    print(f"{name}: P={P} {s.elapsed_time(e) / 3:.3f} ms/decode ", {k: round(v['ms'] / 3, 3) for k, v in prof.items() if v['ms'] > 0})
