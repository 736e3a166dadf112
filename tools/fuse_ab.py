"""A/B of an opt-in decoder kernel on the decode of P prompts: python tools/fuse_ab.py [P] [ENV]  (ENV = SABER_AMD_FUSE_I2T_T2I (default) or
SABER_AMD_T2I_W1 / SABER_AMD_I2T_W1 (default on: A/B is =0 against unset); per-class profile, ms per decode)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd.engine import Engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ENV = sys.argv[2] if len(sys.argv) > 2 else "SABER_AMD_FUSE_I2T_T2I"
eng = Engine("large", device=0, seed=0, max_images=1, max_prompts=1024)
eng.encode(torch.rand(1024, 1024, device="cuda"))
pts = torch.rand(P, 2, device="cuda") * 1024
low, iou, _ = eng.decode_points(pts, slot=0, multimask=True)
mi = torch.clamp(low[:, 0], -32, 32).contiguous()
for fuse in (0, 1, 0, 1):
    if ENV in ("SABER_AMD_T2I_W1", "SABER_AMD_I2T_W1"):          # default ON: A/B is "0" against unset
        if fuse: os.environ.pop(ENV, None)
        else: os.environ[ENV] = "0"
    elif fuse: os.environ[ENV] = "1"
    else: os.environ.pop(ENV, None)
    for name, kw in (("first", dict(multimask=True)), ("m2m", dict(multimask=False, mask_input=mi))):
        for _ in range(2): eng.decode_points(pts, slot=0, **kw)
        torch.cuda.synchronize()
        eng.profile_begin()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): eng.decode_points(pts, slot=0, **kw)
        e.record(); torch.cuda.synchronize()
        prof = eng.profile_end()
        print(f"{ENV}={fuse} {name}: P={P} {s.elapsed_time(e) / 5:.3f} ms/decode ", {k: round(v['ms'] / 5, 3) for k, v in prof.items() if v['ms'] > 0}, flush=True)
