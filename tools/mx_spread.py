"""Realisation spread of the MXFP8 emulation (oracle/sam2_bf16_emul.py, mx=True) on the host: PYTHONPATH=. python tools/mx_spread.py (1 min).
The constants MX_SPREAD of tests/test_gpu_fp8.py come from here."""
import time, numpy as np, torch, torch.nn.functional as F
from oracle import sam2_ref, sam2_bf16_emul as E, fp8_ref
from saber_amd.model_config import get_config
from saber_amd.weights import seeded_weights
cfg = get_config("large"); Wn = seeded_weights(cfg, 0)
W0 = sam2_ref.to_torch(Wn); Wq = sam2_ref.to_torch(fp8_ref.mx_quantise_encoder_weights(Wn, cfg))
rng = np.random.default_rng(7); img = rng.uniform(0, 1, (1024, 1024)).astype(np.float32)
pix = sam2_ref.sam2_transforms(np.repeat(img[..., None], 3, 2))
rel = lambda a, b: ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
t = time.time(); f1 = E.encode_image_emul(Wq, cfg, pix, mx=True); print("emul mx", time.time() - t)
o1, o2 = E.lin, E.lin_mx
def lin64(x, w, b):
    y = F.linear(E.bf(x).double(), E.bf(w).double()).float(); return y if b is None else y + b
def linmx64(x, w, b):
    y = F.linear(x.double(), w.double()).float(); return y if b is None else y + b
E.lin, E.lin_mx = lin64, linmx64
t = time.time(); f2 = E.encode_image_emul(Wq, cfg, pix, mx=True); print("emul mx fp64", time.time() - t)
E.lin, E.lin_mx = o1, o2
print("MX realisation spread:", {k: rel(f2[k], f1[k]) for k in f1})
fb = E.encode_image_emul(W0, cfg, pix)
with torch.no_grad(): f0 = sam2_ref.encode_image(W0, cfg, pix); fq = sam2_ref.encode_image(Wq, cfg, pix)
print("bf16 emul vs fp32:", {k: rel(fb[k], f0[k][0:1] if isinstance(f0[k], torch.Tensor) else f0[k]) for k in fb})
print("mx emul vs fp32:", {k: rel(f1[k], f0[k]) for k in f1})
print("fp32 oracle with MX weights vs fp32:", {k: rel(fq[k], f0[k]) for k in f1})
