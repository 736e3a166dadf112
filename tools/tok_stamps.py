"""In-kernel cycle stamps of dec_tokens_kernel through the engine (development): python tools/tok_stamps.py
One decode per segment: SABER_AMD_TOK_STAMP_SEG picks the segment (call count mod 4) that writes its stamps (decoder_tokens.hip)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from saber_amd import _lib
from saber_amd.engine import Engine
lib = _lib.load()
P = 1024
eng = Engine("large", device=0, seed=0, max_images=1, max_prompts=1024)
eng.encode(torch.rand(1024, 1024, device="cuda"))
pts = torch.rand(P, 2, device="cuda") * 1024
for _ in range(2): eng.decode_points(pts, slot=0, multimask=True)
names = ["load Q", "(1) att_out proj + LN", "(2a) MLP + LN3", "(2b) i2t k/v proj, tk, cb", "(2c) folds k, v", "(3) self attention", "(4) t2i q proj + fold", "(5) store Q + heads"]
for seg in range(4):
    st = torch.zeros(600000, dtype=torch.int64, device="cuda")
    os.environ["SABER_AMD_TOK_STAMP_SEG"] = str(seg)
    torch.cuda.synchronize()
    lib.saber_k_set_stamp_buffer(C.c_void_p(st.data_ptr())); eng.decode_points(pts, slot=0, multimask=True); torch.cuda.synchronize(); lib.saber_k_set_stamp_buffer(None)
    s = st[500000:500000 + (P // 4) * 8].view(P // 4, 8).double().cpu()
    print(f"segment S{seg}: s_memtime ticks per workgroup (mean over {P // 4} workgroups), total {s.sum(-1).mean():.0f}")
    for k, n in enumerate(names):
        if s[:, k].mean() > 50: print(f"   {n:28s} {s[:, k].mean():8.0f}")
os.environ.pop("SABER_AMD_TOK_STAMP_SEG", None)
