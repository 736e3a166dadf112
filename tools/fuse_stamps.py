"""In-kernel cycle stamps of the fused dec_i2t_t2i kernel through the engine (development): python tools/fuse_stamps.py [P]
The stamp buffer is shared by every stamp-aware kernel of the decode; the fused kernel of layer 1 + final attention is the last writer."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SABER_AMD_FUSE_I2T_T2I"] = "1"
import torch
from saber_amd import _lib
from saber_amd.engine import Engine
lib = _lib.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = Engine("large", device=0, seed=0, max_images=1, max_prompts=1024)
eng.encode(torch.rand(1024, 1024, device="cuda"))
pts = torch.rand(P, 2, device="cuda") * 1024
for _ in range(2): eng.decode_points(pts, slot=0, multimask=True)
st = torch.zeros(P * 8 * 6, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
lib.saber_k_set_stamp_buffer(C.c_void_p(st.data_ptr())); eng.decode_points(pts, slot=0, multimask=True); torch.cuda.synchronize(); lib.saber_k_set_stamp_buffer(None)
NW = int(os.environ.get("FUSE_WAVES", "4"))
s = st[:P * NW * 4].view(P, NW, 4).double().cpu() / 256.0
names = ["i2t scores + softmax + P write", "vmcnt wait + barrier", "DMA issue + finish_tile + t2i step (every 2nd)", "i2t PV + residual + stats"]
print("cycles per 16-row tile (s_memtime ticks), mean over workgroups; per wave:")
for k, n in enumerate(names):
    print(f"  {n:48s} " + " ".join(f"{s[:, w, k].mean():7.0f}" for w in range(NW)) + f"   | all {s[:, :, k].mean():7.0f}")
print("  total per tile", s.sum(-1).mean().item())
