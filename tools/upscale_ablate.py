"""What dec_upscale_kernel's time is made of (development).  Needs the development build of the library:
    make -C saber_amd/csrc -j8 EXTRA=-DUP_DEV=1 BUILD=build_dev LIB=../libsaber_amd_dev.so
    SABER_AMD_LIB=saber_amd/libsaber_amd_dev.so python tools/upscale_ablate.py [P]
One 1 024-prompt multimask decode on a Hiera-L handle; the kernel class time of `decoder_upscale` (HIP events around the launch) with single
pieces of the prompt loop switched off at run time (results are garbage, timing only), then the per-phase s_memtime stamps of the full kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from saber_amd.engine import Engine
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = Engine("large", device=0, seed=0, max_images=1, max_prompts=1024)
lib = eng.lib
eng.encode(torch.rand(1024, 1024, device="cuda"))
pts = torch.rand(P, 2, device="cuda") * 1024
FLAGS = {"full kernel": 0, "no GELU (both phases)": 1, "no W1 fragment reads (one address)": 2, "no W2 fragment reads": 4, "no stores": 8, "no hypernetwork product": 16,
         "no X loads": 32, "no LayerNorm lane reductions": 64, "no phase-A MFMAs": 128, "no phase-B MFMAs": 256, "no MFMAs at all": 384, "no W1/W2 reads, no MFMAs": 2 | 4 | 384,
         "no GELU, no hyper, no LN": 1 | 16 | 64, "only loads + stores (everything else off)": 1 | 2 | 4 | 16 | 64 | 384}


def run(flag, reps=3):
    lib.saber_k_set_debug(flag << 8)
    try:
        for _ in range(2):
            eng.decode_points(pts, slot=0, multimask=True)
        torch.cuda.synchronize()
        eng.profile_begin()
        for _ in range(reps):
            eng.decode_points(pts, slot=0, multimask=True)
        prof = eng.profile_end()
    finally:
        lib.saber_k_set_debug(0)
    return prof["decoder_upscale"]["ms"] / reps


if os.environ.get("UP_ONLY_FULL"):      # a compile-time variant of the library (tools/upscale_ablate.sh): one timing, no run-time switches, no stamps
    ms = run(0, reps=5)
    print(f"decoder_upscale {ms * 1e3:8.1f} us per {P}-prompt launch = {ms * 1e3 / (P / 2):6.3f} us per prompt and workgroup")
    eng.close()
    sys.exit(0)
base = None
for name, f in FLAGS.items():
    ms = run(f)
    base = base or ms
    print(f"{name:45s} {ms * 1e3:8.1f} us per {P}-prompt launch  ({ms / base * 100:5.1f} %)  = {ms * 1e3 / (P / 2):6.3f} us per prompt and workgroup")
# per-phase stamps of the full kernel (s_memtime ticks = 100 MHz constant clock on gfx950: 10 ns each)
NW = 4 * int(os.environ.get("UP_NTG", "3"))      # waves per workgroup of the build under test
st = torch.zeros(256 * NW * 5, dtype=torch.int64, device="cuda")
for _ in range(2):
    eng.decode_points(pts, slot=0, multimask=True)
lib.saber_k_set_stamp_buffer(C.c_void_p(st.data_ptr()))
eng.decode_points(pts, slot=0, multimask=True)
torch.cuda.synchronize()
lib.saber_k_set_stamp_buffer(None)
s = st.view(256, NW, 5).double().cpu() / (P * 86.0 / 256.0)       # per (tile, prompt) unit
names = ["phase A: 32 ds_read_b128 + 32 MFMA", "epilogue A: LayerNorm, 16 GELU, pack", "phase B MFMAs (2 x (8 ds_read + 8 MFMA))", "epilogue B: 32 GELU, hyper product, transposes", "stores + loop"]
print("s_memtime ticks per prompt, mean over blocks and waves (a wave's own elapsed time between stamps, so it contains the time its SIMD partner held the issue port):")
tot = s.sum(-1).mean().item()
for k, n in enumerate(names):
    print(f"  {n:50s} {s[:, :, k].mean().item():8.2f}  ({s[:, :, k].mean().item() / tot * 100:5.1f} %)")
print(f"  total {tot:.2f} ticks per prompt")
eng.close()
