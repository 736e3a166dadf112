"""End-to-end single-GPU run of the z-loop on a synthetic tomogram (BASELINE configs[2] volume recipe):
python tools/volume_e2e.py [Z]   ->   slices/s including label-plane painting and the 3-D CC stitch on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from saber_amd.engine import Engine, make_amg_params
from saber_amd.model_config import get_config
from saber_amd.weights import seeded_weights
from saber_amd.segmenters.slice_driver import segment_slice_to_plane, segment_volume_sharded
from oracle import saber_ref   # synthetic input recipe only
Z = int(sys.argv[1]) if len(sys.argv) > 1 else 64
FMT = sys.argv[2] if len(sys.argv) > 2 else "bf16"          # "mxfp8": MXFP8 stage-2/3 block GEMMs on the fp8 MFMA (BASELINE configs[4]); "fp8": the e4m3 storage format
W_ = seeded_weights(get_config("large"), 0)
eng = Engine("large", device=0, weights=W_, max_images=21, max_prompts=1024, weight_format=FMT)
eng2 = Engine("large", device=0, weights=W_, max_images=21, max_prompts=1024, weight_format=FMT)   # two slices in flight per GPU, as slice_by_slice_device does
params = make_amg_params({})
vol = saber_ref.synthetic_volume(seed=1, depth=Z)
dev = torch.from_numpy(vol).cuda()
segment_slice_to_plane(eng, dev[0], params, min_mask_area=50)
torch.cuda.synchronize()
t0 = time.perf_counter()
fns = [lambda z, e=e: segment_slice_to_plane(e, dev[z], params, min_mask_area=50)[0] for e in (eng, eng2)]
out = segment_volume_sharded(vol, fns, stitch=True, engine=eng)
dt = time.perf_counter() - t0
print(f"Z={Z} weights={FMT} graphs={eng.graph_stats()}: {dt:.2f} s end to end = {Z / dt:.2f} slices/s (label volume {out.shape} {out.dtype}, {int(out.max())} labels)")
t0 = time.perf_counter()
sm = segment_volume_sharded(vol, fns, stitch=True, engine=eng, smooth_scale=0.05)      # + the post step of segment_tomogram_core on the device
dt = time.perf_counter() - t0
print(f"Z={Z}: {dt:.2f} s with the per-label Gaussian smoothing = {Z / dt:.2f} slices/s ({sm.dtype}, labels {np.unique(sm).size - 1})")
