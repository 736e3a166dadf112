#!/usr/bin/env python3
"""bench.py - EM slices/sec (1024^2, Hiera-L) on N MI355X, plus the roofline of the dominant kernel and a
CPU baseline timed beside it.

A "step" is one pass of the hot path over one synthetic 1024x1024 uint16 EM slice already resident in HBM:
K0 prepare (prep.prepare) -> automatic mask generation with SABER's cfgAMG defaults (crop_n_layers=2: 21 crops,
3072 grid prompts + 9216 m2m refinements) -> host filter/sort of the reference's
saber2D._apply_classifier (min area, duplicate removal, ascending-area sort) -> uint16 label plane
(propagation.py:185-186).  That is BASELINE.json configs[1] ("single 1024x1024 slice, Hiera-L bf16,
automatic-mask-generator (grid prompts) on 1 MI355X").  Weights are seeded synthetic tensors of the Hiera-L
architecture (no checkpoint is available offline).

N > 1: one process per GPU.  Under a launcher (torchrun: RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process IS one rank;
run plainly (`python bench.py --gpus N`, WORLD_SIZE unset) it starts its N ranks itself BEFORE anything touches a GPU and relays rank 0's
JSON line, as the reference's GPUPool starts one worker per GPU (saber/utils/parallelization.py:137-151, 339-343).  The N > 1 workload is
the north star's: ONE 512-slice tomogram (BASELINE configs[3]) z-sharded over the ranks, value = 512 / end-to-end seconds INCLUDING the
RCCL all-gather of the label planes and the 3-D connected-components stitch (strong scaling), the figure without the stitch beside it
(SURVEY.md 8d "Config 4").  `--weak` keeps the K-private-slices-per-rank form of rounds 1-4 (one all-gather of the K planes per rank inside
the timed region); it is also reported as the extra key `weak_scaling` of the N > 1 line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--crop-n-layers", type=int, default=2, help="cfgAMG.crop_n_layers (SABER default 2)")
    ap.add_argument("--npoints", type=int, default=32)
    ap.add_argument("--max-images", type=int, default=21, help="crops encoded per batched pass (21 = every crop of the default 1+4+16 AMG pyramid)")
    ap.add_argument("--max-prompts", type=int, default=1024, help="prompts decoded per batched pass")
    ap.add_argument("--workers", type=int, default=2, help="engine handles per GPU, each on its own thread and HIP stream, slices dealt round-robin "
                    "(the reference's GPUPool runs one thread per GPU; kernels of two slices in flight fill each other's idle issue slots)")
    ap.add_argument("--dtype", choices=("bf16", "fp16", "fp8", "mxfp8"), default="bf16", help="fp16: the same kernels compiled for IEEE half operands (10 mantissa bits = the TF32 "
                    "arithmetic the reference enables, 1e-3 from the fp32 oracle); fp8: e4m3 weights (per-row power-of-two scales) for the stage-2/3 block GEMMs, "
                    "bf16 activations and MFMA operands, fp32 accumulate (BASELINE configs[4]); SEPARATE lines, never the headline (BASELINE configs[1] says bf16)")
    ap.add_argument("--no-alt-dtypes", action="store_true", help="skip the extra keys `alt_dtypes` (the fp16 and mxfp8 lines measured after the headline on fresh handles)")
    ap.add_argument("--volume", type=int, default=0, metavar="Z", help="STRONG scaling of ONE tomogram of Z slices (BASELINE configs[2] / configs[3]: 64 / 512): "
                    "contiguous z-chunks per rank, one RCCL all-gather of the label planes, 3-D connected components on the device; value = Z / end-to-end seconds "
                    "INCLUDING gather and stitch (the north star's '>= 6x 1 -> 8 GPUs on a 512-slice tomogram'); --steps / --warmup are ignored")
    ap.add_argument("--tomograms", type=int, default=0, metavar="T", help="BASELINE configs[4]: a batch of T tomograms dealt round-robin to the ranks "
                    "(the reference's GPUPool semantics, saber/utils/parallelization.py:137-151: task i -> GPU i %% n_gpus), each segmented and stitched on its own GPU, "
                    "no data-path collective; value = all slices / max-over-ranks seconds")
    ap.add_argument("--tomogram-slices", type=int, default=256, help="slices per tomogram of --tomograms (configs[4]: 256)")
    ap.add_argument("--weak", action="store_true", help="N > 1 only: the weak-scaling form (every rank segments its own --steps slices) as the headline instead of the "
                    "512-slice strong-scaling tomogram")
    ap.add_argument("--no-volume512", action="store_true", help="N = 1: skip the extra keys `volume512` (configs[3] at full size on this GPU) and `alt_dtypes.mxfp8.tomogram256` (configs[4])")
    ap.add_argument("--cu-split", type=int, default=0, metavar="N", help="co-residency experiment (DESIGN.md section 4): every handle's decoder kernels on a stream "
                    "restricted to CUs 0..N-1, its encoder passes on a stream restricted to CUs N..255 (hipExtStreamCreateWithCUMask); 0 = off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-encoder-only", action="store_true", help="skip the extra encoder-only timing (profiling runs: keeps the kernel population of the trace = whole slices)")
    ap.add_argument("--no-video", action="store_true", help="skip the extra measurement of the SAM2 video (memory) path, SURVEY.md 8f-1")
    ap.add_argument("--no-tail", action="store_true", help="skip the extra measurement of the post-filter tail on a slice with a few hundred masks")
    return ap.parse_args()


def cpu_baseline(cfg, weights, image01, crop_n_layers):
    """Oracle ("port") timed on the host cores on a bounded sample: 1 Hiera-L encoder pass + one 64-prompt
    first-pass decoder batch + its 192 m2m refinements + their stability scores (about 10 s of CPU work); extrapolated to one slice
    by the reference's work counts (21 crops / 3072 + 9216 prompts for crop_n_layers=2)."""
    from oracle import sam2_ref
    from oracle.amg_ref import calculate_stability_score
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, 32))  # torch CPU GEMMs stop scaling (and thrash) far below the 256 logical CPUs of the GPU host
    torch.set_num_threads(cores)
    P = sam2_ref.ImagePredictorRef(weights, cfg)
    img3 = np.repeat(image01[..., None], 3, 2)
    t0 = time.perf_counter()
    P.set_image(img3)
    t_enc = time.perf_counter() - t0
    g = (np.arange(8) + 0.5) / 8 * 1024.0
    pts = torch.tensor(np.stack(np.meshgrid(g, g), -1).reshape(-1, 2).astype(np.float32))
    lab = torch.ones(64, 1, dtype=torch.int64)
    t0 = time.perf_counter()
    masks, iou, low = P._predict(pts[:, None], lab, None, True)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    pts3 = pts.repeat_interleave(3, 0)
    lowf = low.flatten(0, 1)
    for b in range(3):      # the three 64-prompt m2m batches of these 64 grid prompts (SABER's points_per_batch = 64)
        m2, i2, _ = P._predict(pts3[64 * b:64 * b + 64, None], torch.ones(64, 1, dtype=torch.int64), lowf[64 * b:64 * b + 64, None], False)
        calculate_stability_score(m2.squeeze(1), 0.0, 0.7)
    t_m2m = time.perf_counter() - t0
    n_side = [2 ** (i + 1) for i in range(crop_n_layers)]
    n_crops = 1 + sum(n * n for n in n_side)
    n_first = sum((32 // (2 ** l)) ** 2 * (1 if l == 0 else (2 ** l) ** 2) for l in range(crop_n_layers + 1))
    per_slice = n_crops * t_enc + n_first / 64.0 * (t_first + t_m2m)
    return {"value": 1.0 / per_slice, "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 torch CPU, {cores} threads: 1 encoder pass ({t_enc:.2f}s) + 64 grid prompts ({t_first:.2f}s) + their 192 "
                      f"m2m refinements incl. stability score ({t_m2m:.2f}s); extrapolated to {n_crops} crops / {n_first} grid prompts per slice",
            "seconds_sampled": t_enc + t_first + t_m2m}


def _is_gemm(name):
    return "gemm_bf16" in name or "gemm_rowln" in name


def pmc_traffic():
    """HBM bytes per launch and MFMA-busy fraction of the dominant kernel class from the latest committed rocprofv3 PMC summary
    (profiles/*_pmc_summary.json: separate FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES passes, FETCH doubled per the gfx950
    correction).  Returns (bytes per launch, mfma busy fraction or None, source file)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")),
                   key=lambda f: [int(t) for t in re.findall(r"\d+", os.path.basename(f))])      # r01_v10 after r01_v9
    if not files:
        return None, None, None
    try:
        k = json.load(open(files[-1]))["kernels"]
        tot = sum(v["hbm_bytes_total"] for n, v in k.items() if _is_gemm(n))
        cnt = sum(v["launches"] for n, v in k.items() if _is_gemm(n))
        busy = sum(v.get("mfma_busy_cycles_total", 0.0) for n, v in k.items() if _is_gemm(n))
        avail = sum(v.get("simd_cycles_total", 0.0) for n, v in k.items() if _is_gemm(n))
        return tot / max(1, cnt), (busy / avail if avail > 0 else None), os.path.basename(files[-1])
    except Exception:
        return None, None, None


def tail_mode(eng, pool, a, segment_slice_to_plane, make_amg_params):
    """The post-filter tail of the metric on a slice that HAS masks.  With the seeded (untrained) weights cfgAMG's default thresholds
    leave ~0.5 masks per slice, so per-crop NMS, the survivors' device copies, cross-crop NMS, pair intersections, dedup and the paint
    kernel run on nothing in the headline number.  Here the score filters are set so that a few hundred masks survive (pred_iou
    threshold = the quantile of this slice's own predicted IoUs that keeps ~250 of the 3 072 full-image m2m masks; stability filter
    and both box NMS disabled because the seeded model's masks are all image-sized blobs that suppress each other), and the same
    slice -> label plane step is timed.  Reported beside the headline, never as it."""
    base = dict(npoints=a.npoints, crop_n_layers=a.crop_n_layers, stability_score_thresh=0.0, box_nms_thresh=1.0, crop_nms_thresh=1.0)
    img = eng.prepare(pool[0])
    _, meta = eng.amg_generate(img, make_amg_params(dict(base, pred_iou_thresh=0.0)), max_masks=16384)
    ious = np.sort(np.array([m.predicted_iou for m in meta], dtype=np.float64))
    if len(ious) < 300:
        return {"skipped": f"only {len(ious)} candidate masks with every filter off"}
    thr = float(ious[-250])
    params = make_amg_params(dict(base, pred_iou_thresh=thr))
    segment_slice_to_plane(eng, pool[0], params, min_mask_area=50, max_masks=4096)
    torch.cuda.synchronize()
    reps, painted, n_amg = 4, 0, 0
    per = []
    for i in range(reps):             # per-slice times, median: a caching-allocator release of the 2-GB mask tensor of the call above lands in one of them
        t0 = time.perf_counter()
        _, n = segment_slice_to_plane(eng, pool[i % len(pool)], params, min_mask_area=50, max_masks=4096)
        torch.cuda.synchronize()
        per.append(time.perf_counter() - t0)
        painted += n
    dt = float(np.median(per))
    _, meta = eng.amg_generate(eng.prepare(pool[0]), params, max_masks=4096)
    n_amg = len(meta)
    syncs = eng.lib.saber_amg_last_syncs(eng.h)
    params0 = make_amg_params(dict(npoints=a.npoints, crop_n_layers=a.crop_n_layers))
    segment_slice_to_plane(eng, pool[0], params0, min_mask_area=50)
    torch.cuda.synchronize()
    per = []
    for i in range(reps):
        t0 = time.perf_counter()
        segment_slice_to_plane(eng, pool[i % len(pool)], params0, min_mask_area=50)
        torch.cuda.synchronize()
        per.append(time.perf_counter() - t0)
    dt0 = float(np.median(per))
    # the same few-hundred-mask slice with the engine's IoU pruning of the m2m pass on (identical masks): what the pruning buys when the IoU head
    # separates candidates the way a trained one does (a few hundred of 9 216 above the threshold)
    eng.set_iou_pruning(True)
    try:
        segment_slice_to_plane(eng, pool[0], params, min_mask_area=50, max_masks=4096)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            segment_slice_to_plane(eng, pool[i % len(pool)], params, min_mask_area=50, max_masks=4096)
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / reps
        _, meta_p = eng.amg_generate(eng.prepare(pool[0]), params, max_masks=4096)
        pruned, seen = eng.last_pruning()
    finally:
        eng.set_iou_pruning(False)
    return {"what": "same step with score filters that leave a few hundred masks (pred_iou_thresh = own quantile, stability / NMS off), one engine handle",
            "pred_iou_thresh": thr, "masks_per_slice": n_amg, "painted_per_slice": painted / reps, "ms_per_slice": dt * 1e3,
            "ms_per_slice_default_thresholds_same_handle": dt0 * 1e3, "tail_ms": (dt - dt0) * 1e3, "host_syncs_per_slice": syncs,
            "with_iou_pruning": {"ms_per_slice": dtp * 1e3, "masks_per_slice": len(meta_p), "m2m_candidates": seen, "pruned": pruned,
                                 "what": "saber_engine_set_iou_pruning(1), the engine's default: candidates that cannot pass pred_iou_thresh skip the mask upscaling; same masks"}}


def precision_check(weights, img, operands="bf16"):
    """One encode + 64 grid prompts (+ their m2m refinement) on a handle that carries both precisions: relative RMS difference of the 16-bit
    production arithmetic (`operands`: bf16 or fp16) from the exact fp32 mode (saber_engine_set_precision; tests/test_gpu_exact.py pins that
    mode to the fp32 CPU oracle at ~1e-6), and the time the exact mode takes."""
    import numpy as np
    import torch
    from saber_amd.engine import Engine

    def rel(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return float(((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-12)).item())
    e = Engine("large", device=0, weights=weights, max_images=1, max_prompts=64, precision="exact", operands=operands)
    try:
        g = np.linspace(1 / 16, 1 - 1 / 16, 8, dtype=np.float32) * 1024
        pts = torch.from_numpy(np.stack(np.meshgrid(g, g), -1).reshape(-1, 2).copy()).cuda()
        res = {}
        for mode in ("exact", operands):
            e.set_precision(mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.encode(img)
            f = {k: v.clone() for k, v in e.get_features(0).items()}
            low, iou, _ = e.decode_points(pts, slot=0, multimask=True)
            mi = torch.clamp(low[:, 0], -32, 32).contiguous()
            low2, iou2, _ = e.decode_points(pts, slot=0, multimask=False, mask_input=mi)
            torch.cuda.synchronize()
            res[mode] = (f, low, iou, low2, iou2, time.perf_counter() - t0)
        x, b = res["exact"], res[operands]
        return {"what": f"{operands} production arithmetic vs the engine's exact (fp32-operand) mode, same handle, same slice: 1 encoder pass + 64 grid prompts (3 masks) + their m2m refinement",
                "rel_rms": {"image_embed": rel(b[0]["image_embed"], x[0]["image_embed"]), "feat_s1": rel(b[0]["feat_s1"], x[0]["feat_s1"]),
                            "feat_s0": rel(b[0]["feat_s0"], x[0]["feat_s0"]), "low_res_logits": rel(b[1], x[1]), "m2m_low_res_logits": rel(b[3], x[3])},
                "pred_iou_max_abs": float((b[2] - x[2]).abs().max().item()), "mask_sign_agreement": float(((b[1] > 0) == (x[1] > 0)).float().mean().item()),
                "seconds": {"exact": x[5], operands: b[5]}}
    finally:
        e.close()


def full_size_keys(a, engines, pool, make_amg_params, Z=512, window=24):
    """BASELINE configs[3] at FULL size on this one GPU, inside the default line: a 1024 x 1024 x Z tomogram resident in HBM
    -> the z-loop on the headline's engine handles -> label planes -> 3-D connected components on the device -> uint32 labels on the host, with the
    size-independent checks of tests/test_gpu_config34_volume.py::test_config3_512_slices_full_size asserted on the result.  The score thresholds
    are that test's (pred_iou 0.5, stability 0.8: the seeded, untrained decoder leaves nothing at cfgAMG's own), so paint / dedup / stitch run on
    real labels; the run at cfgAMG's own thresholds is `bench.py --volume 512`."""
    from saber_amd.segmenters import utils
    from saber_amd.segmenters.slice_driver import segment_slice_to_plane, segment_volume_sharded, shard_bounds
    params = make_amg_params(dict(npoints=a.npoints, crop_n_layers=a.crop_n_layers, pred_iou_thresh=0.5, stability_score_thresh=0.8))
    eng = engines[0]
    vol = torch.stack([pool[z % len(pool)] for z in range(Z)])
    fns = [(lambda z, e_=e_: segment_slice_to_plane(e_, vol[z], params, min_mask_area=50)[0]) for e_ in engines]
    torch.cuda.synchronize()
    tm, keep = {}, []
    t0 = time.perf_counter()
    labels = segment_volume_sharded(vol, fns, stitch=True, min_mask_area=100, engine=eng, timings=tm, keep_on_device=True, planes_out=keep)
    labels_host = labels.cpu()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    planes = keep[0]
    K = int(tm["labels"])
    checks = {}
    # (1) per-slice ids are list positions: contiguous 1..n on every sampled plane (propagation.py:185-186)
    n_fg = int((planes.view(Z, -1).max(1).values > 0).sum().item())
    assert n_fg > Z // 2, f"only {n_fg} of {Z} planes carry masks"
    for z in range(0, Z, 37):       # (contiguous as long as no mask is painted over completely: true for the seeded decoder's masks, not in general)
        ids = torch.unique(planes[z])
        ids = ids[ids > 0]
        assert torch.equal(ids.long(), torch.arange(1, ids.numel() + 1, device=ids.device)), f"plane {z}: ids not contiguous"
    checks["planes_with_masks"] = n_fg
    checks["max_ids_per_plane"] = int(planes.max().item())
    # (2) slices are independent units: first + mid slice of four of the eight 8-rank z-chunks recomputed standalone = the planes of the full run
    with torch.cuda.stream(torch.cuda.Stream()):
        for r in (0, 3, 5, 7):
            z0, z1 = shard_bounds(Z, 8, r)
            assert z1 - z0 == Z // 8
            for z in (z0, z0 + Z // 16 - 1):
                again, _ = segment_slice_to_plane(eng, vol[z], params, min_mask_area=50)
                assert torch.equal(again.view(torch.int16), planes[z]), f"slice {z} recomputed standalone differs"
        torch.cuda.current_stream().synchronize()
    # (3) the stitched volume: K labels, every kept component >= min_mask_area * 10 voxels (utils.py:113-119), labels only on painted voxels
    assert K >= 1 and K == int(labels.max().item())
    counts = torch.bincount(labels.flatten(), minlength=K + 1)
    assert bool((counts[1:] >= 1000).all())
    assert not bool(((labels != 0) & (planes == 0)).any())
    # (4) device stitch = host stitch (scipy) on a z-window, min_mask_area = 0 so that all of the window's components are kept
    w0 = min(200, Z - window)
    win = planes[w0:w0 + window].cpu().numpy().view(np.uint16)
    host = utils.separate_masks(np.ascontiguousarray(win), min_mask_area=0)
    devw, _ = eng.separate_masks(planes[w0:w0 + window].contiguous(), min_mask_area=0)
    assert np.array_equal(devw.cpu().numpy().view(np.uint32), host)
    # (5) the full stitch restricted to the window is a coarsening of the window's own components
    lw = labels_host[w0:w0 + window].numpy().view(np.uint32)
    sel = lw > 0
    pairs = np.unique(np.stack([host[sel].astype(np.int64), lw[sel].astype(np.int64)], 1), axis=0)
    assert len(np.unique(pairs[:, 0])) == len(pairs)
    checks["asserted"] = "ids contiguous per plane; 8 slices at the 8-rank chunk bounds recomputed standalone bit-equal; every label >= 1000 voxels and inside the painted voxels; device stitch == scipy on a 24-plane window; window components map to ONE global label each"
    return {"what": f"BASELINE configs[3] at full size on ONE GPU inside the default line: 1024x1024x{Z} uint16 tomogram (the synthetic slices cycled along z) -> z-loop on the headline's "
                    f"{len(engines)} engine handles -> label planes -> 3-D connected components on the device -> uint32 labels on the host; AMG score thresholds of "
                    f"tests/test_gpu_config34_volume.py (pred_iou 0.5, stability 0.8) so that the planes carry labels",
            "slices": Z, "slices_per_s": Z / dt, "seconds": dt, "labels": K,
            "without_stitch": {"slices_per_s": Z / tm["gathered"], "seconds": tm["gathered"]},
            "phases_s": {"segment": tm["segmented"], "stitch_on_device": tm["stitched"] - tm["gathered"], "labels_to_host": dt - tm["stitched"]},
            "checks": checks}


def tomogram256_key(a, engines, pool, make_amg_params, Z=256):
    """BASELINE configs[4], one tomogram of the batch at full depth on this GPU: Z slices on the MXFP8 handles, hipGraph replay of the per-slice
    encode + decode sequences, stitched on the device (the checks of tests/test_gpu_config34_volume.py::test_config4_256_slices_mxfp8_hipgraph)."""
    from saber_amd.segmenters.slice_driver import segment_slice_to_plane, segment_volume_sharded
    params = make_amg_params(dict(npoints=a.npoints, crop_n_layers=a.crop_n_layers, pred_iou_thresh=0.5, stability_score_thresh=0.8))
    eng = engines[0]
    for e_ in engines:
        e_.set_graphs(True)
    cap0, rep0 = eng.graph_stats()
    vol = torch.stack([pool[z % len(pool)] for z in range(Z)])
    fns = [(lambda z, e_=e_: segment_slice_to_plane(e_, vol[z], params, min_mask_area=50)[0]) for e_ in engines]
    torch.cuda.synchronize()
    tm, keep = {}, []
    t0 = time.perf_counter()
    labels = segment_volume_sharded(vol, fns, stitch=True, min_mask_area=100, engine=eng, timings=tm, keep_on_device=True, planes_out=keep)
    labels_host = labels.cpu()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cap, rep = eng.graph_stats()
    planes, K = keep[0], int(tm["labels"])
    n_fg = int((planes.view(Z, -1).max(1).values > 0).sum().item())
    assert n_fg > Z // 2
    assert rep - rep0 >= 7 * (Z // len(engines) - 6), f"hipGraph replays on handle 0: {rep - rep0}"
    assert K >= 1 and K == int(labels.max().item())
    counts = torch.bincount(labels.flatten(), minlength=K + 1)
    assert bool((counts[1:] >= 1000).all()) and not bool(((labels != 0) & (planes == 0)).any())
    # eager run of a z-subsample on the same handle: planes identical to the replayed run's
    eng.set_graphs(False)
    with torch.cuda.stream(torch.cuda.Stream()):
        for z in (0, Z // 3, Z - 1):
            p, _ = segment_slice_to_plane(eng, vol[z], params, min_mask_area=50)
            assert torch.equal(p.view(torch.int16), planes[z]), z
        torch.cuda.current_stream().synchronize()
    eng.set_graphs(os.environ.get("SABER_AMD_GRAPHS", "1") != "0")
    return {"what": f"BASELINE configs[4], ONE tomogram of the batch at full depth: 1024x1024x{Z}, MXFP8 operands on the fp8 MFMA, hipGraph replay of encode + decode "
                    f"({len(engines)} handles), stitched on the device, labels to the host; thresholds as `volume512`",
            "slices": Z, "slices_per_s": Z / dt, "seconds": dt, "labels": K, "without_stitch": {"slices_per_s": Z / tm["gathered"], "seconds": tm["gathered"]},
            "graph_sequences_captured_handle0": cap, "graph_replays_handle0": rep - rep0,
            "checks": "planes with masks > Z/2; >= 7 replays per slice from the third slice on; labels >= 1000 voxels inside painted voxels; eager planes of 3 slices bit-equal to replayed"}


def volume_modes(a, rank, world, rehearsal, engines, pool, params, dist):
    """--volume Z: ONE tomogram z-sharded over the ranks (strong scaling, gather + stitch inside the timed region);
    --tomograms T: T tomograms dealt round-robin to the ranks, each stitched where it was segmented.  Slices are the synthetic pool slices
    cycled along z (resident in HBM when the timer starts, like the headline); every rank keeps `--workers` engine handles, one thread and
    one HIP stream each, slices dealt round-robin (saber_amd.segmenters.slice_driver.segment_volume_sharded)."""
    from saber_amd.segmenters.slice_driver import segment_slice_to_plane, segment_volume_sharded, shard_bounds
    eng = engines[0]
    dev = pool[0].device

    def fns_for(vol):
        return [(lambda z, e_=e_: segment_slice_to_plane(e_, vol[z], params, min_mask_area=50)[0]) for e_ in engines]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    # warm-up: workspaces, hipGraph capture, communicator
    warm = torch.stack([pool[i % len(pool)] for i in range(2 * len(engines) * max(1, world))])
    segment_volume_sharded(warm, fns_for(warm), stitch=True, min_mask_area=100, engine=eng)
    sync_all()
    base = {"unit": "slices/s", "n_gpus": world, "steps": 1, "warmup": 1, "higher_is_better": True, "vs_baseline": None, "data": "synthetic",
            "dtype": {"bf16": "bf16", "fp16": "fp16", "fp8": "bf16 operands, e4m3 weights", "mxfp8": "mxfp8 operands on the fp8 MFMA (qkv / fc1 / fc2 of stages 2-3), bf16 elsewhere"}[a.dtype]}
    note = " [REHEARSAL on one GPU (gloo): not a multi-GPU measurement]" if rehearsal else ""
    if a.volume > 0:
        Z = a.volume
        vol = torch.stack([pool[z % len(pool)] for z in range(Z)])            # (Z, 1024, 1024) uint16 in HBM
        sync_all()
        # ONE timed pass: this rank's z-chunk -> all-gather of the uint16 planes -> 3-D connected components on the device -> uint32 labels on the
        # host.  `timings` holds the seconds at which the chunk was segmented, the gather had finished and the stitch had finished (each behind a
        # device synchronisation), so the figure without the stitch comes from the same run (rounds 1-4 ran the volume twice for it).
        tm = {}
        t0 = time.perf_counter()
        labels = segment_volume_sharded(vol, fns_for(vol), stitch=True, min_mask_area=100, engine=eng, timings=tm)
        sync_all()
        dt = time.perf_counter() - t0
        t_seg, t_gath, t_st = tm["segmented"], tm["gathered"], tm["stitched"]
        if world > 1:
            t = torch.tensor([dt, t_seg, t_gath, t_st], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, t_seg, t_gath, t_st = (float(x) for x in t.tolist())
        return dict(base, metric="EM slices/sec (1024^2, Hiera-L)" if Z == 512 else "EM slices/sec (1024^2, Hiera-L), one tomogram z-sharded", value=Z / dt, ms_per_step=dt * 1e3, scaling="strong",
                    config={"workload": f"ONE 1024x1024x{Z} uint16 tomogram (BASELINE configs[{2 if Z <= 64 else 3}]): per slice prep.prepare -> SAM2 AMG (Hiera-L, cfgAMG defaults) -> dedup/sort -> label plane; "
                                        f"contiguous z-chunks over {world} rank(s) x {len(engines)} engine handles, ONE all-gather of the uint16 planes, 3-D connected components "
                                        f"(saber_separate_masks) on the device, uint32 labels back on the host: all inside the timed region" + note,
                            "slices": Z, "labels": int(labels.max()) if labels.size else 0, "parallelism": f"STRONG scaling: total work fixed at {Z} slices, {-(-Z // world)} per rank",
                            "engine_handles_per_gpu": len(engines), "weights": "seeded synthetic Hiera-L (no checkpoint offline)"},
                    without_stitch={"slices_per_s": Z / t_gath, "seconds": t_gath, "what": "the same run up to and including the all-gather of the label planes (max over ranks), no 3-D connected components"},
                    phases_s={"segment_own_chunk": t_seg, "all_gather": t_gath - t_seg, "stitch_on_device": t_st - t_gath, "labels_to_host": dt - t_st,
                              "what": "max over ranks of the seconds since the start of the timed region at which each phase ended, differenced"},
                    seconds=dt)
    T, Zt = a.tomograms, a.tomogram_slices
    mine = [t for t in range(T) if t % world == rank]             # GPUPool: task i -> GPU i % n_gpus
    vols = {t: torch.stack([pool[(z + t) % len(pool)] for z in range(Zt)]) for t in mine}
    sync_all()
    t0 = time.perf_counter()
    n_lab = 0
    for t in mine:
        planes_dev = torch.zeros((Zt, 1024, 1024), dtype=torch.int16, device=dev)
        fns = fns_for(vols[t])
        import threading
        errs = []

        def work(w, fns=fns, planes_dev=planes_dev):
            try:
                st = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(st):
                    for z in range(w, Zt, len(fns)):
                        planes_dev[z] = fns[w](z).view(torch.int16)
                    st.synchronize()
            except Exception as ex:
                errs.append(ex)
        th = [threading.Thread(target=work, args=(w,)) for w in range(len(fns))]
        for x in th:
            x.start()
        for x in th:
            x.join()
        if errs:
            raise errs[0]
        labels, n = eng.separate_masks(planes_dev, min_mask_area=100)
        n_lab += n
        labels_host = labels.cpu()                                   # each tomogram's labels leave the GPU, as the reference's writer would take them
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dict(base, metric="EM slices/sec (1024^2, Hiera-L), batch of tomograms, one per GPU at a time", value=T * Zt / dt, ms_per_step=dt * 1e3, scaling="strong",
                config={"workload": f"{T} tomograms of 1024x1024x{Zt} uint16 (BASELINE configs[4]) dealt round-robin to {world} rank(s) (reference GPUPool: task i -> GPU i % n_gpus), each: slice loop on "
                                    f"{len(engines)} engine handles -> 3-D connected components on its GPU -> uint32 labels to the host; no data-path collective" + note,
                        "tomograms": T, "slices_per_tomogram": Zt, "labels_rank0": n_lab, "hipgraph_replay": os.environ.get("SABER_AMD_GRAPHS", "1") != "0",
                        "parallelism": f"total work fixed at {T} tomograms; rank r takes tomograms r, r + {world}, ..."},
                seconds=dt)


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, rendezvous on 127.0.0.1), relay rank 0's JSON line and exit with the worst child status.  The parent
    makes no HIP call and never asks torch for a device (a process that has initialised the GPU must not be the one that forks the ranks),
    which is the reference's multiprocessing mode: one spawned worker per GPU (saber/utils/parallelization.py:339-343)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's peer buffers need it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    # rank 0's stdout is drained by a thread while the parent polls every rank: a rank that dies (no such device, a failed assertion) must not
    # leave the others waiting in a rendezvous or a collective until some outer timeout - they are ended (the exact PIDs started above)
    import threading
    import time
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            rc = failed[0]
            time.sleep(2.0)              # let the others notice by themselves first
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    reader.join(timeout=10)
    out0 = buf[0] if buf else ""
    sys.stdout.write(out0)
    sys.stdout.flush()
    sys.exit(1 if rc else 0)


def spawn_selftest(rank, world):
    """SABER_AMD_BENCH_SPAWN_ONLY=1 (tests/test_bench_launch.py, no GPU): the ranks only rendezvous over gloo and count themselves, so the CPU
    suite covers the launcher of `--gpus N`."""
    import torch.distributed as dist
    if os.environ.get("SABER_AMD_BENCH_SPAWN_FAIL_RANK", "") == str(rank):
        raise SystemExit(3)              # (the launcher must surface a rank's failure: tests/test_bench_launch.py)
    dist.init_process_group("gloo", timeout=__import__("datetime").timedelta(seconds=20))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test", "n_gpus": world, "rank_sum": float(t.item()), "argv": sys.argv[1:]}))
    dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if os.environ.get("SABER_AMD_BENCH_SPAWN_ONLY", "0") == "1" and world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        return spawn_selftest(rank, world)
    import torch.distributed as dist
    # SABER_AMD_BENCH_REHEARSAL=1: rehearse the N > 1 control flow on a ONE-GPU box (every rank on cuda:0, gloo instead of RCCL, the
    # gather staged through host memory).  Never set by the driver; the number it prints is not a multi-GPU measurement.
    rehearsal = os.environ.get("SABER_AMD_BENCH_REHEARSAL", "0") == "1" and world > 1
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    else:
        torch.cuda.set_device(local_rank)

    from saber_amd.engine import Engine, make_amg_params
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    from saber_amd.segmenters.slice_driver import segment_slice_to_plane
    from oracle import saber_ref  # synthetic input recipe only (data generation, not the measured path)

    cfg = get_config("large")
    weights = seeded_weights(cfg, 0)
    def mk_engine(dtype):
        if dtype == "fp16":
            return Engine("large", device=local_rank, weights=weights, max_images=a.max_images, max_prompts=a.max_prompts, precision="fp16")
        return Engine("large", device=local_rank, weights=weights, max_images=a.max_images, max_prompts=a.max_prompts, weight_format=dtype)
    eng = mk_engine(a.dtype)
    amg = dict(npoints=a.npoints, crop_n_layers=a.crop_n_layers)
    params = make_amg_params(amg)

    pool = [torch.from_numpy(saber_ref.synthetic_slice(seed=1000 * rank + i)).cuda() for i in range(2)]
    torch.cuda.synchronize()

    def step(i, planes=None, engine=None):
        plane, n_masks = segment_slice_to_plane(engine or eng, pool[i % len(pool)], params, min_mask_area=50)
        if planes is not None:
            planes[i] = plane
        return n_masks

    engines = [eng] + [mk_engine(a.dtype) for _ in range(a.workers - 1)]
    streams = [torch.cuda.Stream() for _ in engines]
    if a.cu_split > 0:
        import ctypes as C_
        raw = []
        for w, e_ in enumerate(engines):
            hd, he = C_.c_void_p(), C_.c_void_p()
            assert e_.lib.saber_k_stream_create_cu_range(0, a.cu_split, C_.byref(hd)) == 0, e_.lib.saber_k_last_error()
            assert e_.lib.saber_k_stream_create_cu_range(a.cu_split, 256 - a.cu_split, C_.byref(he)) == 0, e_.lib.saber_k_last_error()
            streams[w] = torch.cuda.ExternalStream(hd.value)
            e_.set_encoder_stream(he.value)
            raw += [hd, he]
    # The headline runs with the m2m IoU pruning OFF (every one of the 9 216 refined candidates is upscaled, as in rounds 1-2): how much work the
    # production default (pruning on, identical results) skips depends on the IoU head - a property of the weights, not of the kernels - so the
    # metric does not lean on it.  The time with pruning on and the fraction it pruned are reported beside it (`iou_pruning`).
    for e_ in engines:
        e_.set_iou_pruning(False)

    def run_steps(first, count, planes=None):
        """`count` slices starting at index `first`; with several workers they are dealt round-robin to the engine handles, each driven
        from its own thread on its own stream (the C-ABI calls release the GIL)."""
        if len(engines) == 1:
            with torch.cuda.stream(streams[0]):     # (hipGraph capture needs a non-default stream)
                tot = sum(step(first + i, planes) for i in range(count))
                streams[0].synchronize()
            return tot
        import threading
        totals = [0] * len(engines)

        def work(w):
            with torch.cuda.stream(streams[w]):
                for i in range(w, count, len(engines)):
                    totals[w] += step(first + i, planes, engines[w])
                streams[w].synchronize()
        th = [threading.Thread(target=work, args=(w,)) for w in range(len(engines))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return sum(totals)

    img01_cpu = eng.prepare(pool[0]).cpu().numpy() if (rank == 0 and world == 1 and not a.no_cpu_baseline) else None      # (input of the CPU baseline leg)
    # N > 1 without an explicit workload: the north star's 512-slice strong-scaling tomogram is the line's `value`; the weak-scaling K steps of
    # rounds 1-4 run first (they also warm every handle) and ride along as the extra key `weak_scaling`.
    north_star = world > 1 and not a.weak and a.volume == 0 and a.tomograms == 0
    if (a.volume > 0 or a.tomograms > 0) and not north_star:
        out = volume_modes(a, rank, world, rehearsal, engines, pool, params, dist)
        if rank == 0:
            print(json.dumps(out))
        for e_ in engines:
            e_.close()
        if world > 1:
            dist.destroy_process_group()
        return
    run_steps(0, max(a.warmup, len(engines) if a.warmup else 0))
    planes = torch.zeros((a.steps, 1024, 1024), dtype=torch.uint16, device="cuda")
    gathered = torch.zeros((world * a.steps, 1024, 1024), dtype=torch.uint16, device="cuda") if world > 1 else None
    if world > 1:
        if not rehearsal:       # communicator set-up and the first all-gather's lazy initialisation stay outside the timed region
            dist.all_gather_into_tensor(gathered.view(torch.uint8), planes.view(torch.uint8))
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_masks = run_steps(0, a.steps, planes)
    if world > 1:
        if rehearsal:
            host = torch.empty(gathered.view(torch.uint8).shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, planes.view(torch.uint8).cpu())
            gathered.view(torch.uint8).copy_(host)
        else:
            dist.all_gather_into_tensor(gathered.view(torch.uint8), planes.view(torch.uint8))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        dev_ = "cpu" if rehearsal else "cuda"
        mine = torch.tensor([dt, 1.0], device=dev_, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                       # each rank's own wall time of the timed region (+ a presence flag)
        per_rank = [a.steps / float(t[0].item()) for t in every]
        n_seen = int(sum(float(t[1].item()) for t in every))
        t = torch.tensor([dt], device=dev_, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        n_side = [2 ** (i + 1) for i in range(a.crop_n_layers)]
        n_crops = 1 + sum(n * n for n in n_side)
        n_first = sum((a.npoints // (2 ** l)) ** 2 * (1 if l == 0 else (2 ** l) ** 2) for l in range(a.crop_n_layers + 1))
        alg_flops_slice = n_crops * eng.encoder_flops() + 4 * n_first * 3.639e9
        out = {
            "metric": "EM slices/sec (1024^2, Hiera-L)", "value": world * a.steps / dt, "unit": "slices/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"bf16": "bf16", "fp16": "fp16 (IEEE half MFMA operands, fp32 accumulation: v_mfma_f32_16x16x32_f16)", "fp8": "bf16 operands, e4m3 weights (stage-2/3 block GEMMs)",
                                         "mxfp8": "mxfp8 operands on the fp8 MFMA (qkv / fc1 / fc2 of stages 2-3), bf16 elsewhere"}[a.dtype], "data": "synthetic",
            "config": {"workload": f"1024x1024 uint16 EM slice -> prep.prepare -> SAM2 AMG (Hiera-L, cfgAMG defaults: npoints={a.npoints}, "
                                   f"crop_n_layers={a.crop_n_layers} -> {n_crops} crops, {n_first} grid prompts + {3 * n_first} m2m refinements, multimask) "
                                   f"-> dedup/sort -> uint16 label plane; BASELINE configs[1]",
                       "weights": "seeded synthetic Hiera-L (no checkpoint offline)" + {"bf16": "", "fp16": "; weights, stored activations and every MFMA operand in IEEE half (SABER_PRECISION_FP16): "
                                  "the bf16 kernels compiled for the other 16-bit type, same MFMA rate; NOT the headline precision (BASELINE configs[1] states bf16)", "fp8": "; qkv / proj / fc1 / fc2 of stages 2-3 quantised to OCP e4m3fn with "
                                  "per-row power-of-two scales at load, expanded to bf16 MFMA operands (storage format: the roofline stays the bf16 one)",
                                  "mxfp8": "; qkv (blocks that keep their width) / fc1 / fc2 of stages 2-3 in OCP MXFP8 (e4m3 + e8m0 per 32 K-elements), activations quantised to the same "
                                  "format where produced, products on v_mfma_scale_f32_16x16x128_f8f6f4 (BASELINE configs[4]); NOT the headline precision"}[a.dtype], "slices_per_rank": a.steps, "engine_handles_per_gpu": a.workers,
                       "parallelism": (f"REHEARSAL on one GPU (gloo), not a multi-GPU measurement, x{world}" if rehearsal else
                                       f"WEAK scaling: every one of the {world} ranks (one process per GPU) segments its own {a.steps} slices "
                                       f"(slices are independent units, no data-path collective); one RCCL all_gather of the {world} x {a.steps} uint16 "
                                       f"label planes inside the timed region; value = all ranks' slices / max-over-ranks time") if world > 1 else "single GPU",
                       "masks_per_slice": n_masks / max(1, a.steps), "algorithmic_tflop_per_slice": alg_flops_slice / 1e12},
            "achieved_tflops_algorithmic": alg_flops_slice * world * a.steps / dt / 1e12,
        }
        if world > 1:
            out["n_ranks_seen"] = n_seen
            out["per_rank_slices_per_s"] = [round(v, 4) for v in per_rank]
    if north_star:
        a.volume = int(os.environ.get("SABER_AMD_BENCH_NORTH_STAR_SLICES", "512"))       # (the one-GPU rehearsal may shorten it; the driver never sets this)
        vout = volume_modes(a, rank, world, rehearsal, engines, pool, params, dist)
        if rank == 0:
            vout["weak_scaling"] = {"value": out["value"], "unit": "slices/s", "steps": a.steps, "warmup": a.warmup, "ms_per_step": out["ms_per_step"], "scaling": "weak",
                                    "n_ranks_seen": out["n_ranks_seen"], "per_rank_slices_per_s": out["per_rank_slices_per_s"],
                                    "what": out["config"]["parallelism"]}
            vout["steps"], vout["warmup"] = 1, a.warmup
            vout["steps_note"] = (f"one step = the whole {a.volume}-slice tomogram (total work fixed: strong scaling); --steps {a.steps} / --warmup {a.warmup} "
                                  f"are the per-rank slice counts of `weak_scaling`, which ran first on the same handles")
            print(json.dumps(vout))
        for e_ in engines:
            e_.close()
        dist.destroy_process_group()
        return
    if rank == 0:
        # the stitch that follows the gather (utils.separate_masks, propagation.py:189) on the device; reported beside the metric, not in it
        vol = (gathered if world > 1 else planes).view(torch.int16)
        eng.separate_masks(vol, 100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, n_lab = eng.separate_masks(vol, 100)
        torch.cuda.synchronize()
        out["stitch"] = {"what": "3-D connected components of the gathered label planes on rank 0's GPU (saber_separate_masks)",
                         "slices": int(vol.shape[0]), "ms": (time.perf_counter() - t0) * 1e3, "labels": n_lab}
    if rank == 0 and world == 1 and not a.no_profile and not a.no_tail:
        # hipGraph replay A/B on the same handle (the headline above runs with replay on unless SABER_AMD_GRAPHS=0)
        ab = {}
        for name, on in (("eager", False), ("graphs", True)):
            for e_ in engines:
                e_.set_graphs(on)
            run_steps(0, 2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(0, 6)
            torch.cuda.synchronize()
            ab[f"ms_per_slice_{name}"] = (time.perf_counter() - t0) / 6 * 1e3
        for e_ in engines:             # back to the setting the headline ran with (the A/B loop left replay on)
            e_.set_graphs(os.environ.get("SABER_AMD_GRAPHS", "1") != "0")
        cap, rep = eng.graph_stats()
        out["hipgraph"] = dict(ab, what=f"same step, 6 slices each over the {len(engines)} engine handle(s) of the headline, launch sequences issued eagerly vs replayed from hipGraphs (encoder pass + each decoder batch)",
                               sequences_captured=cap, replays_so_far=rep, headline_uses_graphs=os.environ.get("SABER_AMD_GRAPHS", "1") != "0")
    if rank == 0 and world == 1 and not a.no_profile and not a.no_tail:
        # the production default: IoU pruning of the m2m pass on (results identical, tests/test_gpu_graphs.py); NOT the headline, see above
        for e_ in engines:
            e_.set_iou_pruning(True)
        run_steps(0, 2 * len(engines))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(0, 6)
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / 6
        pruned, seen = eng.last_pruning()
        out["iou_pruning"] = {"what": "same step with saber_engine_set_iou_pruning on (the engine's default): m2m candidates whose four IoU predictions are all <= pred_iou_thresh skip "
                                      "the mask upscaling; identical masks.  The fraction pruned (`pruned` of `m2m_candidates`, measured here) is a property of the weights' IoU head, "
                                      "which is why the headline runs with pruning off",
                              "ms_per_slice": dtp * 1e3, "slices_per_s": 1.0 / dtp, "m2m_candidates": seen, "pruned": pruned, "pruned_fraction": (pruned / seen) if seen else None}
        for e_ in engines:
            e_.set_iou_pruning(False)
    if rank == 0 and world == 1 and not a.no_profile:
        eng.profile_begin()
        step(0)
        prof = eng.profile_end()
        g = prof["gemm_bf16"]
        total_ms = sum(v["ms"] for v in prof.values())
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        traffic, mfma_busy, traffic_src = pmc_traffic()
        alg_bytes = g["bytes"] / max(1, g["launches"])
        out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                           "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": alg_bytes, "traffic_over_algorithmic": (traffic / alg_bytes) if traffic and alg_bytes else None,
                           "mfma_busy_frac": mfma_busy, "mfma_busy_unit": "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), rocprofv3 PMC, same summary file",
                           "kernel": "gemm_bf16 (gemm_bf16_glds2_kernel + gemm_bf16_p256s_kernel + gemm_rowln_kernel + gemm_bf16_glds_kernel<4> + gemm_bf16_kernel<T>; since round 3 the ~300 token-side GEMM launches of the decoder are part of dec_tokens_kernel, class decoder_attention)", "launches_per_slice": g["launches"],
                           "avg_launch_us": g["ms"] * 1e3 / max(1, g["launches"]),
                           "algorithmic_gflop_per_launch": g["flops"] / max(1, g["launches"]) / 1e9,
                           "kernel_ms_per_slice": g["ms"], "share_of_kernel_time": g["ms"] / total_ms if total_ms else None,
                           "measured_on": "one single-handle step after the timed region (with two slices in flight the kernels of the two streams overlap and stretch each other)",
                           # round-to-round comparable form: since round 2 the class's time contains the LayerNorm work that round 1 ran as
                           # separate layernorm-class launches (same FLOPs); FLOPs over (gemm + layernorm) class time: r01 0.204, see DESIGN.md 4
                           "frac_with_layernorm_class": (g["flops"] / ((g["ms"] + prof["layernorm"]["ms"]) * 1e-3) / 1e12 / 2500.0) if g["ms"] > 0 else None}
        if a.dtype == "mxfp8":
            # the fp8-MFMA line's own roofline: the MXFP8 GEMM launches against the dense fp8 peak; the bf16 class keeps its key
            g8 = prof["gemm_mxfp8"]
            ach8 = g8["flops"] / (g8["ms"] * 1e-3) / 1e12 if g8["ms"] > 0 else 0.0
            out["roofline_bf16_gemms"] = out["roofline"]
            tr8, busy8, src8 = None, None, None
            try:       # HBM counters of the MXFP8 GEMM launches: separate FETCH_SIZE / WRITE_SIZE / MFMA-busy passes of `bench.py --dtype mxfp8` (tools/summarize_pmc.py)
                import glob
                f8 = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_mxfp8_pmc.json")))[-1]
                k8 = {n: v for n, v in json.load(open(f8))["kernels"].items() if "gemm_mx" in n}
                tr8 = sum(v["hbm_bytes_total"] for v in k8.values()) / max(1, sum(v["launches"] for v in k8.values()))
                bz, av = sum(v.get("mfma_busy_cycles_total", 0.0) for v in k8.values()), sum(v.get("simd_cycles_total", 0.0) for v in k8.values())
                busy8, src8 = (bz / av if av > 0 else None), os.path.basename(f8)
            except Exception:
                pass
            out["roofline"] = {"bound": "mfma", "achieved": ach8, "peak": 5000.0, "unit": "TFLOP/s", "frac": ach8 / 5000.0, "traffic": tr8, "traffic_source": src8,
                               "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, FETCH_SIZE x2 + WRITE_SIZE)", "algorithmic_bytes_per_launch": g8["bytes"] / max(1, g8["launches"]),
                               "traffic_over_algorithmic": (tr8 / (g8["bytes"] / max(1, g8["launches"]))) if tr8 and g8["bytes"] else None, "mfma_busy_frac": busy8,
                               "kernel": "gemm_mx_kernel (csrc/gemm_fp8.hip: v_mfma_scale_f32_16x16x128_f8f6f4, persistent 256x192 tiles)", "launches_per_slice": g8["launches"],
                               "avg_launch_us": g8["ms"] * 1e3 / max(1, g8["launches"]), "algorithmic_gflop_per_launch": g8["flops"] / max(1, g8["launches"]) / 1e9,
                               "kernel_ms_per_slice": g8["ms"], "share_of_kernel_time": g8["ms"] / total_ms if total_ms else None,
                               "note": "dense fp8 peak 5 PFLOP/s"}
        out["kernel_classes_ms_per_slice"] = {k: round(v["ms"], 3) for k, v in prof.items()}
        out["kernel_classes_launches"] = {k: v["launches"] for k, v in prof.items()}
        mp = prof["mask_post"]
        if mp["ms"] > 0:
            out["mask_post_hbm"] = {"bound": "hbm", "achieved": mp["bytes"] / (mp["ms"] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s"}
    if rank == 0 and world == 1 and not a.no_profile and not a.no_encoder_only:
        # encoder-only figure (the north star's roofline target is defined on the Hiera-L encoder)
        img = eng.prepare(pool[0])
        reps, nb = 3, a.max_images
        crops = [[0, 0, 1024, 1024]] * nb
        eng.encode(img, crops)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.encode(img, crops)
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / (reps * nb)
        out["encoder_only"] = {"passes_per_s": 1.0 / te, "ms_per_pass": te * 1e3, "batch": nb, "algorithmic_tflops": eng.encoder_flops() / te / 1e12,
                               "frac_of_bf16_peak": eng.encoder_flops() / te / 2.5e15}
    if rank == 0 and world == 1 and not a.no_profile and not a.no_tail:
        try:
            out["tail"] = tail_mode(eng, pool, a, segment_slice_to_plane, make_amg_params)
        except Exception as ex:     # the extra key must never take the headline down with it
            out["tail"] = {"error": str(ex)[:300]}
    if rank == 0 and world == 1 and not a.no_profile and not a.no_video and a.dtype == "bf16":
        try:        # the next row of the scope table: frames per second of the memory path, beside the metric, never in it
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
            import video_bench
            out["propagation"] = video_bench.run("large", 32)
        except Exception as ex:
            out["propagation"] = {"error": str(ex)[:300]}
    if rank == 0 and world == 1 and not a.no_profile and not a.no_tail and a.dtype in ("bf16", "fp16"):
        try:        # what the 16-bit operands of the headline cost against the engine's exact (fp32) mode on this slice; never part of `value`
            out["precision"] = precision_check(weights, eng.prepare(pool[0]), a.dtype)
        except Exception as ex:
            out["precision"] = {"error": str(ex)[:300]}
    if rank == 0 and world == 1 and not a.no_profile and not a.no_tail and not a.no_volume512:
        try:        # configs[3] at full size on the headline's handles (about 75 s); never part of `value`
            out["volume512"] = full_size_keys(a, engines, pool, make_amg_params)
        except Exception as ex:
            out["volume512"] = {"error": f"{type(ex).__name__}: {str(ex)[:300]}"}
    if rank == 0 and world == 1 and not a.no_profile and not a.no_tail and not a.no_alt_dtypes and a.dtype == "bf16":
        # The other arithmetic modes as extra keys of the default line, each on fresh handles after the headline's were closed
        # (a Hiera-L handle with its AMG scratch is ~58 GB): the same step, the same slices, `--workers` handles; never part of `value`.
        img_prec = eng.prepare(pool[0]).clone()
        for e_ in engines:
            e_.close()
        engines.clear()
        out["alt_dtypes"] = {}
        for dt_name in ("fp16", "mxfp8"):
            try:
                alt = [mk_engine(dt_name) for _ in range(a.workers)]
                for e_ in alt:
                    e_.set_iou_pruning(False)
                engines.extend(alt)
                eng = engines[0]
                run_steps(0, 2 * len(engines))
                torch.cuda.synchronize()
                n_alt = max(8, a.steps // 2)
                t0 = time.perf_counter()
                run_steps(0, n_alt)
                torch.cuda.synchronize()
                dta = (time.perf_counter() - t0) / n_alt
                rec = {"slices_per_s": 1.0 / dta, "ms_per_slice": dta * 1e3, "slices": n_alt, "engine_handles": len(engines),
                       "vs_headline": (dt / a.steps) / dta}
                eng.profile_begin()
                step(0)
                pr = eng.profile_end()
                rec["kernel_classes_ms_per_slice"] = {k: round(v["ms"], 3) for k, v in pr.items()}
                g = pr["gemm_bf16"]
                if g["ms"] > 0 and dt_name == "fp16":
                    ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
                    rec["roofline"] = {"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0, "kernel": "the gemm class of the headline, compiled for fp16 operands (namespace op_f16)",
                                       "avg_launch_us": g["ms"] * 1e3 / max(1, g["launches"]), "launches_per_slice": g["launches"]}
                if dt_name == "fp16":
                    rec["precision"] = precision_check(weights, img_prec, "fp16")
                if dt_name == "mxfp8" and not a.no_volume512:
                    try:        # configs[4] at full depth (about 35 s)
                        rec["tomogram256"] = tomogram256_key(a, engines, pool, make_amg_params)
                    except Exception as ex:
                        rec["tomogram256"] = {"error": f"{type(ex).__name__}: {str(ex)[:300]}"}
                out["alt_dtypes"][dt_name] = rec
            except Exception as ex:
                out["alt_dtypes"][dt_name] = {"error": str(ex)[:300]}
            for e_ in engines:
                e_.close()
            engines.clear()
        try:        # the same step on the seeded encoder + the FITTED mask decoder (tests/golden/decoder_fit_large_seed0.npz): compact masks
            #           with a spread of predicted IoU / stability, so cfgAMG's own thresholds and both NMS stages leave real masks and the post-filter
            #           tail (per-crop / cross-crop NMS, survivors' copies, pair intersections, dedup, sort, paint) runs on them; never part of `value`
            from saber_amd.weights import fitted_decoder_weights
            Wf = fitted_decoder_weights(cfg, 0)
            alt = [Engine("large", device=local_rank, weights=Wf, max_images=a.max_images, max_prompts=a.max_prompts) for _ in range(a.workers)]
            engines.extend(alt)
            eng = engines[0]
            rec = {}
            for prune in (False, True):
                for e_ in engines:
                    e_.set_iou_pruning(prune)
                run_steps(0, 2 * len(engines))
                torch.cuda.synchronize()
                n_alt = max(8, a.steps // 2)
                t0 = time.perf_counter()
                nm = run_steps(0, n_alt)
                torch.cuda.synchronize()
                dta = (time.perf_counter() - t0) / n_alt
                rec["iou_pruning_on" if prune else "iou_pruning_off"] = {"slices_per_s": 1.0 / dta, "ms_per_slice": dta * 1e3, "slices": n_alt, "masks_per_slice": nm / n_alt, "vs_headline": (dt / a.steps) / dta}
            pruned, seen = eng.last_pruning()
            rec["iou_pruning_on"].update(m2m_candidates=seen, pruned=pruned)
            rec["what"] = ("the headline's step (bf16, cfgAMG defaults, two handles) on the seeded Hiera-L encoder with the fitted mask decoder (oracle/fit_decoder_heads.py): the filters, "
                           "both NMS stages and the paint tail work on real masks; `iou_pruning_on` is the engine's default (m2m candidates that cannot pass pred_iou_thresh skip the upscaling), "
                           "whose saving depends on the IoU head - here a fitted one")
            out["fitted_decoder"] = rec
        except Exception as ex:
            out["fitted_decoder"] = {"error": f"{type(ex).__name__}: {str(ex)[:300]}"}
        for e_ in engines:
            e_.close()
        engines.clear()
        try:        # the exact (fp32-operand) mode on the same step: ONE handle, the whole slice (21 crops, 3 072 + 9 216 prompts); a verification mode, never part of `value`
            ex_eng = Engine("large", device=local_rank, weights=weights, max_images=a.max_images, max_prompts=a.max_prompts, precision="exact")
            ex_eng.set_iou_pruning(False)
            step(0, engine=ex_eng)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(2):
                step(i, engine=ex_eng)
            torch.cuda.synchronize()
            dte = (time.perf_counter() - t0) / 2
            out["alt_dtypes"]["exact"] = {"slices_per_s": 1.0 / dte, "ms_per_slice": dte * 1e3, "slices": 2, "engine_handles": 1, "vs_headline": (dt / a.steps) / dte,
                                          "what": "csrc/exact.hip: fp32 operands end to end on v_mfma_f32_16x16x4_f32, ~1e-6 from the fp32 oracle (tests/test_gpu_exact.py)"}
            ex_eng.close()
        except Exception as ex:
            out["alt_dtypes"]["exact"] = {"error": str(ex)[:300]}
        eng = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, weights, img01_cpu, a.crop_n_layers)
    if rank == 0:
        print(json.dumps(out))
    for e_ in engines:
        e_.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
