"""Device-resident slice pipeline and the z-sharded volume driver.

`segment_slice_to_plane` is one iteration of the reference's volumetric hot loop
(propagationSegmenter.slice_by_slice, saber/segmenters/propagation.py:180-188) with every array kept in HBM:

    segment_image (base.py:143-149) -> adapter.segment_image_2d (adapters/sam2/predictor.py:48-70)
        prep.prepare            -> K0 on device
        mask_generator.generate -> saber_amg_generate (bit-packed masks + per-mask scalars)
        FilteredSAM2MaskGenerator area filter (amg.py:87-110, min_area_filter = min_mask_area)
    _apply_classifier (base.py:159-176): area >= min_mask_area, remove_duplicate_masks, ascending-area sort
    paint idx+1 in list order (later masks overwrite), np.maximum with the zero plane (:185-188)

`segment_volume_sharded` is the multi-GPU form: rank r owns a contiguous z-chunk, planes are all-gathered
(RCCL when the process group is NCCL, gloo in the CPU tests) and rank 0 (or every rank) stitches with separate_masks.
"""
from typing import Callable, Optional, Tuple

import numpy as np
import torch

from . import utils


def select_and_order(meta, inter: np.ndarray, min_mask_area: int, remove_repeating_masks: bool = True):
    """Host decisions of saber2D._apply_classifier (classifier=None) on per-mask scalars.
    Returns indices into `meta` in final (paint) order."""
    idx = [i for i, m in enumerate(meta) if m.area >= min_mask_area]
    if remove_repeating_masks and idx:
        sub = inter[np.ix_(idx, idx)]
        keep = utils.duplicate_groups_from_counts([meta[i].area for i in idx], sub, [meta[i].stability_score for i in idx])
        idx = [idx[k] for k in keep]
    idx.sort(key=lambda i: meta[i].area)  # sorted() is stable: ties keep AMG order
    return idx


def segment_slice_to_plane(engine, raw_slice: torch.Tensor, params, min_mask_area: int = 50,
                           remove_repeating_masks: bool = True, max_masks: int = 2048) -> Tuple[torch.Tensor, int]:
    """raw_slice: (H,W) uint16/float32 device tensor.  Returns ((H,W) uint16 label plane on device, n masks).
    max_masks is the same capacity the adapter path uses (EngineMaskGenerator.max_masks); a denser slice is retried once with the
    count the engine reports instead of failing the volume."""
    H, W = raw_slice.shape
    img = engine.prepare(raw_slice)
    bits, meta = engine.amg_generate(img, params, max_masks=max_masks)
    if len(meta) == 0:
        return torch.zeros((H, W), dtype=torch.uint16, device=raw_slice.device), 0
    inter = engine.pair_intersections(bits, H, W).cpu().numpy() if remove_repeating_masks else None
    order = select_and_order(meta, inter, min_mask_area, remove_repeating_masks)
    plane = engine.label_plane(bits, order, H, W)
    return plane, len(order)


def shard_bounds(Z: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous z-chunks, remainder spread over the first ranks."""
    base, rem = divmod(Z, world)
    z0 = rank * base + min(rank, rem)
    return z0, z0 + base + (1 if rank < rem else 0)


def segment_volume_sharded(volume, slice_fn, stitch: bool = True, min_mask_area: int = 100,
                           group=None, engine=None, smooth_scale: Optional[float] = None, timings: Optional[dict] = None,
                           keep_on_device: bool = False, planes_out: Optional[list] = None) -> Optional[np.ndarray]:
    """Slice-parallel slice_by_slice.  `volume` only supplies the shape (Z,H,W); `slice_fn(z)` returns the uint16
    label plane of slice z as a tensor on this rank's device.  All ranks receive every plane (all_gather of equal,
    zero-padded chunks); the (Z,H,W) uint32 stitched labels are returned (reference: utils.separate_masks).  With `engine`
    given and the planes on its device the stitch runs there (saber_separate_masks, bit-identical to the host path).
    `slice_fn` may be a LIST of callables, one per engine handle of this rank: its slices are then dealt round-robin to one thread
    per handle, each on its own HIP stream (two slices in flight per GPU fill each other's idle issue slots: +6 % measured).
    `smooth_scale` (device stitch only) also applies the post step of segment_tomogram_core (inference_core.py:68-74, scale 0.05):
    fast_3d_gaussian_smoothing on the stitched labels without leaving the device; the result is then the uint8 volume it returns.
    `timings` (a dict, bench.py): filled with the seconds since entry at which this rank's chunk was segmented (`segmented`), every rank's
    planes had arrived (`gathered`) and the stitch had finished on the device (`stitched`), each behind a device synchronisation.
    `keep_on_device` (device stitch only): return the (Z,H,W) int32 device tensor that holds the uint32 labels instead of a host array;
    `planes_out` (device stitch only): a list that receives the gathered (Z,H,W) int16 device tensor of label planes the stitch ran on."""
    import time
    import torch.distributed as dist
    t_entry = time.perf_counter()

    def stamp(key, dev_):
        if timings is not None:
            if dev_ is not None and dev_.type == "cuda":
                torch.cuda.synchronize(dev_)
            timings[key] = time.perf_counter() - t_entry
    Z, H, W = volume.shape
    dist_on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if dist_on else 1
    rank = dist.get_rank(group) if dist_on else 0
    z0, z1 = shard_bounds(Z, world, rank)
    chunk = (Z + world - 1) // world
    dev = None
    local = None
    fns = list(slice_fn) if isinstance(slice_fn, (list, tuple)) else [slice_fn]
    if len(fns) > 1 and z1 - z0 > 1 and torch.cuda.is_available():
        import threading
        first = fns[0](z0)
        dev = first.device
        local = torch.zeros((chunk, H, W), dtype=torch.int16, device=dev)
        local[0] = first.view(torch.int16) if first.dtype == torch.uint16 else first.to(torch.int16)
        torch.cuda.synchronize(dev)
        errors = []
        inf_mode = torch.is_inference_mode_enabled()

        def work(w):
            try:
                st = torch.cuda.Stream(device=dev)
                with torch.inference_mode(inf_mode), torch.cuda.stream(st):     # inference mode is thread-local: match the caller's
                    for i in range(1 + w, z1 - z0, len(fns)):
                        p = fns[w](z0 + i)
                        local[i] = p.view(torch.int16) if p.dtype == torch.uint16 else p.to(torch.int16)
                    st.synchronize()
            except Exception as e:  # surfaced on the calling thread
                errors.append(e)
        th = [threading.Thread(target=work, args=(w,)) for w in range(len(fns))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            raise errors[0]
    else:
        for i, z in enumerate(range(z0, z1)):
            p = fns[0](z)
            if local is None:
                dev = p.device
                local = torch.zeros((chunk, H, W), dtype=torch.int16, device=dev)
            local[i] = p.view(torch.int16) if p.dtype == torch.uint16 else p.to(torch.int16)
    if local is None:  # rank without slices
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() and dist_on and dist.get_backend(group) == "nccl" else torch.device("cpu")
        local = torch.zeros((chunk, H, W), dtype=torch.int16, device=dev)
    stamp("segmented", dev)
    if world > 1:
        full = torch.empty((world * chunk, H, W), dtype=torch.int16, device=dev)
        # byte view: gloo (CPU tests) has no int16 collectives; RCCL moves the same bytes
        if local.is_cuda and dist.get_backend(group) != "nccl":
            # device planes under a host backend (the one-GPU rehearsal of the multi-rank control flow, bench.py SABER_AMD_BENCH_REHEARSAL=1):
            # staged through host memory
            host = torch.empty(full.view(torch.uint8).shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, local.view(torch.uint8).cpu(), group=group)
            full.view(torch.uint8).copy_(host)
        else:
            dist.all_gather_into_tensor(full.view(torch.uint8), local.view(torch.uint8), group=group)
    else:
        full = local
    stamp("gathered", dev)
    if stitch and engine is not None and full.is_cuda:
        parts = []
        for r in range(world):
            a, b = shard_bounds(Z, world, r)
            parts.append(full[r * chunk: r * chunk + (b - a)])
        planes_dev = parts[0] if world == 1 and parts[0].shape[0] == Z else torch.cat(parts, 0)
        planes_dev = planes_dev.contiguous()
        if planes_out is not None:
            planes_out.append(planes_dev)
        labels, n_labels = engine.separate_masks(planes_dev, min_mask_area=min_mask_area)
        stamp("stitched", dev)
        if timings is not None:
            timings["labels"] = n_labels
        if keep_on_device and smooth_scale is None:
            return labels
        if smooth_scale is not None:
            smoothed, _ = engine.smooth_labels(labels, smooth_scale)
            return smoothed.cpu().numpy()
        return labels.cpu().numpy().view(np.uint32)
    planes = np.empty((Z, H, W), dtype=np.uint16)
    full_np = full.cpu().numpy().view(np.uint16)
    for r in range(world):
        a, b = shard_bounds(Z, world, r)
        planes[a:b] = full_np[r * chunk: r * chunk + (b - a)]
    if not stitch:
        return planes
    if smooth_scale is not None:
        raise RuntimeError("smooth_scale needs the device stitch (pass engine=... and device planes): there is no CPU smoothing path")
    return utils.separate_masks(planes, min_mask_area=min_mask_area)
