"""segment_tomogram_core (reference: saber/entry_points/inference_core.py:9-98) - the function both `saber segment tomograms` front ends call
per copick run: read the tomogram, run the segmenter, smooth the label volume (per-label adaptive 3-D Gaussian, on the device here), cast
to uint8, write the segmentation, reset the segmenter's inference state.

On-disk formats are NOT re-implemented: the reference reads and writes through third-party `copick_utils`
(`readers.tomogram`, `writers.segmentation`), which is absent from this image.  The two callables are therefore parameters (defaulting to
copick_utils' own when it is installed), so a maintainer keeps the reference's I/O and swaps only the compute."""
import logging

import numpy as np
import torch

from saber_amd.filters import masks as mask_filters


def _copick_io():
    try:
        from copick_utils.io import readers, writers          # the reference's own I/O layer (inference_core.py:5)
    except ImportError as ex:
        raise ImportError("copick_utils is not installed: pass read_tomogram= / write_segmentation= callables "
                          "(signatures of copick_utils.io.readers.tomogram / writers.segmentation)") from ex
    return readers.tomogram, writers.segmentation


def segment_tomogram_core(run, voxel_size: float, tomogram_algorithm: str, segmentation_name: str, segmentation_session_id: str,
                          slab_thickness: int, num_slabs: int, delta_z: int, display_segmentation: bool, segmenter, gpu_id: int = 0,
                          target_class: int = 1, *, read_tomogram=None, write_segmentation=None):
    logger = logging.getLogger(__name__)
    if read_tomogram is None or write_segmentation is None:
        rd, wr = _copick_io()
        read_tomogram, write_segmentation = read_tomogram or rd, write_segmentation or wr
    vol = read_tomogram(run, voxel_size, algorithm=tomogram_algorithm)
    if vol is None:
        logger.info(f"No Tomogram Found for {run.name}")
        return None
    torch.cuda.set_device(gpu_id)
    img_name = run.name + "-" + segmentation_session_id
    if num_slabs > 1:
        segment_mask = segmenter.segment(vol, slab_thickness, num_slabs, delta_z, img_name, display_segmentation)
    else:
        segment_mask = segmenter.segment(vol, slab_thickness, target_class=target_class, save_run=img_name, display=display_segmentation)
    if segment_mask is None:
        logger.info(f"No Segmentation Found for {run.name}")
        return None
    if not display_segmentation:
        segment_mask = mask_filters.fast_3d_gaussian_smoothing(segment_mask, scale=0.05, deviceID=gpu_id)
        segment_mask = segment_mask.astype(np.uint8)
        write_segmentation(run, segment_mask, "saber", name=segmentation_name, session_id=segmentation_session_id, voxel_size=float(voxel_size))
        logger.info(f"Saved Segmentation for {run.name} as {segmentation_name}")
    del vol, segment_mask
    torch.cuda.empty_cache()
    segmenter.inference_state = None
    return
