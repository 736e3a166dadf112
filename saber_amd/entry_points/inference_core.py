"""segment_tomogram_core (reference: saber/entry_points/inference_core.py:9-98) - the function both `saber segment tomograms` front ends call
per copick run: read the tomogram, run the segmenter, smooth the label volume (per-label adaptive 3-D Gaussian, on the device here), cast
to uint8, write the segmentation, reset the segmenter's inference state.

The copick project layer is NOT re-implemented: the reference reads and writes tomograms through third-party `copick_utils`
(`readers.tomogram`, `writers.segmentation`), which is absent from this image.  The two callables are therefore parameters (defaulting to
copick_utils' own when it is installed), so a maintainer keeps the reference's I/O and swaps only the compute.

segment_micrograph_core (inference_core.py:100-164) - the per-file body of `saber segment micrographs`: read the micrograph (MRC / TIFF),
Fourier-crop to the target resolution, run the 2-D segmenter, write image + label stack as one run of the OME-Zarr store (SURVEY.md 8
row f-4: saber_amd.utils.{mrc,tiff,zarr_v2,zarr_writer})."""
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


def segment_micrograph_core(input: str, output: str, scale_factor: float, target_resolution: float, display_image: bool,
                            use_sliding_window: bool, gpu_id, models):
    import os

    from saber_amd.filters.downsample import FourierRescale2D
    from saber_amd.utils import io, zarr_writer
    segmenter = models["segmenter"]
    zwriter = zarr_writer.get_zarr_writer(output)
    zwriter.set_dict_attr("amg", segmenter.adapter_cfg.amg_cfg.to_dict())
    torch.cuda.set_device(gpu_id)
    image, pixel_size = io.read_micrograph(input)
    image = image.astype(np.float32)
    # (the reference compares target_resolution with a pixel size that may be None for TIFF input and fails there; same here)
    if target_resolution is not None and target_resolution > pixel_size:
        image = FourierRescale2D.run(image, target_resolution / pixel_size)
    elif scale_factor is not None:
        image = FourierRescale2D.run(image, scale_factor)
    segmenter.segment(image, target_class=models.get("target_class", -1), display=False, use_sliding_window=use_sliding_window)
    if isinstance(pixel_size, np.ndarray):
        pixel_size = pixel_size.item()
    masks = mask_filters.masks_to_array(segmenter.masks)
    pixel_size = pixel_size / 10 if pixel_size is not None else 1        # Angstrom -> nanometer (inference_core.py:144-148)
    out_image = segmenter.image
    if out_image.ndim == 3:
        out_image = out_image[:, :, 0]
    zwriter.write(run_name=os.path.splitext(os.path.basename(input))[0], image=out_image, masks=masks, pixel_size=pixel_size)
