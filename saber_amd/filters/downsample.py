"""Fourier-crop down-sampling (reference: saber/filters/downsample.py): the resampling `saber segment micrographs` applies before the 2-D
path (entry_points/inference_core.py:123-128) and read_movie applies per frame.  Caller-side plumbing of the hot path: torch.fft on the
engine's device (rocFFT), centred crop of the shifted spectrum, inverse transform.

FourierRescale2D._rescale (:151-204): new size = int(size / scale) made even, crop start (size - new) // 2 + (size odd), un-normalised
forward and inverse transforms (so intensities scale by new_area / old_area), magnitude of the inverse.  FourierRescale3D (:4-130): sizes
from round(extent / output voxel) made even, 'ortho' transforms, real part."""
import numpy as np
import torch


def _device(device=None):
    if device is not None:
        return device
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


class FourierRescale2D:
    @staticmethod
    def run_resolution(image, input_pixsize: float, target_pixsize: float, device: torch.device = None):
        if target_pixsize <= input_pixsize:
            raise ValueError(f"Target pixel size ({target_pixsize}Å) must be larger than current pixel size ({input_pixsize}Å)")
        return FourierRescale2D._rescale(image, target_pixsize / input_pixsize, device)

    @staticmethod
    def run(image, scale_factor: float, device: torch.device = None):
        if scale_factor < 1:
            raise ValueError("Scale factor must be greater than 1")
        return FourierRescale2D._rescale(image, scale_factor, device)

    @staticmethod
    def _rescale(image, scale_factor: float, device: torch.device = None):
        device = _device(device)
        is_numpy = isinstance(image, np.ndarray)
        x = (torch.from_numpy(image) if is_numpy else image).to(device)
        h, w = x.shape
        hn, wn = int(h / scale_factor), int(w / scale_factor)
        hn, wn = hn - hn % 2, wn - wn % 2
        h0, w0 = (h - hn) // 2 + h % 2, (w - wn) // 2 + w % 2
        spec = torch.fft.fftshift(torch.fft.fft2(x))[h0:h0 + hn, w0:w0 + wn]
        out = torch.abs(torch.fft.ifft2(torch.fft.ifftshift(spec))).cpu()
        return out.numpy() if is_numpy else out


class FourierRescale3D:
    def __init__(self, input_voxel_size, output_voxel_size):
        if isinstance(input_voxel_size, (int, float)):
            input_voxel_size = (input_voxel_size,) * 3
        if isinstance(output_voxel_size, (int, float)):
            output_voxel_size = (output_voxel_size,) * 3
        self.input_voxel_size, self.output_voxel_size = input_voxel_size, output_voxel_size
        if any(o < i for i, o in zip(input_voxel_size, output_voxel_size)):
            raise ValueError("Output voxel size must be greater than or equal to the input voxel size.")
        self.device = _device()

    def calculate_cropping(self, volume):
        dims = volume.shape[-3:]
        new = [int(round(n * i / o)) for n, i, o in zip(dims, self.input_voxel_size, self.output_voxel_size)]
        new = [n - n % 2 for n in new]
        start = [(n - m) // 2 + n % 2 for n, m in zip(dims, new)]
        return (*start, *new)

    def run(self, volume):
        is_numpy = isinstance(volume, np.ndarray)
        v = torch.from_numpy(volume) if is_numpy else volume
        if self.device.type == "cpu" and v.dim() == 4:
            raise AssertionError("Batched volumes are not allowed on CPU. Please provide a single volume.")
        out = self.batched_rescale(v).cpu()
        return out.numpy() if is_numpy else out

    def batched_rescale(self, volume: torch.Tensor):
        v = volume.to(self.device)
        dims = (-3, -2, -1)
        spec = torch.fft.fftshift(torch.fft.fftn(v, dim=dims, norm="ortho"), dim=dims)
        d0, h0, w0, dn, hn, wn = self.calculate_cropping(v)
        spec = spec[..., d0:d0 + dn, h0:h0 + hn, w0:w0 + wn]
        return torch.fft.ifftn(torch.fft.ifftshift(spec, dim=dims), dim=dims, norm="ortho").real

    single_rescale = batched_rescale
