"""Device versions of the volume filters that sit either side of the slice-wise path."""
from .gaussian import gaussian_smoothing_3d
from .masks import fast_3d_gaussian_smoothing

__all__ = ["fast_3d_gaussian_smoothing", "gaussian_smoothing_3d"]
