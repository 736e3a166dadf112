"""One weight-less engine handle per device for the filters (the C-ABI entry points take a handle for the device binding and the
error string; creating one costs no device memory)."""
import threading

import torch

from ..engine import Engine

_lock = threading.Lock()
_handles = {}


def device_index(device=None) -> int:
    """Accepts what the reference passes around: None, an int device id, a torch.device or a 'cuda:N' string."""
    if device is None:
        return torch.cuda.current_device() if torch.cuda.is_available() else 0
    if isinstance(device, int):
        return device
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"saber_amd filters run on a ROCm device only, got '{d}' (there is no CPU fallback)")
    return d.index if d.index is not None else torch.cuda.current_device()


def handle(device=None) -> Engine:
    idx = device_index(device)
    with _lock:
        if idx not in _handles:
            _handles[idx] = Engine.bare(idx)
        return _handles[idx]
