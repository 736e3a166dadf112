import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def lib():
    from saber_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def gpu_lib(lib):
    import torch
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    assert lib.saber_k_init(0) == 0, lib.saber_k_last_error()
    return lib


@pytest.fixture(scope="session")
def large_weights():
    from saber_amd.model_config import get_config
    from saber_amd.weights import seeded_weights
    cfg = get_config("large")
    return cfg, seeded_weights(cfg, 0)


@pytest.fixture(scope="session")
def engine(large_weights):
    from saber_amd.engine import Engine
    cfg, W = large_weights
    eng = Engine("large", device=0, weights=W, max_images=2, max_prompts=32)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def oracle_large(large_weights):
    from oracle import sam2_ref
    cfg, W = large_weights
    return cfg, sam2_ref.to_torch(W)
