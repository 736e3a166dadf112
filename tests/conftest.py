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


@pytest.fixture(scope="session")
def image():
    """the 1024^2 test micrograph of the precision-mode tests (tests/test_gpu_exact.py, tests/test_gpu_fp16.py)"""
    import numpy as np
    rng = np.random.default_rng(7)
    img = rng.uniform(0, 1, (1024, 1024)).astype(np.float32)
    yy, xx = np.mgrid[:1024, :1024]
    for _ in range(10):
        cy, cx = rng.integers(100, 924, 2)
        r = rng.integers(30, 120)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] *= 0.3
    return img


@pytest.fixture(scope="session")
def oracle_feats(image, oracle_large):
    """the fp32 CPU oracle's Hiera-L features of `image` (one ~15-s oracle pass shared by every module that compares against it)"""
    import numpy as np
    import torch
    from oracle import sam2_ref
    cfg, W = oracle_large
    with torch.no_grad():
        return sam2_ref.encode_image(W, cfg, sam2_ref.sam2_transforms(np.repeat(image[..., None], 3, 2)))


def amg_case(name):
    """Masks of one oracle AMG case from tests/golden/amg_cases_large_seed0.npz (oracle/make_golden_amg_cases.py): the quarter-resolution
    samples [2::4, 2::4] of every mask, list of {"segmentation": bool array}; compare with the same samples of the engine's masks"""
    import os
    import numpy as np
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "amg_cases_large_seed0.npz"))
    W = int(G[name + "_width"])
    seg = np.unpackbits(G[name + "_bits"], axis=-1)[..., :W].astype(bool)
    return [{"segmentation": m} for m in seg]
