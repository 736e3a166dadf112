"""CPU: the C-ABI library loads and exports every symbol the headers declare; host-only entry points behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in ("saber_amd.h", "saber_amd_kernels.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"\b(saber_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from saber_amd import _lib
    decl = declared_symbols()
    assert len(decl) >= 24
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert sorted(_lib.SIGNATURES) == decl, "ctypes table and headers disagree"


def test_struct_layouts_match_header():
    from saber_amd import _lib
    assert C.sizeof(_lib.AmgParams) == 13 * 4
    assert C.sizeof(_lib.MaskMeta) == 4 + 16 + 4 + 4 + 8 + 16
    assert C.sizeof(_lib.ProfileClass) == 32


def test_create_rejects_bad_configuration_without_a_gpu(lib):
    h = C.c_void_p()
    assert lib.saber_engine_create(0, b"huge", 1, 8, C.byref(h)) == -1
    assert b"tiny/small/base/large" in lib.saber_last_error(None)
    for trunk in (b"tiny", b"small", b"base"):   # every trunk SABER can name is built; without a GPU creation stops at the device probe
        st = lib.saber_engine_create(0, trunk, 1, 8, C.byref(h))
        if st == 0:
            lib.saber_engine_destroy(h)          # a GPU is present
        else:
            assert st == -3 and b"HIP device" in lib.saber_last_error(None)
    assert lib.saber_engine_create(0, b"large", 0, 8, C.byref(h)) == -1
    assert lib.saber_decoder_flops_per_prompt() == pytest.approx(3.639e9)


def test_engine_wrapper_fails_loudly_without_device(monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("has a GPU")
    from saber_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine("large")
    with pytest.raises(ValueError):
        Engine("huge")


def test_filters_fail_loudly_without_device():
    """the volume filters have no CPU path either: numpy in, but the arithmetic only exists in the HIP library"""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("has a GPU")
    from saber_amd.filters import fast_3d_gaussian_smoothing, gaussian_smoothing_3d
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fast_3d_gaussian_smoothing(np.ones((4, 8, 8), np.uint16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gaussian_smoothing_3d(np.ones((4, 8, 8), bool), 1.0)
    with pytest.raises(ValueError):                     # argument checks come first, as in the reference
        fast_3d_gaussian_smoothing(np.ones((8, 8), np.uint16))
    with pytest.raises(RuntimeError, match="ROCm device only"):
        gaussian_smoothing_3d(np.ones((4, 8, 8), bool), 1.0, device="cpu")


def test_token_order_tiles_every_hiera_l_window(lib):
    """every Hiera-L attention window is a contiguous aligned run of rows; 2x2 pooling groups are 4 consecutive rows"""
    for stage, win in ((0, 8), (1, 4), (2, 16), (3, 8)):
        g = 256 >> stage
        idx = np.array([[lib.saber_k_perm_index(y, x, stage) for x in range(g)] for y in range(g)])
        assert sorted(idx.ravel().tolist()) == list(range(g * g))
        for wy in range(0, g, win):
            for wx in range(0, g, win):
                w = idx[wy:wy + win, wx:wx + win].ravel()
                assert w.max() - w.min() == win * win - 1 and w.min() % (win * win) == 0
        if stage < 3:
            q = idx.reshape(g // 2, 2, g // 2, 2).transpose(0, 2, 1, 3).reshape(-1, 4)
            assert (q == q[:, :1] + np.arange(4)).all() and (q[:, 0] % 4 == 0).all()
            nxt = np.array([[lib.saber_k_perm_index(y, x, stage + 1) for x in range(g // 2)] for y in range(g // 2)])
            assert np.array_equal(q[:, 0].reshape(g // 2, g // 2) >> 2, nxt)  # pooled row = next stage's index
