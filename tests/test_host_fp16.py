"""CPU: the host-side fp32 -> IEEE half conversion saber_engine_finalize applies to the weights in SABER_PRECISION_FP16 (csrc/engine.hip:
saber_host_f2h, exposed as saber_k_host_f32_to_f16) against numpy.float16 - round to nearest even over normals, subnormals, ties, overflow."""
import ctypes as C

import numpy as np


def test_host_f32_to_f16_matches_numpy(lib):
    rng = np.random.default_rng(0)
    parts = [
        rng.standard_normal(200000).astype(np.float32),
        (rng.standard_normal(100000) * 1e-5).astype(np.float32),                       # fp16 subnormals
        (rng.standard_normal(100000) * 1e4).astype(np.float32),
        np.float32(2.0) ** rng.integers(-30, 17, 50000).astype(np.float32) * rng.choice([-1.0, 1.0], 50000).astype(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e9, -1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0000001, 2.0 ** -14, 6.0975552e-05,
                  np.inf, -np.inf], dtype=np.float32),
    ]
    # exact ties between two fp16 neighbours (mantissa bit 12 set, lower bits clear): round to even
    base = rng.uniform(0.001, 1000.0, 50000).astype(np.float16).astype(np.float32)
    u = base.view(np.uint32)
    ties = ((u & ~np.uint32(0x1FFF)) | np.uint32(0x1000)).view(np.float32)
    parts.append(ties)
    x = np.ascontiguousarray(np.concatenate(parts))
    out = np.empty(x.size, dtype=np.uint16)
    lib.saber_k_host_f32_to_f16(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), x.size)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    bad = np.nonzero(out != ref)[0]
    assert bad.size == 0, (x[bad[:5]], out[bad[:5]], ref[bad[:5]])
    nan = np.array([np.nan], dtype=np.float32)
    o = np.empty(1, dtype=np.uint16)
    lib.saber_k_host_f32_to_f16(nan.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), 1)
    assert np.isnan(o.view(np.float16)[0])
