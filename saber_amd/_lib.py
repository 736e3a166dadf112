"""ctypes binding of libsaber_amd.so (the C-ABI declared in include/saber_amd.h and
include/saber_amd_kernels.h).  The product path has no CPU fallback: if the HIP extension is
missing the import of this module raises, loudly."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsaber_amd.so")
CSRC = os.path.join(_HERE, "csrc")


class SaberAmdError(RuntimeError):
    pass


class SaberRangeError(SaberAmdError):
    """SABER_ERR_RANGE: the engine's overflow sentinel found NaN / inf in the 16-bit arithmetic (include/saber_amd.h:
    saber_engine_check_finite) - with fp16 operands, an activation beyond 65 504.  The call's results are invalid."""


# status codes of include/saber_amd.h
SABER_OK, SABER_ERR_INVALID, SABER_ERR_STATE, SABER_ERR_HIP, SABER_ERR_CAPACITY, SABER_ERR_RANGE = 0, -1, -2, -3, -4, -5


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into saber_amd/libsaber_amd.so (in-tree)."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise SaberAmdError("building libsaber_amd.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


class AmgParams(C.Structure):
    _fields_ = [("points_per_side", C.c_int), ("points_per_batch", C.c_int), ("pred_iou_thresh", C.c_float),
                ("stability_score_thresh", C.c_float), ("stability_score_offset", C.c_float), ("mask_threshold", C.c_float),
                ("box_nms_thresh", C.c_float), ("crop_n_layers", C.c_int), ("crop_nms_thresh", C.c_float),
                ("crop_overlap_ratio", C.c_float), ("crop_n_points_downscale_factor", C.c_int), ("use_m2m", C.c_int),
                ("multimask_output", C.c_int)]


class MaskMeta(C.Structure):
    _fields_ = [("area", C.c_int32), ("bbox_xywh", C.c_float * 4), ("predicted_iou", C.c_float),
                ("stability_score", C.c_float), ("point_xy", C.c_float * 2), ("crop_box_xywh", C.c_float * 4)]


class ProfileClass(C.Structure):
    _fields_ = [("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


PROFILE_CLASSES = ("gemm_bf16", "hiera_attention", "layernorm", "decoder_attention", "elementwise", "image_ops", "mask_post", "decoder_t2i", "decoder_i2t", "decoder_upscale", "gemm_mxfp8")

# name -> (restype, argtypes); mirrors include/*.h one-to-one (tests/test_abi.py checks the symbol list)
_vp, _i, _f, _i64p = C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_int64)
SIGNATURES = {
    "saber_engine_create": (_i, [_i, C.c_char_p, _i, _i, C.POINTER(_vp)]),
    "saber_engine_destroy": (None, [_vp]),
    "saber_last_error": (C.c_char_p, [_vp]),
    "saber_engine_set_weight": (_i, [_vp, C.c_char_p, _vp, _i64p, _i]),
    "saber_engine_finalize": (_i, [_vp]),
    "saber_prepare": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "saber_prepare_rgb": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "saber_encode": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_i), _i, _i, _vp]),
    "saber_get_features": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "saber_get_embed_tokens": (_i, [_vp, _i, _vp, _vp]),
    "saber_set_embed_tokens": (_i, [_vp, _i, _vp, _vp]),
    "saber_export_slots": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "saber_import_slots": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "saber_get_decoder_tokens": (_i, [_vp, _i, _vp, _vp]),
    "saber_decode_points": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "saber_decode_prompts": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "saber_amg_generate": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(AmgParams), _vp, _i, C.POINTER(MaskMeta), C.POINTER(_i), _vp]),
    "saber_amg_last_syncs": (_i, [_vp]),
    "saber_engine_set_iou_pruning": (_i, [_vp, _i]),
    "saber_engine_set_device_amg": (_i, [_vp, _i]),
    "saber_amg_last_pruning": (_i, [_vp, _vp, _vp]),
    "saber_engine_set_graphs": (_i, [_vp, _i]),
    "saber_engine_set_encoder_stream": (_i, [_vp, _vp]),
    "saber_engine_set_weight_format": (_i, [_vp, _i]),
    "saber_engine_set_precision": (_i, [_vp, _i]),
    "saber_engine_check_finite": (_i, [_vp, _vp]),
    "saber_engine_graph_stats": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "saber_label_plane": (_i, [_vp, _vp, C.POINTER(_i), _i, _i, _i, _vp, _vp]),
    "saber_mask_pair_intersections": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "saber_separate_masks": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, C.POINTER(_i), _vp]),
    "saber_smooth_labels": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_double, _vp, C.POINTER(_i), _vp]),
    "saber_gaussian_smoothing_3d": (_i, [_vp, _vp, _i, _i, _i, C.c_double, _vp, _vp]),
    "saber_classifier_create": (_i, [_vp, _i, C.POINTER(_vp)]),
    "saber_classifier_destroy": (None, [_vp]),
    "saber_classifier_set_weight": (_i, [_vp, C.c_char_p, _vp, _i64p, _i]),
    "saber_classifier_finalize": (_i, [_vp]),
    "saber_classifier_predict": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "saber_classifier_head": (_i, [_vp, _vp, _i, _vp, _vp]),
    "saber_classifier_get_crops": (_i, [_vp, _i, _vp, _vp, _vp]),
    "saber_profile_begin": (_i, [_vp]),
    "saber_profile_end": (_i, [_vp, C.POINTER(ProfileClass), _i]),
    "saber_encoder_flops": (C.c_double, [_vp]),
    "saber_decoder_flops_per_prompt": (C.c_double, []),
    "saber_k_last_error": (C.c_char_p, []),
    "saber_k_init": (_i, [_i]),
    "saber_k_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "saber_k_gemm_ld": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "saber_k_gemm_rowln": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "saber_k_layernorm": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp]),
    "saber_k_hiera_attention": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "saber_k_hiera_attention_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "saber_k_dec_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "saber_k_prepare": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "saber_k_mask_post": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "saber_k_perm_index": (_i, [_i, _i, _i]),
    "saber_k_set_operand_type": (_i, [_i]),
    "saber_k_stream_create_cu_range": (_i, [_i, _i, C.POINTER(_vp)]),
    "saber_k_stream_destroy": (_i, [_vp]),
    "saber_k_host_f32_to_f16": (None, [_vp, _vp, C.c_int64]),
    "saber_k_dec_i2t": (_i, [_vp, C.c_int64, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp]),
    "saber_k_dec_t2i": (_i, [_vp, C.c_int64, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "saber_k_rope": (_i, [_vp, C.c_int64, _i, _i, _i, _f, _vp, _vp, _vp]),
    "saber_k_softmax_rows": (_i, [_vp, C.c_int64, C.c_int64, _i, _f, _vp, C.c_int64, _vp]),
    "saber_k_conv3x3s2": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "saber_k_conv3x3s2_t": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "saber_k_paint_nearest": (_i, [_vp, _i, _i, _f, _i, _vp, _i, _i, _vp, _vp]),
    "saber_k_unpack_masks": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "saber_k_dwconv7": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "saber_k_dwconv7_t": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "saber_k_conv4x4s4": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "saber_k_resize_plane": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _f, _f, _vp]),
    "saber_k_box_nms": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "saber_k_gemm_mx": (_i, [_vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, _i, _i, _i, _vp]),
    "saber_k_quant_mx": (_i, [_vp, C.c_int64, _i, _vp, C.c_int64, _i, _vp, C.c_int64, C.c_int64, _vp]),
    "saber_k_ln_mx": (_i, [_vp, C.c_int64, _vp, _vp, _f, _i, _vp, C.c_int64, _i, _vp, C.c_int64, C.c_int64, _vp]),
    "saber_k_flash256": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, C.c_int64, _vp]),
    "saber_k_gauss_mirror": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_double, _vp]),
    "saber_k_axpy": (_i, [_vp, _vp, _vp, _f, C.c_int64, _i, _vp, _vp]),
    "saber_k_add_to_bf16": (_i, [_vp, _vp, _i, _vp, _vp, C.c_int64, _i, _vp]),
    "saber_k_bf16_to_f32": (_i, [_vp, C.c_int64, _vp, _vp]),
    "saber_k_gemm_batched": (_i, [_vp, _i, C.c_int64, _vp, _i, C.c_int64, _vp, _vp, _i, C.c_int64, _vp, _i, C.c_int64, _i, _i, _i, _i, _vp]),
    "saber_k_set_debug": (None, [_i]),
    "saber_k_set_stamp_buffer": (None, [_vp]),
}

_lib = None


def load():
    """dlopen the library (idempotent).  Raises SaberAmdError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("SABER_AMD_LIB"):         # development: A/B a differently built library (tools/ab_lib.sh)
        globals()["LIB_PATH"] = os.environ["SABER_AMD_LIB"]
    if not os.path.exists(LIB_PATH):
        raise SaberAmdError(f"{LIB_PATH} is missing: the MI355X HIP extension has not been built "
                            f"(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # torch first: its wheel bundles the HIP runtime this process must share with the engine (device pointers and streams cross the
    # C-ABI).  Loading libsaber_amd.so before torch would bind it to /opt/rocm's copy and leave two runtimes in one process.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    if os.environ.get("SABER_AMD_DEBUG"):      # development A/B switches of individual kernels (tools/ab_flag.sh); 0 in production
        lib.saber_k_set_debug(int(os.environ["SABER_AMD_DEBUG"], 0))
    return lib
