"""ctypes binding of librotmvgaze_hip.so (the C ABI declared in include/rotmvgaze.h).

There is no fallback: if the library is missing or fails to load, ``lib()`` raises.  Build it with
``python rot-mvgaze_amd/csrc/build.py`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librotmvgaze_hip.so")

K_FAMILIES = 18
ABI_VERSION = 10


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("groups", "n", "h", "w", "cin", "cout", "r", "s", "stride", "pad", "ho", "wo")]

    @classmethod
    def make(cls, groups, n, h, w, cin, cout, k, stride, pad):
        ho = (h + 2 * pad - k) // stride + 1
        wo = (w + 2 * pad - k) // stride + 1
        return cls(groups, n, h, w, cin, cout, k, k, stride, pad, ho, wo)

    @classmethod
    def linear(cls, rows, fin, fout):
        return cls(1, rows, 1, 1, fin, fout, 1, 1, 1, 0, 1, 1)


class ProfEntry(C.Structure):
    _fields_ = [("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


_P = C.c_void_p
_I = C.c_int
_I64 = C.c_int64
_F = C.c_float
_D = C.POINTER(ConvDesc)

# name -> (restype, argtypes); must list every function include/rotmvgaze.h declares
SIGNATURES = {
    "mvg_abi_version": (_I, []),
    "mvg_last_error": (C.c_char_p, []),
    "mvg_device_cus": (_I, []),
    "mvg_set_scratch": (_I, [_P, C.c_size_t, _P]),
    "mvg_scratch_bytes": (C.c_size_t, []),
    "mvg_set_reserved_cus": (_I, [_I]),
    "mvg_stream_create_low_priority": (_P, []),
    "mvg_prof_enable": (_I, [_I]),
    "mvg_prof_reset": (_I, []),
    "mvg_prof_collect": (_I, [C.POINTER(ProfEntry)]),
    "mvg_prof_family_name": (C.c_char_p, [_I]),
    "mvg_conv_fprop": (_I, [_D, _P, _P, _P, _P, _I, _P, _P]),
    "mvg_conv_fprop_affine": (_I, [_D, _P, _P, _P, _P, _P, _P, _I, _P]),
    "mvg_conv_stats_partials": (_I, [_D, C.POINTER(C.c_int32)]),
    "mvg_conv_dgrad": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "mvg_conv_wgrad": (_I, [_D, _P, _P, _P, _P, _I, _I, _P]),
    "mvg_conv_wgrad_splits": (_I, [_D]),
    "mvg_linear_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _I, _I, _P]),
    "mvg_fuser_fprop": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, C.c_size_t, _P]),
    "mvg_fuser_wgrad": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "mvg_linear_workspace_floats": (C.c_size_t, [_I, _I, _I]),
    "mvg_linear_fprop": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _P, C.c_size_t, _P]),
    "mvg_linear_dgrad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, C.c_size_t, _P]),
    "mvg_bn_finalize": (_I, [_P, _I, _I, _I, _I64, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "mvg_bn_eval_affine": (_I, [_I, _I, _P, _P, _P, _P, _F, _P, _P, _P]),
    "mvg_bn_apply": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _I, _I64, _I, _P]),
    "mvg_bn_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "mvg_split_f32": (_I, [_P, _P, _I64, _F, _P]),
    "mvg_merge_sp": (_I, [_P, _P, _I64, _F, _P]),
    "mvg_split_weights": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "mvg_weights_prep_batch": (_I, [_P, _I, _I, _I, _P]),
    "mvg_conv_stats_partials_split": (_I, [_D, C.POINTER(C.c_int32)]),
    "mvg_conv_fprop_split": (_I, [_D, _P, _P, _P, _P, _P, _P, _P]),
    "mvg_conv_fprop_split_affine": (_I, [_D, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P]),
    "mvg_conv_dgrad_split": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mvg_bn_apply_split": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _I64, _I, _P]),
    "mvg_bn_bwd_apply_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _P, _P, _P, _I, _P]),
    "mvg_bn_bwd_reduce_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "mvg_bn_relu_maxpool_fwd_split": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mvg_avgpool_fwd_split": (_I, [_P, _P, _I, _I, _I, _P]),
    "mvg_stem_rowwindow_split": (_I, [_P, _P, _I64, _I, _I, _P]),
    "mvg_stem_rowwindow_split_nchw": (_I, [_P, _P, _I64, _I, _I, _P]),
    "mvg_stem_fprop_split": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "mvg_stem_wgrad_splits_split": (_I, [_D]),
    "mvg_stem_wgrad_split": (_I, [_D, _P, _P, _P, _P, _P, _I, _I, _P]),
    "mvg_bn_relu_maxpool_bwd_reduce_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "mvg_bn_relu_maxpool_bwd_apply_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "mvg_conv_dgrad_bn_partials_split": (_I, [_D]),
    "mvg_conv_dgrad_split_bnreduce": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "mvg_conv_wgrad_splits_split": (_I, [_D]),
    "mvg_conv_wgrad_split": (_I, [_D, _P, _P, _P, _P, _P, _I, _I, _P]),
    "mvg_conv_wgrad_split_slabs": (_I, [_D, _P, _P, _P, _P, _I, _P]),
    "mvg_wgrad_reduce_batch": (_I, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), _I, _P]),
    "mvg_bn_apply_bits": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _P]),
    "mvg_bn_bwd_reduce_bits": (_I, [_P, _P, _P, _P, _P, _I, _I64, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "mvg_bn_apply_bits_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _P]),
    "mvg_bn_bwd_reduce_bits_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I64, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "mvg_bn_bwd_workspace_floats": (C.c_size_t, [_I, _I64, _I]),
    "mvg_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _P, _P, _P]),
    "mvg_maxpool3x3s2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mvg_maxpool3x3s2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "mvg_bn_relu_maxpool_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mvg_bn_relu_maxpool_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P,
                                            _P]),
    "mvg_bn_relu_maxpool_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mvg_avgpool_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "mvg_avgpool_bwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "mvg_nchw_to_nhwc4": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mvg_nhwc4_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mvg_preprocess_u8hwc": (_I, [_P, _P, _I, _I, _I, _F, _F, _F, _F, _F, _F, _I, _P]),
    "mvg_gaze_lp_loss": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "mvg_preprocess_u8hwc_resize": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _F, _F, _F, _I, _P]),
    "mvg_multi_erase_nchw": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mvg_rotation_matrix_2d": (_I, [_P, _P, _I, _I, _P]),
    "mvg_relative_rotation": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "mvg_rotcat_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mvg_rotcat_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mvg_rotcat_ext_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mvg_rotcat_ext_bwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "mvg_ibn_scales": (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _P, _I, _I, _I, _P]),
    "mvg_paircat_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "mvg_paircat_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "mvg_segment_sum": (_I, [_P, _I64, _I, _P, _P, _I, _I, _I, _I, _P]),
    "mvg_axpby": (_I, [_P, _P, _F, _F, _I64, _P]),
    "mvg_scale_by": (_I, [_P, _P, _P, _I64, _P]),
    "mvg_linear_skinny_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "mvg_linear_skinny_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "mvg_adam_step": (_I, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _I, _P]),
    "mvg_gaze_angular_loss": (_I, [_P, _P, _I, _F, _P, _I, _P, _P, _P]),
    "mvg_gaze_angular_loss_multi": (_I, [_P, _P, _I, _I, _I, C.POINTER(C.c_float), _P, _P, _P]),
    "mvg_adam_step_dev": (_I, [_P, _P, _P, _P, _I64, _P, _P, _F, _F, _F, _F, _P]),
    # the fusion block on the split kernels
    "mvg_absmax_multi": (_I, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), _I, _P]),
    "mvg_fuse_build_split": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "mvg_fuse_unbuild": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mvg_linear_fprop_split": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P]),
    "mvg_linear_dgrad_split": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mvg_linear_wgrad_split": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "mvg_split_colsum": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _P]),
    # bf16 storage path (same argument lists as the fp32 entry points)
    "mvg_cast_weights_bf16": (_I, [_D, _P, _I, _P, _P, _P]),
    "mvg_conv_fprop_bf16": (_I, [_D, _P, _P, _P, _P, _I, _P, _P]),
    "mvg_conv_stats_partials_bf16": (_I, [_D, C.POINTER(C.c_int32)]),
    "mvg_conv_dgrad_bf16": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "mvg_conv_dgrad_bn_partials_bf16": (_I, [_D]),
    "mvg_conv_dgrad_bf16_bnreduce": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "mvg_conv_wgrad_bf16": (_I, [_D, _P, _P, _P, _P, _I, _I, _P]),
    "mvg_stem_rowwindow_bf16": (_I, [_P, _P, _I64, _I, _I, _P]),
    "mvg_stem_fprop_bf16": (_I, [_D, _P, _P, _P, _P, _P]),
    "mvg_stem_wgrad_splits_bf16": (_I, [_D]),
    "mvg_stem_wgrad_bf16": (_I, [_D, _P, _P, _P, _P, _I, _I, _P]),
    "mvg_conv_wgrad_splits_bf16": (_I, [_D]),
    "mvg_conv_wgrad_bf16_slabs": (_I, [_D, _P, _P, _P, _I, _P]),
    "mvg_linear_fprop_mixed": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "mvg_linear_dgrad_mixed": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "mvg_linear_wgrad_mixed": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _I, _I, _P]),
    "mvg_bn_apply_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _I, _I64, _I, _P]),
    "mvg_bn_bwd_reduce_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "mvg_bn_bwd_apply_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _I, _P, _P, _P]),
    "mvg_bn_relu_maxpool_fwd_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mvg_bn_relu_maxpool_bwd_reduce_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I,
                                                 _P, _P]),
    "mvg_bn_relu_maxpool_bwd_apply_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "mvg_avgpool_fwd_bf16": (_I, [_P, _P, _I, _I, _I, _P]),
    "mvg_avgpool_bwd_bf16": (_I, [_P, _P, _I, _I, _I, _P]),
    "mvg_nchw_to_nhwc8_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mvg_mt19937_seed": (_I, [_P, C.c_uint64]),
    "mvg_pair_index_build": (_I64, [_P, _P, _I, _I, _P, _I64]),
}

_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the library; raise if it is not there - no CPU fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is required (python rot-mvgaze_amd/csrc/build.py); "
            "this package has no CPU fallback")
    try:
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)           # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if l.mvg_abi_version() != ABI_VERSION:
        raise RuntimeError(f"librotmvgaze_hip.so ABI {l.mvg_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = l
    return l


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().mvg_last_error()
        raise RuntimeError(f"librotmvgaze_hip {what} failed (rc={rc}): {msg.decode() if msg else '?'}")
