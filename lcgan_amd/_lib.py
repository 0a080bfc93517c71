"""ctypes binding of liblcgan_hip.so (C ABI: include/lcgan_hip.h).

There is NO CPU fallback: if the library is missing it is built with hipcc, and if that fails (or a kernel
returns a non-zero status) a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblcgan_hip.so")

P, I, F, D, LL = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_longlong

# name -> argtypes (every entry point returns int); must mirror include/lcgan_hip.h exactly
SIGNATURES = {
    "lcgan_conv_weight_prep": [P, I, I, I, F, I, P, I, P, P],
    "lcgan_conv_wgrad_unprep": [P, I, I, I, F, I, P, P, P, P],
    "lcgan_conv_weight_prep_group": [P, P, P, I, P, P, D, P],
    "lcgan_conv_fwd": [P, P, P, I, I, I, I, I, I, I, I, P, P, P, F, I, F, P, I, P, P, P, I, P],
    "lcgan_conv_fwd_m": [P, P, P, I, I, I, I, I, I, I, I, P, P, P, F, I, F, P, I, P, P, P, P, P, I, P],
    "lcgan_conv_bwd_data": [P, P, P, I, I, I, I, I, I, I, I, P, P, P, F, I, F, P, I, P, P, I, P],
    "lcgan_conv_wgrad": [P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, P, I, P],
    "lcgan_conv_wgrad_fused": [P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, P, I, F, I, P, P, P, P],
    "lcgan_box3_act": [P, P, I, I, I, I, I, F, I, P],
    "lcgan_box3_act_bwd": [P, P, P, I, I, I, I, I, F, I, P],
    "lcgan_box3_actbwd_reduce": [P, P, P, P, I, I, I, I, I, I, F, I, P],
    "lcgan_box3_actbwd_reduce_m": [P, P, P, P, P, I, I, I, I, I, I, F, I, P],
    "lcgan_up2box": [P, P, P, I, I, I, I, I, P],
    "lcgan_up2box_bwd": [P, P, I, I, I, I, I, P],
    "lcgan_avgpool2": [P, P, I, I, I, I, I, P],
    "lcgan_avgpool2_bwd": [P, P, I, I, I, I, I, P],
    "lcgan_act_bwd_reduce": [P, P, P, P, F, P, P, I, I, I, I, I, F, I, P],
    "lcgan_act_bwd_reduce_m": [P, P, P, P, P, F, P, P, I, I, I, I, I, F, I, P],
    "lcgan_act_bwd_reduce_s": [P, P, P, P, P, P, F, P, P, I, I, I, I, I, F, I, P],
    "lcgan_scale_reduce": [P, P, P, P, I, I, I, I, P],
    "lcgan_scale_reduce_res": [P, P, P, P, P, I, I, I, I, P],
    "lcgan_warp_fwd": [P, P, P, I, I, I, I, F, I, P],
    "lcgan_warp_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, I, F, I, P],
    "lcgan_cast_from_f32": [P, P, LL, I, P],
    "lcgan_mbstd_fwd": [P, P, I, I, I, I, I, I, P],
    "lcgan_mbstd_bwd": [P, P, P, I, I, I, I, I, I, P],
    "lcgan_mbstd_bwd2": [P, P, P, P, P, I, I, I, I, I, I, P],
    "lcgan_rgb_expand": [P, P, P, F, P, I, I, I, I, I, I, F, P, I, I, P],
    "lcgan_rgb_reduce": [P, P, P, F, P, I, I, I, I, I, P],
    "lcgan_rgb_wgrad": [P, P, P, I, I, I, I, I, P],
    "lcgan_rgb_expand_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, I, F, I, P],
    "lcgan_rgb_expand_bwd_r": [P, P, P, P, P, F, I, P, P, P, I, I, I, I, I, I, F, I, P],
    "lcgan_rgb_reduce_bwd_act": [P, P, P, P, F, P, P, P, P, I, I, I, I, I, I, F, I, P],
    "lcgan_rgb_reduce_bwd_act_s": [P, P, P, P, F, P, P, P, P, P, I, I, I, I, I, I, F, I, P],
    "lcgan_flow_col2im": [P, P, P, P, I, I, I, I, I, P],
    "lcgan_flow_im2col": [P, P, P, I, I, I, I, I, P],
    "lcgan_nchw_to_nhwc": [P, P, I, I, I, I, I, I, P],
    "lcgan_nhwc_to_nchw": [P, P, I, I, I, I, I, I, P],
    "lcgan_linear_fwd": [P, P, P, P, I, I, I, F, F, I, F, P],
    "lcgan_linear_bwd_data": [P, P, P, I, I, I, F, P],
    "lcgan_linear_wgrad": [P, P, P, I, I, I, F, P],
    "lcgan_colsum": [P, P, I, I, F, P],
    "lcgan_linear_wgrad_bias": [P, P, P, P, I, I, I, F, F, P],
    "lcgan_linear_group_fwd": [P, P, P, P, P, P, P, I, I, I, I, F, P],
    "lcgan_linear_group_bwd": [P, P, P, P, P, P, I, I, I, P, P, P, P],
    "lcgan_linear_multi_fwd": [P, P, P, P, P, P, P, P, I, I, I, F, P],
    "lcgan_linear_multi_bwd": [P, P, P, P, P, P, P, I, I, P, P, P, P],
    "lcgan_act_bwd_f32": [P, P, P, LL, I, F, P],
    "lcgan_demod_fwd": [P, P, P, I, I, I, I, F, P],
    "lcgan_demod_group": [P, P, P, P, P, P, I, I, F, P],
    "lcgan_demod_bwd": [P, P, P, P, P, P, I, I, I, I, P],
    "lcgan_bce_fwd": [P, I, I, P, P],
    "lcgan_bce_bwd": [P, I, I, P, P, P],
    "lcgan_contrastive_fwd": [P, P, P, I, I, F, P, P, P],
    "lcgan_contrastive_bwd": [P, P, P, P, P, I, I, F, P, P, P, P],
    "lcgan_l2norm_fwd": [P, P, P, I, I, F, P],
    "lcgan_l2norm_bwd": [P, P, P, P, I, I, P],
    "lcgan_powsum": [P, LL, I, F, P, P],
    "lcgan_powsum_bwd": [P, LL, I, F, P, P, P],
    "lcgan_qr_householder": [P, P, P, I, I, P],
    "lcgan_avg_latent": [P, P, I, I, F, P],
    "lcgan_multi_tensor": [P, P, P, I, I, F, F, F, D, P],
    "lcgan_make_views": [P, P, P, P, P, I, I, P],
    "lcgan_conv_weight_prep_fp8": [P, I, I, I, F, I, P, P, P],
    "lcgan_conv_fwd_fp8": [P, P, P, P, I, I, I, I, I, I, I, I, P, P, P, F, I, F, P, I, P],
    "lcgan_conv_bwd_data_fp8": [P, P, P, P, I, I, I, I, I, I, I, I, P, P, P, F, I, F, P, I, P],
    "lcgan_set_option": [I, I],
    "lcgan_prof_enable": [I],
    "lcgan_prof_collect": [P, P, P, P],
    "lcgan_prof_active": [],
    "lcgan_prof_dump": [P],
}

_lib = None


def load(build_if_missing: bool = True):
    """Returns the loaded CDLL with argtypes set; raises RuntimeError when it cannot be had."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _build
    if _build._stale():                      # missing, or built from other sources than the ones on disk (content hash)
        if not build_if_missing:
            raise RuntimeError(f"{LIB_PATH} is missing or stale (run `python -m lcgan_amd.build`); there is no CPU fallback")
        try:
            _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise RuntimeError(f"could not build {LIB_PATH} with hipcc: {e}; there is no CPU fallback") from e
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = lib
    return lib


def check(status: int, name: str):
    if status != 0:
        raise RuntimeError(f"{name} failed with status {status} ({'invalid argument' if status == -1 else 'HIP launch error'})")
