"""ctypes binding of libsatflow_hip.so (the C ABI declared in include/satflow_hip.h).

The library is the product; there is no fallback.  If it is missing, importing
the symbols raises with the build command.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_LIB_PATH = os.environ.get("SATFLOW_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libsatflow_hip.so")

SF_F32, SF_BF16, SF_F16, SF_F32E = 0, 1, 2, 3
SF_EPI_LINEAR, SF_EPI_SIGMOID = 0, 1
SF_CPAD = 16
ABI_VERSION = 8  # == SF_ABI_VERSION of include/satflow_hip.h; bumped on every signature / workspace-layout change


class sfTensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("c", C.c_int32), ("stride", C.c_int32), ("idiv", C.c_int32), ("imod", C.c_int32),
                ("dtype", C.c_int32), ("amax", C.c_void_p)]


class sfBlock(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64), ("src_stride", C.c_int64),
                ("dst_stride", C.c_int64), ("transpose", C.c_int64)]


# name -> (restype, argtypes); mirrors include/satflow_hip.h one to one
_i32, _i64, _vp, _sz = C.c_int32, C.c_int64, C.c_void_p, C.c_size_t
PROTOTYPES = {
    "sf_abi_version": (C.c_int, []),
    "sf_last_error_string": (C.c_char_p, []),
    "sf_amax": (C.c_int, [sfTensor, _i64, _vp, _vp, _i32, _vp]),
    "sf_conv3x3_packed_elems": (_sz, [_i32, _i32]),
    "sf_conv3x3_pack_weights": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "sf_conv3x3_fwd": (C.c_int, [sfTensor, sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, sfTensor, _i32, _vp]),
    "sf_conv3x3_fwd_splitk_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "sf_conv3x3_fwd_splitk": (C.c_int, [sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, sfTensor, _vp, _sz, _i32, _vp]),
    "sf_conv3x3_stats_tiles": (_i32, [_i32, _i32]),
    "sf_conv3x3_fwd_stats": (C.c_int, [sfTensor, sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, sfTensor, _vp, _i32, _vp]),
    "sf_conv3x3_fold_pack": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "sf_conv3x3_fwd_folded": (C.c_int, [sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, sfTensor, _vp, _i32, _vp]),
    "sf_conv3x3_fwd_folded_pool_supported": (_i32, [_i32] * 8),
    "sf_conv3x3_fwd_folded_pool": (C.c_int, [sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, sfTensor, _i32, _i32, _vp, _i32, _vp]),
    "sf_convlstm_cell_fwd": (
        C.c_int,
        [sfTensor, sfTensor, sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, sfTensor, sfTensor, sfTensor, _i32, _vp],
    ),
    "sf_convlstm_cell_bwd_gates": (
        C.c_int,
        [sfTensor, sfTensor, sfTensor, sfTensor, sfTensor, sfTensor, sfTensor, _i64, _i32, sfTensor, sfTensor, _i32, _vp],
    ),
    "sf_conv3x3_bwd_weight_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "sf_conv3x3_bwd_weight": (
        C.c_int,
        [sfTensor, sfTensor, sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _sz, _i32, _vp],
    ),
    "sf_conv3x3_bwd_weight_folded_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "sf_conv3x3_bwd_weight_folded": (
        C.c_int,
        [sfTensor, sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp],
    ),
    "sf_conv3x3_bwd_weight_folded_sparse24_supported": (_i32, [_i32] * 6),
    "sf_conv3x3_bwd_weight_folded_sparse24": (
        C.c_int,
        [sfTensor, sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, sfTensor, _vp, _i32, _i32,
         _vp, _sz, _i32, _vp],
    ),
    "sf_conv3x3_bwd_data_bn": (C.c_int, [sfTensor, _i32, _i32, _i32, _vp, _i32, _i32, sfTensor, _vp, _i32, sfTensor, _i32, _vp]),
    "sf_batchnorm_train_bwd_coef": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "sf_nchw_to_nhwc": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, sfTensor, _i32, _vp]),
    "sf_metnet_preprocess_fwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, sfTensor, _i32, _vp]),
    "sf_metnet_preprocess_bwd": (C.c_int, [sfTensor, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "sf_maxpool2_fwd": (C.c_int, [sfTensor, _i64, _i32, _i32, sfTensor, _i32, _i32, _i32, _vp]),
    "sf_maxpool2_bwd": (C.c_int, [sfTensor, sfTensor, _i64, _i32, _i32, sfTensor, _i32, _i32, _i32, _vp]),
    "sf_maxpool2_dropout_fwd": (C.c_int, [sfTensor, _i64, _i32, _i32, sfTensor, _i32, _i32, C.c_float, C.c_float, _i64, C.c_uint64, C.c_uint64,
                                          _i32, _vp]),
    "sf_maxpool2_dropout_bwd": (C.c_int, [sfTensor, sfTensor, _i64, _i32, _i32, sfTensor, _i32, _i32, C.c_float, C.c_float, _i64, C.c_uint64,
                                          C.c_uint64, _i32, _vp]),
    "sf_maxpool2_route_fwd": (C.c_int, [sfTensor, _i64, _i32, _i32, sfTensor, _i32, _i32, C.c_float, C.c_float, _i64, C.c_uint64, C.c_uint64, _vp,
                                        _i32, _vp]),
    "sf_maxpool2_route_bwd": (C.c_int, [_vp, sfTensor, _i64, _i32, _i32, sfTensor, _i32, _i32, C.c_float, C.c_float, _i64, C.c_uint64, C.c_uint64,
                                        _vp, _i32, _vp]),
    "sf_leadtime_pool_workspace_floats": (_sz, [_i32, _i32]),
    "sf_stlstm_gates_fwd": (C.c_int, [sfTensor] * 5 + [_i64, _i32, C.c_float] + [sfTensor] * 7 + [_i32, _vp]),
    "sf_stlstm_gates_bwd": (C.c_int, [sfTensor] * 9 + [_i64, _i32] + [sfTensor] * 5 + [_i32, _vp]),
    "sf_layernorm_chw_fwd": (C.c_int, [sfTensor, _i64, _i64, _i32, _i32, _i32, _vp, _vp, C.c_float, _vp, sfTensor, _vp]),
    "sf_layernorm_chw_bwd": (C.c_int, [sfTensor, sfTensor, _i64, _i64, _i32, _i32, _i32, _vp, C.c_float, _vp, _vp, sfTensor, _vp, _vp, _vp]),
    "sf_stlstm_out_fwd": (C.c_int, [sfTensor] * 3 + [_i64, _i32, sfTensor, sfTensor, _i32, _vp]),
    "sf_stlstm_out_bwd": (C.c_int, [sfTensor, sfTensor, _i64, _i32, sfTensor, sfTensor, _i32, _vp]),
    "sf_leadtime_pool_fwd": (C.c_int, [sfTensor, _i64, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, sfTensor, _i32, _vp]),
    "sf_leadtime_pool_stats_tiles": (_i32, []),
    "sf_leadtime_pool_fwd_stats": (C.c_int, [sfTensor, _i64, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, sfTensor, _vp, _i32, _vp]),
    "sf_leadtime_pool_bwd": (C.c_int, [sfTensor, sfTensor, _i64, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, sfTensor, _vp, _i32, _vp]),
    "sf_batchnorm_train_fwd": (
        C.c_int,
        [sfTensor, _i64, _i32, _i32, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, sfTensor, _i32, _vp],
    ),
    "sf_batchnorm_train_fwd_stats": (
        C.c_int,
        [sfTensor, _i64, _i32, _i32, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, sfTensor, _i32, _vp],
    ),
    "sf_batchnorm_eval_fwd": (C.c_int, [sfTensor, _i64, _i32, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, sfTensor, _i32, _vp]),
    "sf_batchnorm_eval_bwd": (C.c_int, [sfTensor, sfTensor, _i64, _i32, _vp, C.c_float, _vp, _vp, _vp, _vp, sfTensor, _vp, _vp, _i32, _vp]),
    "sf_batchnorm_train_bwd": (C.c_int, [sfTensor, sfTensor, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, sfTensor, _vp, _vp, _i32, _vp]),
    "sf_convgru_step_fwd": (C.c_int, [sfTensor, sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, sfTensor, sfTensor, _i32, _vp]),
    "sf_convgru_seq_fwd": (C.c_int, [sfTensor, sfTensor, _i32, _i32, _i32, _i32, _vp, _vp, _i32, sfTensor, sfTensor, _vp, _sz, _i32, _vp]),
    "sf_convgru_seq_fwd_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "sf_convgru_seq_bwd": (C.c_int, [sfTensor, sfTensor, sfTensor, sfTensor, _i32, _i32, _i32, _i32, _vp, _i32, sfTensor, sfTensor, _vp, _sz, _i32, _vp]),
    "sf_convgru_seq_bwd_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "sf_convgru_bwd_gates": (
        C.c_int,
        [sfTensor, sfTensor, sfTensor, sfTensor, sfTensor, _i64, _i32, sfTensor, sfTensor, sfTensor, _i32, _vp],
    ),
    "sf_linear_fwd": (C.c_int, [sfTensor, _i64, _vp, _i32, _vp, sfTensor, _i32, _vp]),
    "sf_linear_bwd_weight_workspace_bytes": (_sz, [_i32, _i32, _i64]),
    "sf_linear_bwd_weight": (C.c_int, [sfTensor, _i32, sfTensor, _i64, _vp, _vp, _vp, _sz, _i32, _vp]),
    "sf_axial_attention_core_fwd": (C.c_int, [sfTensor, _i64, _i32, _i32, _i32, _i32, _i32, sfTensor, _i32, _vp]),
    "sf_axial_attention_core_bwd": (C.c_int, [sfTensor, sfTensor, _i64, _i32, _i32, _i32, _i32, _i32, sfTensor, _i32, _vp]),
    "sf_mse_loss": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "sf_dropout2": (C.c_int, [_vp, _i64, C.c_float, C.c_float, _i64, C.c_uint64, C.c_uint64, _vp, _vp]),
    "sf_copy_blocks": (C.c_int, [C.POINTER(sfBlock), _i32, _vp]),
    "sf_dropout2_bf16": (C.c_int, [_vp, _i64, C.c_float, C.c_float, _i64, C.c_uint64, C.c_uint64, _vp, _vp]),
    "sf_conv2d_fwd": (C.c_int, [sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_float, sfTensor, _i32, _vp]),
    "sf_conv2d_bwd_data": (C.c_int, [sfTensor, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, sfTensor, _i32, _vp]),
    "sf_conv2d_bwd_weight_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "sf_conv2d_bwd_weight": (C.c_int, [sfTensor, sfTensor, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _sz, _i32, _vp]),
    "sf_leaky_relu": (C.c_int, [_vp, _vp, _i64, C.c_float, _vp, _vp]),
    "sf_sigmoid_bwd": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "sf_l1_loss": (C.c_int, [sfTensor, sfTensor, _i64, _i32, _i32, sfTensor, _vp, _vp, _vp]),
    "sf_bce_logits_loss": (C.c_int, [sfTensor, C.c_float, C.c_float, _i64, _i32, _i32, sfTensor, _vp, _vp, _vp]),
    "sf_gan_loss": (C.c_int, [_i32, sfTensor, C.c_float, C.c_float, _i64, _i32, _i32, sfTensor, _vp, _vp, _vp]),
    "sf_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _i64, C.c_float, C.c_float, C.c_float, C.c_float, _i32, C.c_float, _vp]),
    "sf_spectral_norm_workspace_floats": (_sz, [_i32, _i32]),
    "sf_spectral_norm_fwd": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "sf_spectral_norm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "sf_pool2": (C.c_int, [sfTensor, _i64, _i32, _i32, _i32, _i32, C.c_float, sfTensor, sfTensor, _vp]),
    "sf_expand2": (C.c_int, [sfTensor, _i64, _i32, _i32, _i32, _i32, C.c_float, sfTensor, _vp]),
    "sf_time_stack3_fwd": (C.c_int, [sfTensor, _i32, _i64, sfTensor, _vp]),
    "sf_time_stack3_bwd": (C.c_int, [sfTensor, _i32, _i64, sfTensor, _vp]),
    "sf_pad_shift_stack4_fwd": (C.c_int, [sfTensor, _i64, _i32, _i32, sfTensor, _vp]),
    "sf_pad_shift_stack4_bwd": (C.c_int, [sfTensor, _i64, _i32, _i32, sfTensor, _vp]),
    "sf_regroup5x5_fwd": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "sf_regroup5x5_bwd": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "sf_conv5x5_fwd_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "sf_conv5x5_fwd": (C.c_int, [sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, sfTensor, _vp, _sz, _i32, _vp]),
    "sf_conv5x5_bwd_weight_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "sf_conv5x5_bwd_weight": (C.c_int, [sfTensor, sfTensor, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _sz, _i32, _vp]),
    "sf_space_to_depth2": (C.c_int, [sfTensor, _i64, _i32, _i32, _i32, sfTensor, _vp]),
    "sf_regroup5x5_s2d_fwd": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "sf_regroup5x5_s2d_bwd": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "sf_pad_s2d_fwd": (C.c_int, [sfTensor, _i64, _i32, _i32, sfTensor, _vp]),
    "sf_pad_s2d_bwd": (C.c_int, [sfTensor, _i64, _i32, _i32, sfTensor, _vp]),
    "sf_border": (C.c_int, [sfTensor, _i64, _i32, _i32, _i32, _i32, sfTensor, _vp]),
    "sf_maxpool3d_fwd": (C.c_int, [sfTensor, _i64] + [_i32] * 9 + [sfTensor, _vp]),
    "sf_maxpool3d_bwd": (C.c_int, [sfTensor, sfTensor, _i64] + [_i32] * 9 + [sfTensor, _vp]),
    "sf_film_act_fwd": (C.c_int, [sfTensor, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, sfTensor, _vp]),
    "sf_film_act_bwd_workspace_floats": (_sz, [_i64, _i32, _i32, _i32, _i32]),
    "sf_film_act_bwd": (C.c_int, [sfTensor, sfTensor, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, sfTensor, _vp, _vp, _vp]),
    "sf_relu_sum_pixels_fwd": (C.c_int, [sfTensor, _i64, _i64, _vp, _vp]),
    "sf_relu_sum_pixels_bwd": (C.c_int, [_vp, sfTensor, _i64, _i64, sfTensor, _vp]),
    "sf_axpy": (C.c_int, [_vp, _vp, _vp, C.c_float, _i64, _vp, _vp]),
    "sf_dot": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "sf_tanh": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "sf_dvdgru_gates_fwd": (C.c_int, [sfTensor, sfTensor, sfTensor, _i64, _i32, sfTensor, sfTensor, _vp]),
    "sf_dvdgru_gates_bwd": (C.c_int, [sfTensor, sfTensor, sfTensor, sfTensor, _i64, _i32, sfTensor, sfTensor, _vp]),
    "sf_dvdgru_out_fwd": (C.c_int, [sfTensor, sfTensor, sfTensor, sfTensor, _i64, _i32, sfTensor, sfTensor, _vp]),
    "sf_dvdgru_out_bwd": (C.c_int, [sfTensor, sfTensor, sfTensor, sfTensor, _i64, _i32, sfTensor, sfTensor, sfTensor, _vp]),
    "sf_bmm_f32": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, C.c_float, C.c_float, _vp]),
    "sf_bmm_bf16": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, C.c_float, C.c_float, _vp]),
    "sf_bmm_f16": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, C.c_float, C.c_float, _vp]),
    "sf_flash_attention_fwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, C.c_float, _vp, _i32, _vp, _i32, _vp]),
    "sf_flash_attention_bwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.c_float, _vp, _i32, _vp, _i32,
                                         _vp, _i32, _vp, _i32, _vp]),
    "sf_softmax_rows_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "sf_softmax_rows_bwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "sf_nhwc_to_nchw": (C.c_int, [sfTensor, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _i64, _i32, _vp]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"satflow_amd: {_LIB_PATH} is missing - the HIP library is the product path and has no fallback. "
                "Build it with `python -m satflow_amd.build` (needs hipcc, cross-compiles for gfx950)."
            )
        L = C.CDLL(_LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError here == header/library drift
            fn.restype, fn.argtypes = res, args
        if L.sf_abi_version() != ABI_VERSION:
            raise RuntimeError(f"satflow_amd: ABI mismatch, library {L.sf_abi_version()} != binding {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().sf_last_error_string().decode()}")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """The current HIP stream of the current device as a raw pointer - what every library call is handed.  Round 6: read through torch's C entry
    points (0.3 us) instead of ``torch.cuda.current_stream().cuda_stream``, which builds a Stream object per call: 10 us of host time on EVERY launch
    (tools/probe_axial_host.py's profile: 800 calls, 8 ms) - half of what a small launch costs the host in the host-bound workloads."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        try:
            return _RAW_STREAM(_GET_DEVICE())
        except Exception:  # noqa: BLE001 - an uninitialised context or a changed private API: the public route
            pass
    return torch.cuda.current_stream().cuda_stream


def require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"satflow_amd: `{name}` lives on {t.device}; this path runs only on a HIP device (MI355X) - "
            "there is no CPU implementation behind it (the CPU oracle is test infrastructure)."
        )


NULL = sfTensor(None, 0, 0, 0, 0, 0, None)


# ----------------------------------------------------------------------------------------------
# Device-side error words.  Kernels that can fail at run time (today: the two-workgroup ConvGRU sequence kernels, whose
# boundary-row hand-off has a bounded spin) OR into the STICKY first word of their workspace and poison their outputs with NaN.
# The workspaces are allocated here, zeroed once, cached per (kind, shape, device, stream) and never handed back to the
# allocator, so that the words can be inspected at any later time: ``device_errors()`` / ``check_device_errors()``
# (one synchronisation; FlatAdam.step() calls the latter every ``check_errors_every`` steps).
# ----------------------------------------------------------------------------------------------
_WORKSPACES: dict = {}


def sticky_workspace(kind: str, shape_key, nbytes: int, device) -> Optional[torch.Tensor]:
    """The cached, zero-initialised int64 workspace of ``nbytes`` for kernels of ``kind`` on the current stream (None if 0)."""
    if nbytes <= 0:
        return None
    key = (kind, tuple(shape_key), str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() * 8 < nbytes:
        # never allocate under hipGraph capture: the tensor would live in the graph's private pool and its zeroing would be a memset node
        # replayed with every launch, while the tensor is cached here and reused by eager launches
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"satflow_amd: the {kind} workspace for shape {tuple(shape_key)} must exist before hipGraph capture - "
                               "run the step once eagerly on this stream first")
        new = torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=device)
        if ws is not None:   # a grown workspace keeps the sticky error word of the one it replaces (stream-ordered, no synchronisation)
            new[0].copy_(ws[0])
        _WORKSPACES[key] = ws = new
    return ws


def device_errors() -> dict:
    """{workspace key: error word} for every workspace whose sticky error word is non-zero (synchronises the device)."""
    out = {}
    for key, ws in _WORKSPACES.items():
        v = int(ws[0].item())
        if v:
            out[key] = v
    return out


def check_device_errors() -> None:
    bad = device_errors()
    if bad:
        raise RuntimeError(
            "satflow_amd: a kernel reported a run-time failure (its outputs were set to NaN): "
            + "; ".join(f"{k[0]} {k[1]} on {k[2]}: word {v:#x}" for k, v in bad.items())
            + " - for the ConvGRU sequence kernels this means a boundary-row hand-off between the two workgroups of a map timed "
            "out (SF_GRU_NO_SPLIT=1 selects the one-workgroup kernels)")


def clear_device_errors() -> None:
    for ws in _WORKSPACES.values():
        ws[0].zero_()


def T(t: Optional[torch.Tensor], c: Optional[int] = None, offset: int = 0, idiv: int = 0, imod: int = 0,
      amax: Optional[torch.Tensor] = None) -> sfTensor:
    """Describe a channels-last tensor ``[..., C]`` (or a channel slice ``offset:offset+c`` of it).

    ``idiv`` / ``imod``: image-index remap for convolution inputs (see ``sfTensor`` in the header).
    ``amax``: the device word ``sf_amax`` wrote for this tensor (a gradient operand of the "f32e" kernels; the caller keeps the word alive)."""
    if t is None:
        return sfTensor(None, c or 0, 0, 0, 0, 0, None)
    assert t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16), (t.shape, t.dtype, t.is_contiguous())
    stride = t.shape[-1]
    return sfTensor(t.data_ptr() + t.element_size() * offset, stride - offset if c is None else c, stride, idiv, imod,
                    SF_BF16 if t.dtype == torch.bfloat16 else SF_F32, amax.data_ptr() if amax is not None else None)


# Parameter generation: bumped whenever parameters are rewritten through raw pointers (sf_adam_step),
# which torch's per-tensor version counters cannot see.  Packed-weight caches key on it.
_GENERATION = [0]


def bump_generation() -> None:
    _GENERATION[0] += 1


def generation():
    """Cache key component: parameter generation + the compute dtype the packed images were built for."""
    return (_GENERATION[0], _COMPUTE[0])


# Compute dtype of the convolution kernels: SF_F32 = exact-fp32 MFMA (parity mode, rtol 1e-4);
# SF_BF16 = bf16 operands / fp32 accumulate / fp32 storage (the arithmetic of torch.autocast(bfloat16)
# around the reference's Conv2d; throughput mode).  Weight gradients stay on the fp32 pipe.
# SF_F16 ("f16") = fp16 operands / fp32 accumulate / fp32 storage: the `precision: 16` of the reference's configs/trainer/half.yaml:33 for the
# DGMR-style layers (3x3 convolutions forward / input gradient / weight gradient, split-K launches, the attention products); kernels without an
# fp16 instantiation (recurrent cells, folded BatchNorm, persistent kernels) refuse it.
# "bf16a" = SF_BF16 kernels AND bf16 storage of the MetNet image encoder's activations and their gradients
# (what torch.autocast(bfloat16) leaves in memory between the reference's Conv2d layers); everything from the
# encoder's last pooling on (ConvGRU, attention, head, loss, parameters, optimizer state) stays fp32.
# "f32e" (round 6) = SF_F32E: fp32-EQUIVALENT convolutions on the fp16 matrix pipe - every fp32 operand split into two fp16 parts while it is staged
# (22 mantissa bits), three fp16 products per fp32 product, fp32 accumulation and storage; meets the fp32 parity gate (profiles/r06_f32e_numerics.txt)
# at ~3/16 of the exact-fp32 MFMA cost.  The 3x3 convolutions (forward / input gradient / weight gradient) and the fused recurrent cells take it;
# every other kernel runs its exact-fp32 form in this mode.
_COMPUTE = [SF_F32]
_ENCODER_BF16 = [False]


def set_compute_dtype(name: str) -> None:
    _COMPUTE[0] = {"f32": SF_F32, "fp32": SF_F32, "float32": SF_F32, "bf16": SF_BF16, "bfloat16": SF_BF16, "bf16a": SF_BF16,
                   "f16": SF_F16, "fp16": SF_F16, "float16": SF_F16, "f32e": SF_F32E}[name]
    _ENCODER_BF16[0] = name == "bf16a"


def compute_dtype() -> int:
    return _COMPUTE[0]


def compute_dtype_name() -> str:
    if _COMPUTE[0] == SF_BF16:
        return "bf16a" if _ENCODER_BF16[0] else "bf16"
    return {SF_F16: "f16", SF_F32E: "f32e"}.get(_COMPUTE[0], "f32")


def exact_dtype(dt: Optional[int] = None) -> int:
    """The dtype to hand to an entry point that has no SF_F32E instantiation: its exact-fp32 kernels in "f32e" mode, unchanged otherwise."""
    dt = _COMPUTE[0] if dt is None else dt
    return SF_F32 if dt == SF_F32E else dt


def gate_storage_dtype():
    """torch dtype the ConvLSTM stack keeps its saved gates / gate gradients in (bf16 in "bf16a" mode; they are read by the
    backward pass only: forward results do not depend on it)."""
    return torch.bfloat16 if _ENCODER_BF16[0] else torch.float32


def state_storage_dtype():
    """torch dtype the ConvLSTM stack keeps its HIDDEN states in (bf16 in "bf16a" mode).  A hidden state is only ever read as
    an MFMA operand (next cell's convolution, the output convolution, the weight gradient), which rounds it to bf16 anyway:
    storing the rounded value changes no result of the "bf16" mode, halves its traffic and lets the kernels stage it by
    LDS-DMA.  Cell states stay fp32."""
    return torch.bfloat16 if _ENCODER_BF16[0] else torch.float32


def encoder_storage_dtype():
    """torch dtype the MetNet image encoder keeps its activations in (bf16 only in "bf16a" mode)."""
    return torch.bfloat16 if _ENCODER_BF16[0] else torch.float32


def cpad(c: int) -> int:
    return (c + SF_CPAD - 1) // SF_CPAD * SF_CPAD
