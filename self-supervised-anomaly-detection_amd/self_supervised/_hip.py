"""ctypes binding of libssad_hip.so (C ABI declared in include/ssad.h).

The product path has no CPU fallback: if the shared object is missing or a tensor is not a
contiguous fp32 ROCm tensor, these wrappers raise.  PyTorch is used only for device memory and
streams (``data_ptr()``, ``torch.cuda.current_stream()``).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libssad_hip.so")

_c_fp = ctypes.c_void_p
_c_i = ctypes.c_int
_c_l = ctypes.c_int64
_c_f = ctypes.c_float

# name -> argtypes; mirrors include/ssad.h one to one (tests/test_host_cpu.py::test_library_builds_and_exports_every_declared_symbol checks the header too)
SIGNATURES = {
    "ssad_repack_oihw_to_ohwi": [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_repack_ohwi_to_oihw": [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_pack_stem_weight": [_c_fp, _c_fp, _c_fp],
    "ssad_pack_stem_weight_ohwi": [_c_fp, _c_fp, _c_fp],
    "ssad_pack_stem_weight16_ohwi": [_c_fp, _c_fp, _c_i, _c_fp],
    "ssad_stem_fwd": [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_fp, _c_fp],
    "ssad_stem_stats_rows": [],
    "ssad_stem_fwd_stats": [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_f, _c_f, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_pack_stem_weight_folded": [_c_fp, _c_fp, _c_fp],
    "ssad_stem_patch_pool_fwd": [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_i, _c_fp, _c_fp],
    "ssad_stem_patch_pool_fwd_ring": [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp],
    "ssad_stem_patch_border_fwd": [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_maxpool3x3s2_fwd": [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_conv_igemm_fwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                            _c_i, _c_i, _c_fp],
    "ssad_conv_igemm_fwd_hwnc": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                 _c_i, _c_i, _c_fp],
    "ssad_conv_igemm_fwd_hwnc_ring": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                      _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_patch_gather_hwnc": [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_patch_gather_hwnc_band": [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_conv_igemm_tile": [_c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i],
    "ssad_gap_fwd": [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_l2_normalize_rows": [_c_fp, _c_fp, _c_l, _c_i, _c_fp],
    "ssad_cosine_knn_mean": [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp],
    "ssad_cosine_knn_fused": [_c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_fp],
    "ssad_blur_relu_bilinear": [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_flip_transpose_weight": [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_flip_transpose_batch": [_c_fp, _c_fp, ctypes.POINTER(ctypes.c_int64), _c_i, _c_fp],
    "ssad_conv_igemm_dgrad": [_c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                              _c_fp],
    "ssad_bn_small_ok": [_c_l, _c_i],
    "ssad_bn_small_fwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_f, _c_f, _c_i, _c_fp],
    "ssad_bn_small_bwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp],
    "ssad_linear_small_max_rows": [],
    "ssad_linear_wgrad_small": [_c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_fp],
    "ssad_linear_wgrad_small_r": [_c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_wgrad_splits": [_c_l, _c_i, _c_i, _c_i, _c_i],
    "ssad_wgrad_splits_bf16": [_c_l, _c_i, _c_i, _c_i, _c_i],
    "ssad_conv_wgrad": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_conv_igemm_fwd_bf16": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                 _c_i, _c_i, _c_fp],
    "ssad_conv_igemm_fwd_x3": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                               _c_i, _c_i, _c_i, _c_fp],
    "ssad_conv_igemm_fwd_x6": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                               _c_i, _c_i, _c_i, _c_fp],
    "ssad_conv_igemm_dgrad_x3": [_c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                 _c_i, _c_fp],
    "ssad_conv_igemm_dgrad_x6": [_c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                 _c_i, _c_fp],
    "ssad_conv_igemm_dgrad_bf16": [_c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                   _c_i, _c_fp],
    "ssad_conv_wgrad_bf16": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_conv3x3_c64_stats_rows": [_c_l, _c_i, _c_i],
    "ssad_conv3x3_c64": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp, _c_f, _c_f,
                         _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_conv3x3_c64_op": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp, _c_f, _c_f,
                         _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_fp],
    "ssad_conv3x3_c64_eval": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_bn_apply_fwd_mask": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp],
    "ssad_bn_bwd_reduce_mask": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_fp],
    "ssad_bn_apply_bwd_mask": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp],
    "ssad_conv_igemm_dgrad_masked": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                     _c_i, _c_fp],
    "ssad_wgrad3x3_halo_ok": [_c_i, _c_i, _c_i, _c_i, _c_i, _c_i],
    "ssad_wgrad3x3_halo_splits": [_c_l, _c_i, _c_i, _c_i, _c_i],
    "ssad_conv_wgrad3x3_halo": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_pack_stem_weight16": [_c_fp, _c_fp, _c_i, _c_fp],
    "ssad_stem_fwd_stats16": [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_f, _c_f, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_fp],
    "ssad_wgrad3x3_halo16_ok": [_c_i, _c_i, _c_i, _c_i, _c_i, _c_i],
    "ssad_wgrad3x3_halo16_splits": [_c_l, _c_i, _c_i, _c_i, _c_i],
    "ssad_conv_wgrad3x3_halo16": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_conv_wgrad3x3s2_halo": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_conv_igemm_fwd_f16": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                _c_i, _c_i, _c_fp],
    "ssad_conv_igemm_dgrad_f16": [_c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                  _c_i, _c_fp],
    "ssad_conv_wgrad_f16": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_sgd_step_dev": [_c_fp, _c_fp, _c_fp, _c_l, _c_fp, _c_fp, _c_fp],
    "ssad_scale_by_loss_scale": [_c_fp, _c_l, _c_fp, _c_fp],
    "ssad_check_finite": [_c_fp, _c_l, _c_fp, _c_fp],
    "ssad_loss_scaler_update": [_c_fp, _c_f, _c_f, _c_i, _c_fp],
    "ssad_conv_wgrad_x3": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_conv_wgrad_x6": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_wgrad_reduce_batch": [ctypes.POINTER(ctypes.c_int64), _c_i, _c_fp],
    "ssad_wgrad_reduce": [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_stem_im2col": [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_pack_stem_weight_2d": [_c_fp, _c_fp, _c_fp],
    "ssad_colreduce_workspace": [_c_l, _c_i],
    "ssad_bn_stats": [_c_fp, _c_l, _c_i, _c_f, _c_f, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_bn_apply_fwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp],
    "ssad_bn_bwd_reduce": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_fp],
    "ssad_bn_relu_maxpool_fwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_fp],
    "ssad_pool_bn_relu_bwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i,
                              _c_l, _c_fp, _c_fp],
    "ssad_gradcam_map": [_c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_fp],
    "ssad_stem_wgrad_workspace": [_c_i, _c_i, _c_i],
    "ssad_stem_wgrad": [_c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_l, _c_i, _c_i, _c_fp, _c_fp],
    "ssad_conv_stats_workspace": [_c_l, _c_i, _c_i, _c_i],
    "ssad_conv_igemm_fwd_stats": [_c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                  _c_f, _c_f, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_bn_bwd_reduce_zmask": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_fp],
    "ssad_bn_apply_bwd_zmask": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp],
    "ssad_bn_apply_bwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp],
    "ssad_maxpool3x3s2_bwd": [_c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_maxpool3x3s2_fwd_idx": [_c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_fp],
    "ssad_maxpool3x3s2_bwd_idx": [_c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_gap_bwd": [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_softmax_ce": [_c_fp, _c_fp, _c_i, _c_i, _c_fp, _c_fp, _c_i, _c_f, _c_fp],
    "ssad_sgd_step": [_c_fp, _c_fp, _c_fp, _c_l, _c_f, _c_f, _c_f, _c_f, _c_fp],
    "ssad_auroc_workspace": [_c_l],
    "ssad_auroc": [_c_fp, _c_fp, _c_l, _c_fp, _c_l, _c_fp, _c_fp],
    "ssad_pro_curve_workspace": [_c_l],
    "ssad_pro_curve": [_c_fp, _c_fp, _c_fp, _c_l, ctypes.c_double, ctypes.c_double, _c_fp, _c_l, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_best_f1_workspace": [_c_l],
    "ssad_best_f1_threshold": [_c_fp, _c_fp, _c_l, _c_fp, _c_l, _c_fp, _c_fp],
    "ssad_confusion_counts": [_c_fp, _c_fp, _c_l, _c_f, _c_fp, _c_fp],
    "ssad_aug_params_size": [],
    "ssad_cutpaste_augment": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i,
                              ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _c_fp],
    "ssad_affine_window_sum_u8": [_c_fp, _c_i, _c_i, ctypes.POINTER(ctypes.c_int32), _c_i, _c_i, _c_i, _c_i,
                                  ctypes.POINTER(ctypes.c_int64)],
    "ssad_u8hwc_to_f32chw": [_c_fp, _c_fp, _c_i, _c_i, _c_i, _c_fp],
    "ssad_obj_mask_workspace": [_c_i, _c_i, _c_i],
    "ssad_obj_mask": [_c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, ctypes.POINTER(ctypes.c_double), _c_i, ctypes.c_double, ctypes.c_double,
                      _c_fp, _c_fp],
    "ssad_resize_bicubic_u8": [_c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_i, _c_fp, _c_fp, _c_i, _c_fp],
    "ssad_u8hwc_to_f32chw_norm": [_c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                                  _c_fp],
    # half-tensor forms of the precision-16 training step (include/ssad.h, last section)
    "ssad_cvt_f32_f16": [_c_fp, _c_fp, _c_l, _c_fp],
    "ssad_flip_transpose_batch_h": [_c_fp, _c_fp, ctypes.POINTER(ctypes.c_int64), _c_i, _c_fp],
    "ssad_stem_fwd_stats16_h": [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp, _c_f, _c_f, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_bn_relu_maxpool_fwd_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_fp],
    "ssad_conv_igemm_fwd_stats_h": [_c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                    _c_f, _c_f, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_conv_igemm_dgrad_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_conv3x3_c64_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp, _c_f, _c_f,
                           _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_conv3x3_h_ok": [_c_i, _c_i],
    "ssad_conv3x3_h_stats_rows": [_c_l, _c_i, _c_i, _c_i],
    "ssad_conv3x3_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_f, _c_f,
                       _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_conv3x3_hw_ok": [_c_l, _c_i, _c_i, _c_i, _c_i],
    "ssad_conv3x3_hw_packed_size": [_c_i, _c_i],
    "ssad_conv3x3_hw_stats_rows": [_c_l, _c_i, _c_i, _c_i],
    "ssad_conv3x3_hw_pack_batch": [_c_fp, _c_fp, _c_fp, _c_i, _c_fp],
    "ssad_conv3x3_hw": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_f, _c_f,
                        _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_bn_apply_fwd_mask_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp],
    "ssad_bn_bwd_reduce_mask_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_fp],
    "ssad_bn_apply_bwd_mask_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp],
    "ssad_conv3x3_fw_ok": [_c_l, _c_i, _c_i, _c_i, _c_i],
    "ssad_conv3x3_fw_pack_batch": [_c_fp, _c_fp, _c_fp, _c_i, _c_fp],
    "ssad_conv3x3_fw": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_f,
                        _c_f, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_bn_relu_maxpool_fwd_win": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_fp],
    "ssad_bn_relu_maxpool_fwd_win_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_fp],
    "ssad_pool_bn_relu_bwd_apply": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_pool_bn_relu_bwd_apply_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_conv3x3_fw_eval_ok": [_c_l, _c_i, _c_i, _c_i, _c_i],
    "ssad_conv3x3_fw_pack_scaled": [_c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_fp],
    "ssad_conv3x3_fw_eval": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_bn_stats_h": [_c_fp, _c_l, _c_i, _c_f, _c_f, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "ssad_bn_apply_fwd_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp],
    "ssad_gap_fwd_h": [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_gap_bwd_h": [_c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_fp],
    "ssad_bn_bwd_reduce_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_fp],
    "ssad_bn_bwd_reduce_zmask_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp, _c_fp],
    "ssad_bn_apply_bwd_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_fp],
    "ssad_bn_apply_bwd_zmask_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_fp],
    "ssad_pool_bn_relu_bwd_h": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i,
                                _c_l, _c_fp, _c_fp],
    "ssad_wgrad3x3_g16_ok": [_c_i, _c_i, _c_i, _c_i, _c_i, _c_i],
    "ssad_wgrad3x3_g16_splits": [_c_l, _c_i, _c_i, _c_i, _c_i, _c_i],
    "ssad_conv_wgrad3x3_g16_h": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_conv_wgrad3x3_halo16_h": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_conv_wgrad_f16_h": [_c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_l, _c_fp],
    "ssad_stem_wgrad_h": [_c_fp, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_l, _c_i, _c_i, _c_fp, _c_fp],
}
RESTYPES = {"ssad_conv3x3_c64_stats_rows": _c_l, "ssad_conv3x3_h_stats_rows": _c_l, "ssad_conv3x3_hw_stats_rows": _c_l, "ssad_conv3x3_hw_packed_size": _c_l, "ssad_colreduce_workspace": _c_l, "ssad_conv_stats_workspace": _c_l, "ssad_stem_wgrad_workspace": _c_l, "ssad_auroc_workspace": _c_l, "ssad_obj_mask_workspace": _c_l, "ssad_pro_curve_workspace": _c_l,
            "ssad_best_f1_workspace": _c_l}

_lib = None


class HipExtensionError(RuntimeError):
    pass


def lib():
    """Loads the shared object once; raises HipExtensionError when it is absent (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
        l = ctypes.CDLL(LIB_PATH)
        l.ssad_version.restype = _c_i
        l.ssad_last_error.restype = ctypes.c_char_p
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = RESTYPES.get(name, _c_i)
        _lib = l
    return _lib


LAUNCHES = 0          # C-ABI calls made so far (graph capture uses it to recognise segments that recorded nothing)


def check(rc):
    global LAUNCHES
    LAUNCHES += 1
    if rc != 0:
        raise HipExtensionError(lib().ssad_last_error().decode())


def ptr(t, allow_none=False, dtype=torch.float32):
    if t is None:
        if allow_none:
            return None
        raise HipExtensionError("null tensor")
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise HipExtensionError(
            f"expected a contiguous {dtype} ROCm tensor, got device={t.device} dtype={t.dtype} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def hptr(t, allow_none=False):
    """Device pointer of a contiguous half tensor (the activations of the precision-16 step)."""
    return ptr(t, allow_none, torch.float16)


def stream():
    return torch.cuda.current_stream().cuda_stream
