"""ctypes binding of the C-ABI HIP library (``libmis_hip.so``, declared in ``include/mis_hip.h``).

The product path has NO CPU fallback: if the shared library is missing or a
kernel entry point reports an error, a ``RuntimeError`` is raised.  torch is
only used for device memory (``tensor.data_ptr()``) and the current HIP stream.
"""
import ctypes
import os

import torch  # imported first on purpose: libmis_hip.so binds to the libamdhip64 torch has loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIS_HIP_LIB overrides the library file (A/B timing of kernel variants); it is still a HIP build
LIB_PATH = os.environ.get("MIS_HIP_LIB") or os.path.join(_HERE, "libmis_hip.so")

_lib = None

_ERRORS = {
    -1: "MIS_ERR_ARG (bad pointer / size / stride)",
    -2: "MIS_ERR_UNSUPPORTED (shape not covered by the gfx950 kernel family)",
    -3: "MIS_ERR_LAUNCH (HIP launch failure)",
    -4: "MIS_ERR_WORKSPACE (workspace too small)",
}

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_ll = ctypes.c_longlong
c_f = ctypes.c_float
c_d = ctypes.c_double
c_u = ctypes.c_uint
c_ull = ctypes.c_ulonglong

# name -> (restype, argtypes); must stay in sync with include/mis_hip.h
PROTOTYPES = {
    "mis_abi_version": (c_i, []),
    "mis_conv_cin_pad": (c_i, [c_i]),
    "mis_conv_cout_pad": (c_i, [c_i]),
    "mis_conv_packed_floats": (c_ll, [c_i, c_i, c_i, c_i]),
    "mis_conv_pack_weights": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "mis_conv_pack_job_bytes": (c_i, []),
    "mis_conv_pack_job": (c_ll, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_ll]),
    "mis_conv_pack_batch": (c_i, [c_p, c_i, c_ll, c_p]),
    "mis_conv_fwd": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_conv_fwd_stat_tiles": (c_ll, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "mis_conv_fwd_stats": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_ll,
                                 c_ll, c_p]),
    "mis_norm_stats_finalize": (c_i, [c_p, c_i, c_i, c_ll, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_f, c_p]),
    "mis_conv_fwd_kernel_name": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, ctypes.c_char_p, c_i]),
    "mis_conv3d_wino_select": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "mis_conv3d_wino_stat_tiles": (c_ll, [c_i, c_i, c_i, c_i]),
    "mis_conv3d_wino_kernel_name": (c_i, [c_i, ctypes.c_char_p, c_i]),
    "mis_conv3d_wino_fwd": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_ll, c_i, c_p]),
    "mis_conv1x1_gemm_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_ll]),
    "mis_conv1x1_gemm": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_p, c_ll, c_i, c_i, c_i, c_ll, c_i, c_p, c_ll, c_p]),
    "mis_conv1x1_wgrad_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_ll]),
    "mis_conv1x1_wgrad": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, c_ll, c_i, c_p, c_ll, c_p]),
    "mis_conv3d_wino_fwd_splits": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "mis_conv3d_wino_fwd_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "mis_conv3d_wino_fwd_ws": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_ll, c_i, c_p, c_ll,
                                     c_p]),
    "mis_conv3d_wino_dgrad_norm": (c_i, [c_p, c_ll, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_p, c_f, c_p,
                                         c_ll, c_ll, c_i, c_p]),
    "mis_norm_res_act_fwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_ll, c_i, c_p, c_p, c_p, c_p, c_f, c_i, c_p]),
    "mis_norm_res_act_bwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, c_ll, c_i, c_p, c_p,
                                   c_p, c_p, c_f, c_p, c_p, c_i, c_i, c_p, c_ll, c_p]),
    "mis_norm_act_bwd_tiles": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_ll, c_p, c_p, c_f, c_p, c_i, c_p, c_p]),
    "mis_conv2d_wino_select": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "mis_conv2d_wino_stat_tiles": (c_ll, [c_i, c_i, c_i]),
    "mis_conv2d_wino_kernel_name": (c_i, [c_i, ctypes.c_char_p, c_i]),
    "mis_conv2d_wino_fwd": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_ll, c_i, c_p]),
    "mis_conv2d_wino_wgrad_select": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "mis_conv2d_wino_wgrad_kernel_name": (c_i, [c_i, ctypes.c_char_p, c_i]),
    "mis_conv2d_wino_wgrad_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "mis_conv2d_wino_wgrad": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_conv3d_wino_wgrad_select": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "mis_conv3d_wino_wgrad_kernel_name": (c_i, [c_i, ctypes.c_char_p, c_i]),
    "mis_conv3d_wino_wgrad_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "mis_conv3d_wino_wgrad": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_conv_wgrad_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "mis_conv_wgrad_kernel_name": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, ctypes.c_char_p, c_i]),
    "mis_conv_wgrad": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                             c_i, c_p]),
    "mis_norm_workspace_bytes": (c_ll, [c_i, c_i, c_ll, c_i]),
    "mis_norm_stats": (c_i, [c_p, c_ll, c_i, c_i, c_ll, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_ll, c_p]),
    "mis_channel_sum": (c_i, [c_p, c_ll, c_i, c_i, c_ll, c_p, c_i, c_p, c_ll, c_p]),
    "mis_norm_stats_from_running":(c_i, [c_p, c_p, c_f, c_p, c_p, c_i, c_p]),
    "mis_norm_act_fwd": (c_i, [c_p, c_ll, c_p, c_ll, c_i, c_i, c_ll, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_u, c_p,
                               c_p, c_p]),
    "mis_norm_act_bwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_ll, c_i, c_p, c_p, c_p, c_p, c_f, c_f,
                               c_u, c_p, c_p, c_p, c_p, c_i, c_p, c_ll, c_p]),
    "mis_norm_act_fwd_g": (c_i, [c_p, c_ll, c_p, c_ll, c_i, c_i, c_ll, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_u, c_p,
                                 c_p, c_p]),
    "mis_norm_act_bwd_g": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_ll, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_f,
                                 c_f, c_u, c_p, c_p, c_p, c_p, c_i, c_p, c_ll, c_p]),
    "mis_norm_act_fwd_pool": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p,
                                    c_p, c_f, c_f, c_u, c_p, c_p, c_p]),
    "mis_norm_act_bwd_pool": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p,
                                    c_p, c_p, c_p, c_f, c_f, c_u, c_p, c_p, c_p, c_p, c_i, c_p, c_ll, c_p]),
    "mis_norm_act_bwd_sums": (c_i, [c_p, c_ll, c_p, c_ll, c_i, c_i, c_ll, c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_i,
                                    c_p, c_ll, c_p]),
    "mis_conv_wgrad_cin1_norm_eligible": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "mis_conv_wgrad_cin1_norm": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p,
                                       c_f, c_p, c_p, c_ll, c_i, c_p]),
    "mis_norm_head_eligible": (c_i, [c_i, c_i]),
    "mis_norm_head_workspace_bytes": (c_ll, [c_i, c_i, c_ll, c_i, c_i]),
    "mis_norm_head_fwd": (c_i, [c_p, c_ll, c_i, c_i, c_ll, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_u, c_p, c_p, c_p, c_p, c_i,
                                c_p, c_ll, c_p]),
    "mis_norm_head_bwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_ll, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_u,
                                c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_ll, c_p]),
    "mis_maxpool2_fwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_maxpool2_bwd": (c_i, [c_p, c_ll, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_upsample2_fwd": (c_i, [c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_upsample2_bwd": (c_i, [c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_loss_tail_workspace_bytes": (c_ll, [c_i, c_i, c_ll]),
    "mis_loss_tail": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_ll, c_f, c_p, c_f, c_p, c_p, c_ll,
                            c_p, c_ll, c_p]),
    "mis_cross_teaching_tail_workspace_bytes": (c_ll, [c_i, c_i, c_ll]),
    "mis_cross_teaching_tail": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_ll, c_f, c_p, c_p, c_p, c_ll,
                                      c_p, c_ll, c_p]),
    "mis_softmax_mean_accumulate": (c_i, [c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, c_ll, c_f, c_i, c_p]),
    "mis_uamt_tail_workspace_bytes": (c_ll, [c_i, c_i, c_ll]),
    "mis_uamt_tail": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_ll, c_f, c_p, c_ll, c_d, c_f,
                            c_p, c_p, c_ll, c_p, c_ll, c_p]),
    "mis_cross_pseudo_tail": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_ll, c_f, c_p, c_i, c_p, c_p, c_ll,
                                    c_p, c_ll, c_p]),
    "mis_cross_pseudo_mt_tail": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_ll, c_f, c_f, c_p,
                                       c_i, c_p, c_p, c_ll, c_p, c_ll, c_p]),
    "mis_dice_workspace_bytes": (c_ll, [c_i, c_i, c_ll]),
    "mis_dice_loss_fwd": (c_i, [c_p, c_ll, c_p, c_i, c_i, c_i, c_ll, c_p, c_p, c_p, c_ll, c_p]),
    "mis_dice_loss_bwd": (c_i, [c_p, c_ll, c_p, c_i, c_i, c_i, c_ll, c_p, c_p, c_p, c_ll, c_p]),
    "mis_softmax_mse": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_ll, c_i, c_p]),
    "mis_ema_update": (c_i, [c_p, c_p, c_ll, c_f, c_p]),
    "mis_sgd_ema_step": (c_i, [c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_f, c_p, c_p]),
    "mis_teacher_noise": (c_i, [c_p, c_p, c_ll, c_f, c_f, c_u, c_p, c_p]),
    "mis_step_init": (c_i, [c_p, c_ull, c_ll, c_d, c_d, c_d, c_d, c_d, c_ll, c_ll, c_i, c_p]),
    "mis_step_advance": (c_i, [c_p, c_d, c_d, c_d, c_d, c_d, c_ll, c_ll, c_i, c_p]),
    "mis_argmax_channels": (c_i, [c_p, c_ll, c_p, c_i, c_i, c_ll, c_p]),
    "mis_conv_k2s2_eligible": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "mis_conv_k2s2_down": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_conv_k2s2_up": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_conv_k2s2_wgrad_eligible": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "mis_conv_k2s2_wgrad_workspace_bytes": (c_ll, [c_i, c_i]),
    "mis_conv_k2s2_wgrad": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_p]),
    "mis_debug_poison_lds": (c_i, [c_p, c_p]),
    "mis_debug_spin": (c_i, [c_i, c_i, c_i, c_p, c_p]),
    "mis_debug_wgrad_prof": (c_i, [c_p]),
    "mis_space_to_depth2": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_space_to_depth2d": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_add": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_ll, c_p]),
    # token-major (SwinUnet) kernels
    "mis_gemm_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_i]),
    "mis_gemm_dw_workspace_bytes": (c_ll, [c_i, c_i, c_i]),
    "mis_gemm_dw": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_p, c_ll, c_p]),
    "mis_gemm_dw_parts": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_p, c_ll, c_p, c_p]),
    "mis_gemm_set_split_precision": (c_i, [c_i]),
    "mis_gemm_nt_kernel_name": (c_i, [c_i, c_i, c_i, c_i, ctypes.c_char_p, c_i]),
    "mis_gemm_tn_kernel_name": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, ctypes.c_char_p, c_i]),
    "mis_gemm_expand_ln_head": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_f, c_p, c_p,
                                      c_p, c_ll, c_p]),
    "mis_gemm_split_bytes": (c_ll, [c_i, c_i]),
    "mis_gemm_split_b": (c_i, [c_p, c_ll, c_i, c_i, c_p, c_p]),
    "mis_gemm_split_b_layout": (c_i, [c_p, c_ll, c_i, c_i, c_p, c_i, c_p]),
    "mis_gemm_split_job_layout": (c_ll, [c_p, c_p, c_ll, c_i, c_i, c_p, c_ll, c_i]),
    "mis_gemm_nt_split_natural": (c_i, [c_i, c_i, c_i]),
    "mis_gemm_nt_split_layout": (c_i, [c_p, c_ll, c_p, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_p, c_ll, c_p, c_ll,
                                       c_i, c_i, c_i, c_i, c_p, c_ll, c_i, c_p]),
    "mis_gemm_nt_split_layout_kernel_name": (c_i, [c_i, c_i, c_i, c_i, c_i, ctypes.c_char_p, c_i]),
    "mis_gemm_expand_ln_head_split": (c_i, [c_p, c_ll, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_f, c_p, c_p, c_p,
                                            c_ll, c_p]),
    "mis_gemm_nt_residual_ln": (c_i, [c_p, c_ll, c_p, c_p, c_i, c_i, c_i, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_p, c_f, c_p, c_ll,
                                      c_p, c_p, c_p]),
    "mis_gemm_split_job_bytes": (c_ll, []),
    "mis_gemm_split_job": (c_ll, [c_p, c_p, c_ll, c_i, c_i, c_p, c_ll]),
    "mis_gemm_split_batch": (c_i, [c_p, c_i, c_ll, c_p]),
    "mis_gemm_nt_split_workspace_bytes": (c_ll, [c_i, c_i, c_i]),
    "mis_gemm_nt_split": (c_i, [c_p, c_ll, c_p, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_p, c_ll, c_p, c_ll,
                                c_i, c_i, c_i, c_i, c_p, c_ll, c_p]),
    "mis_gemm_nt_split_kernel_name": (c_i, [c_i, c_i, c_i, c_i, ctypes.c_char_p, c_i]),
    "mis_gemm_expand": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_gemm": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_ll, c_p]),
    "mis_gemm_ex": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p,
                          c_ll, c_p]),
    "mis_droppath_table": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    "mis_transpose": (c_i, [c_p, c_ll, c_p, c_ll, c_i, c_i, c_p]),
    "mis_win3d_gather": (c_i, [c_p, c_p] + [c_i] * 12 + [c_p]),
    "mis_win3d_windows": (c_ll, [c_i] * 7),
    "mis_merge3d": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_win3d_attn_fwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "mis_win3d_attn_workspace_bytes": (c_ll, [c_i, c_i, c_i]),
    "mis_win3d_attn_bwd": (c_i, [c_p, c_ll, c_p, c_p, c_ll, c_p, c_ll, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p,
                                 c_ll, c_p]),
    "mis_transpose_job_bytes": (c_ll, []),
    "mis_transpose_job": (c_ll, [c_p, c_p, c_p, c_i, c_i, c_ll]),
    "mis_transpose_batch": (c_i, [c_p, c_i, c_ll, c_p]),
    "mis_layernorm_fwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_p, c_p, c_p, c_ll, c_i, c_f, c_p]),
    "mis_colreduce_workspace_bytes": (c_ll, [c_ll, c_i]),
    "mis_layernorm_bwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_p,
                                c_ll, c_p]),
    "mis_layernorm_bwd_parts": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_ll, c_p]),
    "mis_layernorm_bwd_final": (c_i, [c_p, c_ll, c_ll, c_i, c_p, c_p, c_i, c_p]),
    "mis_colsum_job_bytes": (c_ll, []),
    "mis_colreduce_slabs": (c_ll, [c_ll]),
    "mis_colsum_job": (c_ll, [c_p, c_p, c_ll, c_ll, c_i, c_i, c_p, c_p, c_i, c_ll]),
    "mis_colsum_batch": (c_i, [c_p, c_i, c_ll, c_p]),
    "mis_layernorm_bwd_residual_parts": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_i, c_p, c_ll, c_p, c_ll, c_p, c_p,
                                               c_p, c_ll, c_i, c_p, c_ll, c_p]),
    "mis_colsum": (c_i, [c_p, c_ll, c_ll, c_i, c_p, c_i, c_p, c_ll, c_p]),
    "mis_gelu": (c_i, [c_p, c_p, c_p, c_ll, c_i, c_p]),
    "mis_residual_droppath": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_ll, c_ll, c_i, c_ll, c_f, c_u, c_p, c_p,
                                    c_i, c_p]),
    "mis_token_rearrange": (c_i, [c_p, c_ll, c_p, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_patch_im2col": (c_i, [c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_p]),
    "mis_ln_head_fwd": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_ll, c_i, c_i, c_f, c_p]),
    "mis_ln_head_workspace_bytes": (c_ll, [c_ll, c_i, c_i]),
    "mis_ln_head_bwd": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_p, c_ll, c_i, c_p, c_p, c_p, c_i, c_i, c_ll,
                              c_i, c_i, c_p, c_ll, c_p]),
    "mis_ln_head_bwd_unshuffle": (c_i, [c_p, c_ll, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_p, c_ll, c_i, c_p, c_p, c_p, c_i, c_i, c_i,
                                        c_i, c_i, c_i, c_i, c_p, c_ll, c_p]),
    "mis_head_fwd": (c_i, [c_p, c_ll, c_p, c_p, c_ll, c_i, c_ll, c_i, c_i, c_p]),
    "mis_head_workspace_bytes": (c_ll, [c_i, c_i]),
    "mis_head_bwd": (c_i, [c_p, c_ll, c_p, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_ll, c_i, c_i, c_p, c_ll, c_p]),
    "mis_window_attention_fwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    "mis_window_attention_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_i]),
    "mis_window_attention_bwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f,
                                       c_p, c_ll, c_p]),
    "mis_patch_im2col_c": (c_i, [c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_window_attention_fwd_ws": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "mis_window_attention_workspace_bytes_ws": (c_ll, [c_i, c_i, c_i, c_i, c_i]),
    "mis_window_attention_bwd_ws": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f,
                                          c_i, c_p, c_ll, c_p]),
    "mis_window_attention_bwd_parts_ws": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p,
                                                c_ll, c_p]),
    "mis_window_attention_dtable_ws": (c_i, [c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_window_attention_table_partials": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "mis_full_attention_fwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_f, c_p]),
    "mis_full_attention_workspace_bytes": (c_ll, [c_i, c_i, c_i]),
    "mis_full_attention_bwd": (c_i, [c_p, c_ll, c_p, c_ll, c_p, c_ll, c_p, c_i, c_i, c_i, c_f, c_p, c_ll, c_p]),
    "mis_patch3d_im2col": (c_i, [c_p, c_ll, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "mis_add_rowcycle": (c_i, [c_p, c_ll, c_p, c_p, c_ll, c_ll, c_i, c_i, c_p]),
    "mis_sum_rowcycle": (c_i, [c_p, c_ll, c_p, c_ll, c_i, c_i, c_p]),
    "mis_augment2d": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "mis_crop_rotflip3d": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p]),
}

STEP_STATE_BYTES = 40  # sizeof(MisStepState)
AUG2D_BYTES = 88       # sizeof(MisAug2D)
CROP3D_BYTES = 48      # sizeof(MisCrop3D)


class StreamPtr(ctypes.c_void_p):
    """The hipStream_t argument of a launching entry point (``stream_ptr()``): what marks a call as a launch for a LaunchTape."""


class LaunchTape:
    """One training step as a flat list of ``(callable, args)``: every C-ABI launch the eager step issued (the ctypes function and
    its already converted arguments: raw pointers into the plans' static buffers), the stream dependencies between them (HIP
    event record / wait pairs) and the Python callbacks of the gradient exchange (bucket all-reduces, on their streams).

    Why: the step is static -- same buffers, same launch sequence, the per-step scalars (RNG offset, learning rate, EMA alpha,
    consistency weight) live in a device-resident MisStepState -- but enqueueing it through the op graph costs the host 8 .. 10 ms
    of a 20 ms SwinUnet step (623 launches x ~14 us of Python per launch), and a captured hipGraph is no way out on this stack:
    replaying the SwinUnet step's graph costs the host 10.9 ms (measured, profiles/r06_*: ROCm enqueues the nodes one by one).
    Replaying the tape is one ctypes call per launch (~2.5 us).  ``recording()`` is the context the trainers run ONE eager step
    in; afterwards ``replay()`` is the step.  Everything the recorded step touched must stay where it is: inputs are copied
    into the static tensors the recording saw, grow-only scratch buffers are never freed (ops.scratch)."""

    def __init__(self):
        self.items = []
        self.keep = []          # objects the recorded arguments point into (events, tensors)

    def recording(self):
        return _Recording(self)

    def replay(self):
        for fn, args in self.items:
            st = fn(*args)
            if st:
                raise RuntimeError(f"launch tape: {getattr(fn, '__name__', fn)} failed with status {st}")

    def __len__(self):
        return len(self.items)


class _Recording:
    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        global TAPE
        if TAPE is not None:
            raise RuntimeError("a launch tape is already being recorded")
        TAPE = self.tape
        return self.tape

    def __exit__(self, *exc):
        global TAPE
        TAPE = None
        return False


TAPE = None        # the LaunchTape being recorded (None: eager)


class _RecordingLib:
    """Stands in for the ctypes library while a tape is recorded: calls go through, launches (last argument a StreamPtr) are
    appended to the tape with their argument tuples."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*args):
            st = fn(*args)
            # a refused call (MIS_ERR_UNSUPPORTED: the caller goes on to another entry point) launched nothing
            if st == 0 and TAPE is not None and args and isinstance(args[-1], StreamPtr):
                TAPE.items.append((fn, args))
            return st
        call.__name__ = name
        setattr(self, name, call)
        return call


_rec_lib = None


def load():
    """Load libmis_hip.so (once) and attach prototypes.  Raises if it is not built."""
    global _lib, _rec_lib
    if _lib is not None:
        if TAPE is not None:
            if _rec_lib is None:
                _rec_lib = _RecordingLib(_lib)
            return _rec_lib
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C cv-ssl-mis_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed: {_ERRORS.get(int(status), status)}")


def stream_ptr():
    """hipStream_t of torch's current stream, as a void*."""
    return StreamPtr(torch.cuda.current_stream().cuda_stream)


_hip = None


def _hip_runtime():
    """The libamdhip64 torch has loaded (for event record / wait pairs on a tape)."""
    global _hip
    if _hip is None:
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64.so" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError("libamdhip64.so is not loaded in this process")
        h = ctypes.CDLL(path)
        h.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        h.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        h.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        for fn in (h.hipEventCreateWithFlags, h.hipEventRecord, h.hipStreamWaitEvent):
            fn.restype = ctypes.c_int
        _hip = h
    return _hip


def wait_stream(waiter, waited):
    """``waiter.wait_stream(waited)``; while a tape is recorded the dependency is an event record / wait pair of the tape's own
    (replayed as two HIP runtime calls)."""
    if TAPE is None:
        waiter.wait_stream(waited)
        return
    h = _hip_runtime()
    ev = ctypes.c_void_p()
    if h.hipEventCreateWithFlags(ctypes.byref(ev), 0x2) != 0:          # hipEventDisableTiming
        raise RuntimeError("hipEventCreateWithFlags failed")
    rec = (h.hipEventRecord, (ev, ctypes.c_void_p(waited.cuda_stream)))
    wai = (h.hipStreamWaitEvent, (ctypes.c_void_p(waiter.cuda_stream), ev, 0))
    for fn, args in (rec, wai):
        if fn(*args) != 0:
            raise RuntimeError("HIP event record / wait failed")
        TAPE.items.append((fn, args))
    TAPE.keep.append(ev)


def tape_call(fn, *args):
    """``fn(*args)`` now, and again -- with the stream that is current now -- at this point of every replay when a tape is being
    recorded (the Python side of the gradient exchange: torch.distributed collectives order themselves behind the current
    stream).  Returns fn's result."""
    r = fn(*args)
    if TAPE is not None:
        stream = torch.cuda.current_stream()

        def again():
            with torch.cuda.stream(stream):
                fn(*args)
        again.__name__ = getattr(fn, "__name__", "callback")
        TAPE.items.append((again, ()))
    return r


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mis_hip kernels need device tensors (HIP/gfx950); got a CPU tensor. "
                               "There is no CPU fallback in the product path.")


def side_stream(kind):
    """A side stream of the step (``kind``: "wgrad" = the weight-gradient stream of a plan, "side" = the teacher / second network of a
    trainer).  MIS_WGRAD_PRIORITY / MIS_SIDE_PRIORITY give it a HIP stream priority (torch's convention: lower = more urgent, the
    range is ``torch.cuda.Stream.priority_range()``); unset: the default priority, like the main stream."""
    p = os.environ.get({"wgrad": "MIS_WGRAD_PRIORITY", "side": "MIS_SIDE_PRIORITY"}[kind])
    if p is None or p == "":
        return torch.cuda.Stream()
    return torch.cuda.Stream(priority=int(p))
