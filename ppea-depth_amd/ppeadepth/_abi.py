"""ctypes binding of libppea_depth.so (C ABI: include/ppea_depth.h).

This is the only place the product talks to native code.  There is NO fallback:
if the shared library is missing or an entry point is absent, importing this module
raises, and every op refuses tensors that are not resident on a HIP device.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libppea_depth.so")

ABI_VERSION = 12

_vp, _i, _l, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float

# name -> argtypes (restype is always int = hipError_t); mirrors include/ppea_depth.h
SIGNATURES = {
    "ppea_abi_version": [],
    "ppea_timestamp": [_vp, _vp],
    "ppea_dwconv_lk_fwd_f32": [_vp] * 5 + [_i] * 6 + [_vp],
    "ppea_dwconv_lk_fwd_bf16": [_vp] * 5 + [_i] * 6 + [_vp],
    "ppea_dwconv_lk_bwd_data_f32": [_vp] * 5 + [_i] * 6 + [_vp],
    "ppea_dwconv_lk_bwd_data_bf16": [_vp] * 5 + [_i] * 6 + [_vp],
    "ppea_dwconv_lk_bwd_filter_f32": [_vp] * 3 + [_i] * 5 + [_vp],
    "ppea_dwconv3x3_fwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_dwconv3x3_fwd_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_dwconv3x3_bwd_data_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_dwconv3x3_bwd_data_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_dwconv_lk_packed_bytes": [_i, _i],
    "ppea_dwconv_lk_pack_bf16": [_vp, _vp, _i, _i, _i, _vp],
    "ppea_dwconv_lk_fwd_bf16p": [_vp] * 5 + [_i] * 6 + [_vp],
    "ppea_dwconv_lk_bwd_data_bf16p": [_vp] * 5 + [_i] * 6 + [_vp],
    "ppea_dwconv_lk_fwd_bn_bf16p": [_vp] * 6 + [_i, _l, _vp, _vp, _f, _f] + [_vp] * 4 + [_i] * 6 + [_vp],
    "ppea_pwconv_bf16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_pwconv_ex_bf16": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_pwgrad_workspace_bytes": [_i, _i, _i, _i],
    "ppea_pwgrad_bf16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_pwgrad_ex_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "ppea_pwgrad_pair_workspace_bytes": [_i, _vp, _vp, _i],
    "ppea_pwgrad_ex_pair_bf16": [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ppea_nhwc_bn_slabs": [_i, _i],
    "ppea_nhwc_bn_stats_f32": [_vp, _vp, _i, _i, _i, _vp],
    "ppea_nhwc_bn_stats_bf16": [_vp, _vp, _i, _i, _i, _vp],
    "ppea_nhwc_bn_finalize_f32": [_vp, _i, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "ppea_nhwc_bn_apply_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_bn_apply_bf16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_bn_bwd_reduce_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_bn_bwd_reduce_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_bn_bwd_finalize_f32": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "ppea_nhwc_bn_bwd_apply_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_bn_bwd_apply_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_bias_elu_chunks": [_i, _i, _i],
    "ppea_bias_elu_fwd_f32": [_vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "ppea_bias_elu_fwd_bf16": [_vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "ppea_bias_elu_bwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ppea_bias_elu_bwd_bf16": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ppea_nhwc_reflect_pad1_fwd_f32": [_vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_reflect_pad1_fwd_bf16": [_vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_reflect_pad1_bwd_f32": [_vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_reflect_pad1_bwd_bf16": [_vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_bias_elu_slabs": [_i, _i],
    "ppea_nhwc_bias_elu_fwd_f32": [_vp, _vp, _i, _vp, _i, _i, _vp],
    "ppea_nhwc_bias_elu_fwd_bf16": [_vp, _vp, _i, _vp, _i, _i, _vp],
    "ppea_nhwc_bias_elu_bwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "ppea_nhwc_bias_elu_bwd_bf16": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "ppea_adam_flat_f32": [_vp, _vp, _vp, _vp, _vp, ctypes.c_long, ctypes.c_long, _vp, _f, _f, _f, _vp],
    "ppea_adam_flat_scaled_f32": [_vp, _vp, _vp, _vp, _vp, ctypes.c_long, ctypes.c_long, _vp, _f, _f, _f, _f, _vp],
    "ppea_tapsum_fwd_bf16": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_tapsum_bwd_bf16": [_vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_stats_f32": [_vp, _vp, _i, _i, _i, _vp],
    "ppea_bn_stats_bf16": [_vp, _vp, _i, _i, _i, _vp],
    "ppea_bn_finalize_f32": [_vp, _i, _i, _i, _f, _f] + [_vp] * 5 + [_vp],
    "ppea_bn_stats_final_f32": [_vp, _i, _i, _i, _f, _f] + [_vp] * 5 + [_vp],
    "ppea_bn_stats_final_bf16": [_vp, _i, _i, _i, _f, _f] + [_vp] * 5 + [_vp],
    "ppea_bn_bwd_reduce_final_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_bwd_reduce_final_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_stats_packed_f32": [_vp, _i, _i, _i, _vp, _vp],
    "ppea_bn_stats_packed_bf16": [_vp, _i, _i, _i, _vp, _vp],
    "ppea_bn_finalize_packed_f32": [_vp, _i, _i, _i, _vp, _vp],
    "ppea_bn_sync_combine_f32": [_vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "ppea_bn_apply_f32": [_vp] * 6 + [_f, _vp] + [_i] * 4 + [_vp],
    "ppea_bn_apply_bf16": [_vp] * 6 + [_f, _vp] + [_i] * 4 + [_vp],
    "ppea_bn_bwd_reduce_f32": [_vp] * 6 + [_i] * 4 + [_vp],
    "ppea_bn_bwd_reduce_bf16": [_vp] * 6 + [_i] * 4 + [_vp],
    "ppea_bn_bwd_finalize_f32": [_vp, _i, _i, _vp, _vp],
    "ppea_bn_bwd_apply_f32": [_vp] * 6 + [_f, _vp, _vp] + [_i] * 4 + [_vp],
    "ppea_bn_bwd_apply_bf16": [_vp] * 6 + [_f, _vp, _vp] + [_i] * 4 + [_vp],
    "ppea_bn_sync_bwd_apply_f32": [_vp] * 6 + [_f, _vp, _vp, _vp, _vp, _f] + [_i] * 4 + [_vp],
    "ppea_bn_sync_bwd_apply_bf16": [_vp] * 6 + [_f, _vp, _vp, _vp, _vp, _f] + [_i] * 4 + [_vp],
    "ppea_bn_sync_stats_workspace_bytes": [_i] * 4,
    "ppea_bn_sync_stats_f32": [_vp] * 4 + [_i] * 3 + [_vp],
    "ppea_bn_sync_stats_bf16": [_vp] * 4 + [_i] * 3 + [_vp],
    "ppea_bn_sync_stats_from_sums_f32": [_vp, _i, _i, _l, _vp, _vp],
    "ppea_bn_sync_apply_f32": [_vp, _vp, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _f, _vp, _vp] + [_i] * 4 + [_vp],
    "ppea_bn_sync_apply_bf16": [_vp, _vp, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _f, _vp, _vp] + [_i] * 4 + [_vp],
    "ppea_reflect_pad1_fwd_f32": [_vp, _vp, _l, _i, _i, _vp],
    "ppea_reflect_pad1_fwd_bf16": [_vp, _vp, _l, _i, _i, _vp],
    "ppea_reflect_pad1_bwd_f32": [_vp, _vp, _l, _i, _i, _vp],
    "ppea_reflect_pad1_bwd_bf16": [_vp, _vp, _l, _i, _i, _vp],
    "ppea_backproject_project_fwd_f32": [_vp] * 4 + [_i] * 3 + [_f, _vp],
    "ppea_backproject_project_bwd_workspace_bytes": [_i] * 3,
    "ppea_backproject_project_bwd_f32": [_vp] * 7 + [_i] * 3 + [_f, _vp],
    "ppea_grid_sample_fwd_f32": [_vp] * 3 + [_i] * 7 + [_vp],
    "ppea_grid_sample_bwd_grid_f32": [_vp] * 4 + [_i] * 7 + [_vp],
    "ppea_pose_matrix_fwd_f32": [_vp, _vp, _vp, _i, _i, _vp],
    "ppea_pose_matrix_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "ppea_ssim_l1_fwd_f32": [_vp, _vp, _vp, _l] + [_i] * 4 + [_f, _vp],
    "ppea_ssim_l1_bwd_f32": [_vp, _vp, _vp, _l, _vp] + [_i] * 4 + [_f, _vp],
    "ppea_smooth_num_partials": [],
    "ppea_smooth_fwd_f32": [_vp] * 3 + [_i] * 4 + [_vp],
    "ppea_smooth_bwd_f32": [_vp, _vp, _f, _f, _vp] + [_i] * 4 + [_vp],
    "ppea_loss_select_f32": [_vp] * 9 + [_i] * 5 + [_vp],
    "ppea_bn_fwd_channel_f32": [_vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_fwd_channel_bf16": [_vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_fwd_channel_sums_f32": [_vp, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_fwd_channel_sums_bf16": [_vp, _vp, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_bwd_channel_f32": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_bwd_channel_bf16": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_fwd_channel_next_f32": [_vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp],
    "ppea_bn_fwd_channel_next_bf16": [_vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp],
    "ppea_bn_bwd_channel_next_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ppea_bn_bwd_channel_next_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ppea_bn_bwd_channel_dup_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_bwd_channel_dup_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_bwd_channel_next_dup_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ppea_bn_bwd_channel_next_dup_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ppea_bn_bwd_reduce_final_dup_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_bn_bwd_reduce_final_dup_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_conv_image_packed_bytes": [_i, _i],
    "ppea_conv_image_pack_weights": [_vp, _i, _vp, _i, _i, _i, _vp],
    "ppea_conv_image_bf16": [_vp, _vp, _vp] + [_i] * 10 + [_vp],
    "ppea_conv_image_wgrad_workspace_bytes": [_i] * 5,
    "ppea_conv_image_wgrad_bf16": [_vp, _vp, _vp, _i, _vp] + [_i] * 10 + [_vp],
    "ppea_conv_packed_bytes": [_i] * 5,
    "ppea_conv_pack_weights": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_image_to_nhwc_bf16": [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp],
    "ppea_conv_nhwc_bf16": [_vp, _vp, _vp, _i, _vp] + [_i] * 15 + [_vp],
    "ppea_conv_wgrad_workspace_bytes": [_i] * 8,
    "ppea_conv_wgrad_nhwc_bf16": [_vp, _vp, _vp, _i, _vp] + [_i] * 12 + [_vp],
    "ppea_nhwc_up2cat_fwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_nhwc_up2cat_fwd_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_nhwc_up2cat_bwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_nhwc_up2cat_bwd_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ppea_dwconv_lk_stats_partials": [_i] * 6,
    "ppea_dwconv_lk_fwd_stats_bf16p": [_vp] * 6 + [_i] * 6 + [_vp],
    "ppea_pwconv_stats_partials": [_i] * 4,
    "ppea_pwconv_stats_bf16": [_vp] * 5 + [_i] * 4 + [_vp],
    "ppea_bn_finalize_sums_f32": [_vp, _i, _i, ctypes.c_long, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "ppea_conv2d_f32_fwd": [_vp] * 6 + [_i] * 9 + [_vp],
    "ppea_conv2d_f32_dgrad": [_vp] * 5 + [_i] * 11 + [_vp],
    "ppea_conv2d_f32_wgrad_workspace_bytes": [_i] * 7,
    "ppea_conv2d_f32_wgrad": [_vp] * 6 + [_i] * 11 + [_vp],
    "ppea_conv2d_bf16_fwd": [_vp] * 6 + [_i] * 9 + [_vp],
    "ppea_conv2d_bf16_dgrad": [_vp] * 5 + [_i] * 11 + [_vp],
    "ppea_conv2d_bf16_wgrad": [_vp] * 6 + [_i] * 11 + [_vp],
    "ppea_cost_volume_fwd_f32": [_vp] * 7 + [_i] * 5 + [_f, _vp],
    "ppea_nhwc_maxpool3x3s2_fwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_maxpool3x3s2_fwd_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_maxpool3x3s2_bwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_nhwc_maxpool3x3s2_bwd_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "ppea_loss_tail_blocks": [_l],
    "ppea_loss_tail_fwd_f32": [_vp] * 10 + [_i] * 4 + [_vp],
    "ppea_loss_tail_bwd_f32": [_vp] * 9 + [_i] * 3 + [_vp],
    "ppea_cost_volume_fwd_bf16": [_vp] * 8 + [_i] * 5 + [_f, _vp],
    "ppea_cost_volume_reduce_f32": [_vp] * 6 + [_i] * 4 + [_vp],
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C ppea-depth_amd/csrc). "
            "There is no CPU / PyTorch fallback for the PPEA-Depth hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.argtypes = argtypes
        fn.restype = ctypes.c_long if name.endswith("_bytes") else ctypes.c_int
    got = lib.ppea_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libppea_depth.so ABI {got} != expected {ABI_VERSION}: rebuild")
    return lib


lib = _load()


class PpeaKernelError(RuntimeError):
    pass


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PpeaKernelError("PPEA-Depth HIP kernels need tensors on a HIP device "
                              "(no CPU fallback); got a CPU tensor")
    if not t.is_contiguous():
        raise PpeaKernelError("non-contiguous tensor passed to a HIP kernel")
    if dtype is not None and t.dtype != dtype:
        raise PpeaKernelError(f"expected {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def check(err, name):
    if err != 0:
        what = {-1: "unsupported argument combination", -2: "inconsistent arguments"}.get(err, f"hipError_t {err}")
        raise PpeaKernelError(f"{name} failed: {what}")


def call(name, *args):
    check(getattr(lib, name)(*args), name)
