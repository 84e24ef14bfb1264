"""ctypes binding of the C ABI in ``include/ie_hip.h`` (``libie_hip.so``).

BASELINE.json's north_star names cffi for this layer; cffi is not installed in the build
image (probed), and the ABI is plain C, so ctypes is the loader.  The library is built
in-tree by ``__graft_entry__.build()`` / ``csrc/build.sh``.  If it is missing, every entry
point raises ``IEError`` -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IE_HIP_LIB") or os.path.join(_HERE, "libie_hip.so")   # IE_HIP_LIB: A/B builds

IE_F32, IE_F16, IE_BF16, IE_U8 = 0, 1, 2, 3
DTYPES = {"f32": IE_F32, "fp32": IE_F32, "float32": IE_F32, "f16": IE_F16, "fp16": IE_F16, "float16": IE_F16,
          "bf16": IE_BF16, "bfloat16": IE_BF16}
ACT = {"none": 0, "relu": 1, "lrelu": 2, "tanh": 3, "sigmoid": 4}
PAD = {"zeros": 0, "replicate": 1}
SUMSQ_WS_DOUBLES = 2048
VAR_SYNC_STAGE, VAR_NARROW_STORE, VAR_NO_DMA, VAR_UNFUSED, VAR_PURE_BF16, VAR_NO_K345, VAR_K67_LEGACY, VAR_P_SCHEDULE, VAR_K345_LEGACY = 1, 2, 4, 8, 16, 32, 64, 128, 256

# every exported symbol of include/ie_hip.h (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "ie_abi_version", "ie_last_error", "ie_device_info",
    "ie_nchw_f32_to_nhwc", "ie_nhwc_to_nchw_f32", "ie_nchw_f32_to_u8_nhwc", "ie_u8_nhwc_to_nchw_f32",
    "ie_resize_bilinear_nchw_f32",
    "ie_conv2d_nhwc_f32", "ie_pack_weight_hwio_f32", "ie_le_curve",
    "ie_convn_nhwc_f32", "ie_instnorm_nhwc_f32", "ie_maxpool2x2_nhwc_f32", "ie_resize_nhwc_f32", "ie_pixel_shuffle_nhwc_f32",
    "ie_conv_transpose2x2_nhwc_f32", "ie_pad_nhwc_f32", "ie_eltwise_nhwc_f32", "ie_max_concat_nhwc_f32",
    "ie_maxpool2x2_nhwc", "ie_mconv_pool_nhwc", "ie_mconv_tail_applicable", "ie_mconv_tail_nhwc", "ie_mconv_first_applicable", "ie_mconv_first_pool_nhwc", "ie_resize_nhwc", "ie_resize_pad_nhwc", "ie_pad_nhwc", "ie_pixel_shuffle_nhwc", "ie_eltwise_nhwc", "ie_cast_nhwc", "ie_concat_cast_nhwc", "ie_instnorm_nhwc", "ie_dwconv3x3_nhwc",
    "ie_mconv_packed_bytes", "ie_mconv_pack_weights", "ie_mconv_pack_weights_dev", "ie_mconv_nhwc", "ie_dsconv_nhwc",
    "ie_norm_forward_nhwc_f32", "ie_norm_backward_nhwc_f32", "ie_convn_wgrad_nhwc_f32", "ie_convn_dgrad_nhwc_f32",
    "ie_maxpool2x2_backward_nhwc_f32", "ie_resize_backward_nhwc_f32", "ie_crop_nhwc_f32", "ie_pixel_unshuffle_nhwc_f32",
    "ie_act_backward_f32", "ie_act_backward", "ie_pixel_unshuffle_nhwc", "ie_l1_loss_f32", "ie_ssim_forward_f32", "ie_ssim_backward_f32", "ie_sumsq_f32", "ie_adam_step_f32", "ie_adam_step_counted_f32", "ie_set_deterministic", "ie_get_deterministic", "ie_zdce_forward_u8", "ie_mconv_pair_applicable", "ie_mconv_pair_nhwc", "ie_mconv_pair_thin_applicable", "ie_mconv_pair_thin_nhwc", "ie_mconv_pair_pool_applicable", "ie_mconv_pair_pool_nhwc", "ie_retinex_decom_input", "ie_retinex_outputs_f32", "ie_unet_up_head_applicable", "ie_unet_up_head_nhwc", "ie_unet_first_packed_bytes", "ie_unet_first_pack_weights", "ie_unet_first_applicable", "ie_unet_first_pool_nchw", "ie_conv9x9_c4_packed_bytes", "ie_conv9x9_c4_pack_weights", "ie_conv9x9_c4_nhwc", "ie_conv3x3_c4_packed_bytes", "ie_conv3x3_c4_pack_weights", "ie_conv3x3_c4_nhwc", "ie_paste_nhwc", "ie_channel_stats_nhwc", "ie_channel_affine_nhwc", "ie_channel_affine_add_nhwc", "ie_simam_nhwc", "ie_retinex_recon_f32", "ie_l1_detached_f32", "ie_retinex_smooth_ws_floats",
    "ie_retinex_smooth_f32", "ie_pad_replicate_nhwc_f32", "ie_pad_replicate_backward_nhwc_f32", "ie_pad_replicate_nhwc", "ie_pad_replicate_backward_nhwc", "ie_zero_insert2_nhwc", "ie_rgb_to_hsv_nhwc_f32", "ie_hsv_to_rgb_nhwc_f32", "ie_rgb_to_hvi_nhwc_f32", "ie_hvi_to_rgb_nhwc_f32", "ie_le_curve_nhwc_f32", "ie_box_filter_nhwc_f32", "ie_guided_filter_nhwc_f32", "ie_fast_guided_filter_nhwc_f32", "ie_resize_bicubic_nhwc_f32", "ie_reduce_max_f32", "ie_median_blur_nhwc_f32", "ie_sobel_nhwc_f32", "ie_channel_stats_nhwc_f32", "ie_lin_coeffs_f32", "ie_stats_mean_f32", "ie_channel_affine_nhwc_f32", "ie_simam_nhwc_f32",
    "ie_norm_forward_nhwc", "ie_norm_backward_nhwc", "ie_norm_backward_pre_nhwc", "ie_convn_wgrad_nhwc", "ie_maxpool2x2_backward_nhwc", "ie_resize_backward_nhwc", "ie_crop_nhwc",
    "ie_zref_workspace_floats", "ie_zref_image_losses_f32", "ie_tv_loss_nhwc_f32", "ie_le_curve_backward",
    "ie_zdce_create", "ie_zdce_set_layer", "ie_zdce_finalize", "ie_zdce_reserve", "ie_zdce_forward",
    "ie_zdce_get_feature", "ie_zdce_workspace_bytes", "ie_zdce_destroy", "ie_zdce_time_forward",
    "ie_zdce_profile_forward", "ie_zdce_set_tuning",
]


class IEError(RuntimeError):
    pass


class ZdceConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("num_channels", C.c_int), ("num_iters", C.c_int),
                ("compute_dtype", C.c_int), ("device", C.c_int), ("variant", C.c_int)]


class Conv2dDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C0", C.c_int), ("C1", C.c_int),
                ("Cout", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("pad_mode", C.c_int), ("groups", C.c_int), ("act", C.c_int), ("slope", C.c_float),
                ("Ho", C.c_int), ("Wo", C.c_int)]


class ConvNDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("nsrc", C.c_int), ("Cout", C.c_int), ("kh", C.c_int),
                ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("pad_mode", C.c_int), ("act", C.c_int),
                ("slope", C.c_float), ("clamp", C.c_int), ("clamp_lo", C.c_float), ("clamp_hi", C.c_float),
                ("Ho", C.c_int), ("Wo", C.c_int)]


class MConvDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("nsrc", C.c_int), ("Cout", C.c_int), ("kh", C.c_int),
                ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("pad_mode", C.c_int), ("act", C.c_int),
                ("slope", C.c_float), ("clamp", C.c_int), ("clamp_lo", C.c_float), ("clamp_hi", C.c_float),
                ("in_dtype", C.c_int), ("out_dtype", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
                ("src_h", C.c_int * 8), ("src_w", C.c_int * 8), ("pixel_shuffle", C.c_int)]


_lib = None
ABI_VERSION = 2          # == IE_ABI_VERSION of include/ie_hip.h


def lib():
    """Load libie_hip.so once; raise loudly if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IEError(f"HIP library not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
                      f"g.build()'` (needs hipcc). There is no CPU fallback for the product path.")
    L = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.c_int, C.c_float
    L.ie_abi_version.restype = ip
    # (IE_HIP_LIB_ABI_OK=1: diagnostics only -- timing an older build whose struct layouts are known to match, tools/ab_bench.py)
    if L.ie_abi_version() != ABI_VERSION and not os.environ.get("IE_HIP_LIB_ABI_OK"):      # the ctypes structs below mirror ONE layout of include/ie_hip.h
        raise IEError(f"{LIB_PATH} reports ABI version {L.ie_abi_version()}, this loader was written against {ABI_VERSION}: "
                      f"rebuild the library (csrc/build.sh) -- a stale build would misread ie_mconv_desc.")
    L.ie_last_error.restype = C.c_char_p
    L.ie_device_info.argtypes = [ip, C.c_char_p, ip, C.POINTER(ip), C.POINTER(C.c_size_t)]
    L.ie_nchw_f32_to_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, vp]
    L.ie_nhwc_to_nchw_f32.argtypes = [vp, ip, vp, ip, ip, ip, ip, ip, vp]
    L.ie_nchw_f32_to_u8_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_u8_nhwc_to_nchw_f32.argtypes = [vp, vp, ip, ip, ip, ip, vp]
    L.ie_resize_bilinear_nchw_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_conv2d_nhwc_f32.argtypes = [C.POINTER(Conv2dDesc), vp, vp, vp, vp, vp, vp]
    L.ie_pack_weight_hwio_f32.argtypes = [vp, ip, ip, ip, ip, vp, vp]
    L.ie_le_curve.argtypes = [vp, vp, ip, vp, ip, ip, ip, ip, vp]
    L.ie_convn_nhwc_f32.argtypes = [C.POINTER(ConvNDesc), C.POINTER(vp), C.POINTER(ip), vp, vp, vp, vp, vp]
    L.ie_instnorm_nhwc_f32.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, fp, ip, fp, vp]
    L.ie_maxpool2x2_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, vp]
    L.ie_resize_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_pixel_shuffle_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_conv_transpose2x2_nhwc_f32.argtypes = [vp, vp, vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_pad_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_eltwise_nhwc_f32.argtypes = [ip, vp, vp, vp, C.c_size_t, ip, ip, fp, fp, vp]
    L.ie_max_concat_nhwc_f32.argtypes = [vp, vp, C.c_size_t, ip, vp]
    i64, sz = C.c_int64, C.c_size_t
    L.ie_maxpool2x2_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_resize_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_resize_pad_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_pad_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_pixel_shuffle_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, vp]
    L.ie_eltwise_nhwc.argtypes = [ip, vp, vp, vp, ip, C.c_size_t, ip, ip, vp]
    L.ie_cast_nhwc.argtypes = [vp, ip, ip, vp, ip, ip, C.c_size_t, vp]
    L.ie_instnorm_nhwc.argtypes = [vp, vp, vp, vp, ip, vp, ip, ip, ip, ip, ip, ip, fp, ip, fp, vp]
    L.ie_dwconv3x3_nhwc.argtypes = [vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, fp, vp]
    L.ie_mconv_packed_bytes.argtypes = [C.POINTER(ip), ip, ip, ip, ip]
    L.ie_mconv_packed_bytes.restype = C.c_size_t
    L.ie_mconv_pack_weights.argtypes = [vp, C.POINTER(ip), ip, ip, ip, ip, ip, vp, vp]
    L.ie_mconv_pack_weights_dev.argtypes = [vp, C.POINTER(ip), ip, ip, ip, ip, ip, ip, ip, vp, vp]
    L.ie_mconv_nhwc.argtypes = [C.POINTER(MConvDesc), C.POINTER(vp), C.POINTER(ip), vp, vp, vp, vp, vp]
    L.ie_mconv_pool_nhwc.argtypes = [C.POINTER(MConvDesc), C.POINTER(vp), C.POINTER(ip), vp, vp, vp, vp, vp]
    L.ie_mconv_first_applicable.argtypes = [ip, ip, ip]
    L.ie_mconv_first_pool_nhwc.argtypes = [ip, ip, ip, ip, ip, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ie_mconv_tail_applicable.argtypes = [C.POINTER(MConvDesc), C.POINTER(ip), ip]
    L.ie_mconv_pair_applicable.argtypes = [ip] * 11
    L.ie_mconv_pair_nhwc.argtypes = [vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_mconv_pair_thin_applicable.argtypes = [ip] * 11
    L.ie_mconv_pair_thin_nhwc.argtypes = [vp, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_mconv_pair_pool_applicable.argtypes = [ip] * 12
    L.ie_mconv_pair_pool_nhwc.argtypes = [vp, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_pad_replicate_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, vp]
    L.ie_pad_replicate_backward_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_zero_insert2_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_rgb_to_hvi_nhwc_f32.argtypes = [vp, vp, C.c_size_t, fp, fp, vp]
    L.ie_hvi_to_rgb_nhwc_f32.argtypes = [vp, vp, C.c_size_t, fp, fp, ip, vp]
    L.ie_concat_cast_nhwc.argtypes = [C.POINTER(vp), C.POINTER(ip), ip, vp, ip, ip, C.c_size_t, vp]
    L.ie_unet_first_packed_bytes.restype = C.c_size_t
    L.ie_unet_first_packed_bytes.argtypes = []
    L.ie_unet_first_pack_weights.argtypes = [vp, ip, vp, vp]
    L.ie_unet_first_applicable.argtypes = [ip, ip, ip]
    L.ie_unet_first_pool_nchw.argtypes = [vp, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, vp]
    L.ie_unet_up_head_applicable.argtypes = [ip] * 6
    L.ie_unet_up_head_nhwc.argtypes = [vp, vp, ip, ip, vp, vp, vp, vp, vp, vp, ip, ip, fp, fp, vp, ip, ip, ip, ip, vp]
    L.ie_retinex_decom_input.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_conv9x9_c4_packed_bytes.restype = C.c_size_t
    L.ie_conv9x9_c4_packed_bytes.argtypes = []
    L.ie_conv9x9_c4_pack_weights.argtypes = [vp, ip, vp, vp]
    L.ie_conv9x9_c4_nhwc.argtypes = [vp, vp, vp, vp, ip, ip, ip, ip, vp]
    L.ie_conv3x3_c4_packed_bytes.restype = C.c_size_t
    L.ie_conv3x3_c4_packed_bytes.argtypes = []
    L.ie_conv3x3_c4_pack_weights.argtypes = [vp, ip, vp, vp]
    L.ie_conv3x3_c4_nhwc.argtypes = [vp, vp, vp, vp, ip, ip, ip, ip, vp]
    L.ie_retinex_outputs_f32.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, vp]
    L.ie_mconv_tail_nhwc.argtypes = [C.POINTER(MConvDesc), C.POINTER(vp), C.POINTER(ip), vp, vp, vp, vp, ip, ip, fp, fp, vp, vp]
    L.ie_dsconv_nhwc.argtypes = [C.POINTER(MConvDesc), C.POINTER(vp), C.POINTER(ip), vp, vp, vp, vp, vp, vp]
    L.ie_norm_forward_nhwc_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, fp, ip, ip, ip, ip, ip, ip, ip, fp, ip, fp, vp]
    L.ie_norm_backward_nhwc_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, ip, fp, vp]
    L.ie_convn_wgrad_nhwc_f32.argtypes = [C.POINTER(ConvNDesc), C.POINTER(vp), C.POINTER(ip), vp, vp, vp, ip, vp, vp]
    L.ie_convn_dgrad_nhwc_f32.argtypes = [C.POINTER(ConvNDesc), vp, vp, C.POINTER(vp), C.POINTER(ip), C.POINTER(ip), vp]
    L.ie_maxpool2x2_backward_nhwc_f32.argtypes = [vp, vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_resize_backward_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_crop_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_pixel_unshuffle_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, vp]
    L.ie_act_backward_f32.argtypes = [vp, vp, vp, sz, ip, fp, fp, fp, ip, vp]
    L.ie_act_backward.argtypes = [vp, vp, vp, ip, sz, ip, fp, fp, fp, ip, vp]
    L.ie_pixel_unshuffle_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_l1_loss_f32.argtypes = [vp, vp, sz, vp, vp, fp, ip, vp]
    L.ie_ssim_forward_f32.argtypes = [vp, vp, ip, ip, ip, ip, i64, i64, i64, i64, fp, fp, fp, fp, vp, vp, vp]
    L.ie_ssim_backward_f32.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, ip, i64, i64, i64, i64, fp, fp, ip, ip, vp]
    L.ie_sumsq_f32.argtypes = [vp, sz, fp, vp, vp, ip, vp]
    L.ie_adam_step_f32.argtypes = [vp, vp, vp, vp, sz, ip, fp, fp, fp, fp, fp, fp, vp, fp, vp]
    L.ie_rgb_to_hsv_nhwc_f32.argtypes = [vp, vp, vp, sz, fp, vp]
    L.ie_hsv_to_rgb_nhwc_f32.argtypes = [vp, vp, vp, sz, vp]
    L.ie_le_curve_nhwc_f32.argtypes = [vp, vp, vp, vp, sz, ip, ip, ip, vp]
    L.ie_box_filter_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_guided_filter_nhwc_f32.argtypes = [vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, fp, vp]
    L.ie_fast_guided_filter_nhwc_f32.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, fp, ip, fp, fp, vp]
    L.ie_resize_bicubic_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, vp]
    L.ie_reduce_max_f32.argtypes = [vp, sz, vp, vp, vp]
    L.ie_median_blur_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_sobel_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, fp, vp]
    L.ie_channel_stats_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, vp]
    L.ie_lin_coeffs_f32.argtypes = [vp, vp, vp, vp, vp, vp, ip, ip, ip, fp, vp]
    L.ie_stats_mean_f32.argtypes = [vp, vp, ip, ip, ip, vp]
    L.ie_channel_affine_nhwc_f32.argtypes = [vp, vp, vp, vp, ip, ip, ip, ip, ip, fp, vp]
    L.ie_simam_nhwc_f32.argtypes = [vp, vp, vp, ip, ip, ip, ip, fp, vp]
    L.ie_set_deterministic.argtypes = [ip]
    L.ie_get_deterministic.argtypes = []
    L.ie_adam_step_counted_f32.argtypes = [vp, vp, vp, vp, sz, vp, fp, fp, fp, fp, fp, fp, vp, fp, vp]
    L.ie_norm_forward_nhwc.argtypes = [vp, vp, vp, vp, ip, vp, vp, vp, vp, vp, fp, ip, ip, ip, ip, ip, ip, ip, fp, ip, fp, vp]
    L.ie_norm_backward_nhwc.argtypes = [vp, vp, vp, ip, vp, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, ip, fp, vp]
    L.ie_norm_backward_pre_nhwc.argtypes = [vp, vp, ip, vp, vp, vp, vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, ip, fp, vp]
    L.ie_convn_wgrad_nhwc.argtypes = [C.POINTER(ConvNDesc), C.POINTER(vp), C.POINTER(ip), vp, ip, vp, vp, ip, vp, vp]
    L.ie_maxpool2x2_backward_nhwc.argtypes = [vp, vp, vp, ip, ip, ip, ip, ip, ip, vp]
    L.ie_resize_backward_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_crop_nhwc.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, ip, ip, ip, vp]
    L.ie_zref_workspace_floats.argtypes = [ip, ip, ip]
    L.ie_zref_workspace_floats.restype = C.c_size_t
    L.ie_zref_image_losses_f32.argtypes = [vp, vp, ip, ip, ip, fp, fp, fp, fp, vp, vp, vp, ip, vp]
    L.ie_tv_loss_nhwc_f32.argtypes = [vp, ip, ip, ip, ip, fp, vp, vp, ip, vp]
    L.ie_le_curve_backward.argtypes = [vp, vp, ip, vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_zdce_create.argtypes = [C.POINTER(ZdceConfig), C.POINTER(vp)]
    L.ie_zdce_set_layer.argtypes = [vp, ip, vp, vp, ip, ip]
    L.ie_zdce_finalize.argtypes = [vp, vp]
    L.ie_zdce_reserve.argtypes = [vp, ip, ip, ip]
    L.ie_zdce_forward.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, vp]
    L.ie_zdce_get_feature.argtypes = [vp, ip, vp, vp]
    L.ie_zdce_forward_u8.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, vp]
    L.ie_paste_nhwc.argtypes = [vp, vp] + [ip] * 9 + [vp]
    L.ie_channel_stats_nhwc.argtypes = [vp, ip, vp, ip, ip, ip, ip, vp]
    L.ie_channel_affine_nhwc.argtypes = [vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, C.c_float, vp]
    L.ie_channel_affine_add_nhwc.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, C.c_float, vp]
    L.ie_simam_nhwc.argtypes = [vp, vp, vp, ip, ip, ip, ip, ip, C.c_float, vp]
    L.ie_retinex_recon_f32.argtypes = [vp, ip, vp, ip, vp, C.c_float, vp, vp, vp, ip, ip, ip, vp]
    L.ie_l1_detached_f32.argtypes = [vp, vp, ip, ip, C.c_float, vp, vp, C.c_size_t, vp]
    L.ie_retinex_smooth_ws_floats.argtypes = [ip, ip, ip]
    L.ie_retinex_smooth_ws_floats.restype = C.c_size_t
    L.ie_retinex_smooth_f32.argtypes = [vp, ip, vp, ip, C.c_float, vp, vp, vp, vp, ip, ip, ip, vp]
    L.ie_pad_replicate_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp]
    L.ie_pad_replicate_backward_nhwc_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, vp]
    L.ie_zdce_workspace_bytes.argtypes = [vp]
    L.ie_zdce_workspace_bytes.restype = C.c_size_t
    L.ie_zdce_destroy.argtypes = [vp]
    L.ie_zdce_time_forward.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, ip, vp, C.POINTER(fp)]
    L.ie_zdce_profile_forward.argtypes = [vp, vp, vp, vp, vp, ip, ip, ip, vp, C.POINTER(fp),
                                          C.POINTER(C.c_char_p), C.POINTER(ip), ip, C.POINTER(ip)]
    L.ie_zdce_set_tuning.argtypes = [vp, ip, ip, ip]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("ie_abi_version",):
            fn.restype = ip
    _lib = L
    return L


def set_deterministic(on: bool = True) -> bool:
    """Process-wide: two-stage fixed-order reduction of the conv weight gradients instead of split-K fp32 atomics
    (include/ie_hip.h: ie_set_deterministic).  Returns the previous setting."""
    L = lib()
    prev = bool(L.ie_get_deterministic())
    L.ie_set_deterministic(1 if on else 0)
    return prev


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().ie_last_error()
        raise IEError(f"{what or 'ie call'} failed (status {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def device_info(device: int = 0):
    name = C.create_string_buffer(64)
    cu = C.c_int(0)
    mem = C.c_size_t(0)
    check(lib().ie_device_info(device, name, 64, C.byref(cu), C.byref(mem)), "ie_device_info")
    return {"arch": name.value.decode(), "cu": cu.value, "hbm_bytes": mem.value}
