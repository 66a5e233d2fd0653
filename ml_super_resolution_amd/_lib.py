"""ctypes binding of libsrx.so (C ABI: include/srx.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is
raised.  Nothing here imports the oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libsrx.so')

SRX_OK = 0
PAD_SAME, PAD_VALID = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH, ACT_LRELU, ACT_SIGMOID = 0, 1, 2, 3, 4
OP_FWD, OP_BWD_DATA, OP_BWD_FILTER = 0, 1, 2

ACT_BY_NAME = {None: ACT_NONE, 'none': ACT_NONE, 'relu': ACT_RELU, 'tanh': ACT_TANH,
               'lrelu': ACT_LRELU, 'leaky_relu': ACT_LRELU, 'sigmoid': ACT_SIGMOID}
PAD_BY_NAME = {'same': PAD_SAME, 'valid': PAD_VALID}

# every symbol include/srx.h declares
EXPORTS = [
    'srx_version', 'srx_last_error', 'srx_set_conv_path', 'srx_set_wgrad_path', 'srx_conv2d_workspace_bytes', 'srx_conv2d_fwd',
    'srx_conv2d_bwd_data', 'srx_conv2d_bwd_filter', 'srx_conv2d_bwd_filter_partials', 'srx_conv2d_bwd_filter_reduce', 'srx_act_bwd', 'srx_depth_to_space',
    'srx_space_to_depth', 'srx_stream_copy', 'srx_mse_fwd_bwd', 'srx_l2_loss', 'srx_reduce_scratch_bytes',
    'srx_adam_tf_step', 'srx_adam_tf_step_dev', 'srx_momentum_clip_step', 'srx_rownorm_loss_fwd_bwd', 'srx_psnr', 'srx_ssim', 'srx_ssim_scratch_bytes', 'srx_saturate_u8', 'srx_affine', 'srx_u8_to_unit_float', 'srx_gaussian_blur', 'srx_resize_bilinear',
    'srx_upsample_nearest', 'srx_upsample_nearest_bwd', 'srx_add_relu_grad',
    'srx_conv2d_bwd_data_acc', 'srx_conv3x3_blocked', 'srx_conv3x3_blocked_bwd_filter_workspace_bytes', 'srx_conv3x3_blocked_bwd_filter',
    'srx_espcn_forward', 'srx_espcn_forward_keep', 'srx_debug_poison_lds', 'srx_srcnn_forward', 'srx_maxpool2x2', 'srx_maxpool2x2_bwd', 'srx_maxpool2x2_bwd_masked', 'srx_subsample2', 'srx_subsample2_bwd',
    'srx_channel_blocks_to_nhwc', 'srx_nhwc_to_channel_blocks', 'srx_channel_normalize', 'srx_channel_normalize_bwd',
    'srx_extract_patches16', 'srx_texture_gram', 'srx_texture_gram_bwd', 'srx_pil_resample_ksize', 'srx_pil_resample_coeffs', 'srx_resample_u8', 'srx_u8_to_pm1', 'srx_log_loss', 'srx_vgg_preprocess', 'srx_add_scaled', 'srx_resize_bicubic_tf', 'srx_column_sums', 'srx_gemm_workspace_bytes', 'srx_gemm',
]


class SrxError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ('N', 'H', 'W', 'Cin', 'Cout', 'KH', 'KW', 'stride', 'pad_mode', 'act',
                 'post_add_relu', 'precision', 'subpixel_r')]


class GemmDesc(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ('M', 'N', 'K', 'batch')] +
                [(n, ctypes.c_int64) for n in ('a_row_stride', 'a_col_stride', 'a_batch_stride', 'b_row_stride',
                                               'b_col_stride', 'b_batch_stride', 'c_row_stride', 'c_col_stride',
                                               'c_batch_stride')] +
                [('alpha', ctypes.c_float), ('act', ctypes.c_int32), ('accumulate', ctypes.c_int32)])


_lib = None


def lib():
    """Load libsrx.so once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SrxError('libsrx.so not found at %s -- build it with `make -C %s` (or '
                       '__graft_entry__.build()); there is no CPU fallback' %
                       (LIB_PATH, os.path.join(_HERE, 'csrc')))
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so, and the tensors handed to the library
    # live in that runtime.  Loading libsrx.so first would bind it to the system copy instead, and its launches then
    # fail with "no ROCm-capable device is detected" -- so torch's libraries are loaded before ours.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, i, f = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_float
    dp = ctypes.POINTER(ConvDesc)
    L.srx_version.restype = ctypes.c_char_p
    L.srx_last_error.restype = ctypes.c_char_p
    L.srx_set_conv_path.argtypes = [i]
    L.srx_set_wgrad_path.argtypes = [i]
    L.srx_conv2d_workspace_bytes.argtypes = [dp, i]
    L.srx_conv2d_workspace_bytes.restype = sz
    L.srx_reduce_scratch_bytes.restype = sz
    L.srx_conv2d_fwd.argtypes = [dp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.srx_conv2d_bwd_data.argtypes = [dp, vp, vp, vp, i, vp, vp, sz, vp]
    L.srx_conv2d_bwd_filter.argtypes = [dp, vp, vp, vp, vp, vp, f, vp, sz, vp]
    L.srx_conv2d_bwd_filter_partials.argtypes = [dp, vp, vp, vp, sz, ctypes.POINTER(ctypes.c_int), vp]
    L.srx_conv2d_bwd_filter_reduce.argtypes = [dp, vp, i, vp, vp, vp, f, vp]
    L.srx_act_bwd.argtypes = [vp, vp, vp, sz, i, vp]
    L.srx_depth_to_space.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.srx_space_to_depth.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.srx_stream_copy.argtypes = [vp, vp, sz, vp]
    L.srx_mse_fwd_bwd.argtypes = [vp, vp, sz, f, vp, i, vp, vp, vp]
    L.srx_l2_loss.argtypes = [vp, vp, sz, f, vp, i, vp, vp]
    L.srx_adam_tf_step.argtypes = [vp, vp, vp, vp, sz, f, f, f, f, ctypes.c_int64, f, vp]
    L.srx_adam_tf_step_dev.argtypes = [vp, vp, vp, vp, sz, vp, f, f, f, f, vp]
    L.srx_momentum_clip_step.argtypes = [vp, vp, vp, sz, f, f, f, f, vp]
    L.srx_rownorm_loss_fwd_bwd.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp]
    L.srx_psnr.argtypes = [vp, vp, vp, i, sz, f, vp]
    L.srx_ssim.argtypes = [vp, vp, vp, i, i, i, i, f, vp, vp]
    L.srx_ssim_scratch_bytes.argtypes = [i]
    L.srx_ssim_scratch_bytes.restype = sz
    L.srx_saturate_u8.argtypes = [vp, vp, sz, vp]
    L.srx_affine.argtypes = [vp, vp, sz, f, f, vp]
    L.srx_u8_to_unit_float.argtypes = [vp, vp, sz, vp]
    L.srx_gaussian_blur.argtypes = [vp, vp, vp, i, i, i, i, f, vp]
    L.srx_resize_bilinear.argtypes = [vp, vp, i, i, i, i, i, i, vp]
    L.srx_upsample_nearest.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.srx_upsample_nearest_bwd.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.srx_add_relu_grad.argtypes = [vp, vp, vp, vp, sz, vp]
    L.srx_conv2d_bwd_data_acc.argtypes = [dp, vp, vp, vp, vp, vp, sz, vp]
    L.srx_espcn_forward.argtypes = [vp] * 8 + [i, i, i, i, vp]
    L.srx_espcn_forward_keep.argtypes = [vp] * 10 + [i, i, i, i, vp]
    L.srx_debug_poison_lds.argtypes = [vp]
    L.srx_srcnn_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, vp]
    L.srx_conv3x3_blocked.argtypes = [vp, vp, vp, vp, i, vp, i, i, i, i, i, i, i, vp]
    L.srx_conv3x3_blocked_bwd_filter_workspace_bytes.restype = sz
    L.srx_conv3x3_blocked_bwd_filter_workspace_bytes.argtypes = [i, i, i, i, i]
    L.srx_conv3x3_blocked_bwd_filter.argtypes = [vp, vp, vp, vp, i, i, i, i, i, vp, sz, vp]
    L.srx_texture_gram.argtypes = [vp, vp, i, i, i, i, f, vp]
    L.srx_texture_gram_bwd.argtypes = [vp, vp, vp, i, i, i, i, f, f, vp]
    L.srx_pil_resample_ksize.argtypes = [i, i, i]
    L.srx_pil_resample_coeffs.argtypes = [i, i, i, vp, vp]
    L.srx_resample_u8.argtypes = [vp, vp, ctypes.c_long, i, i, ctypes.c_long, vp, vp, i, vp]
    L.srx_u8_to_pm1.argtypes = [vp, vp, sz, vp]
    L.srx_maxpool2x2.argtypes = [vp, vp, i, i, i, i, vp]
    L.srx_maxpool2x2_bwd.argtypes = [vp, vp, vp, i, i, i, i, vp]
    L.srx_maxpool2x2_bwd_masked.argtypes = [vp, vp, vp, i, i, i, i, i, vp]
    L.srx_subsample2.argtypes = [vp, vp, i, i, i, i, i, i, vp]
    L.srx_subsample2_bwd.argtypes = [vp, vp, i, i, i, i, i, i, vp]
    L.srx_channel_blocks_to_nhwc.argtypes = [vp, vp, sz, i, vp]
    L.srx_nhwc_to_channel_blocks.argtypes = [vp, vp, sz, i, vp]
    L.srx_channel_normalize.argtypes = [vp, vp, sz, i, f, vp]
    L.srx_channel_normalize_bwd.argtypes = [vp, vp, vp, sz, i, f, vp]
    L.srx_extract_patches16.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.srx_log_loss.argtypes = [vp, f, i, f, f, f, vp, i, vp, vp]
    L.srx_vgg_preprocess.argtypes = [vp, vp, sz, i, vp]
    L.srx_column_sums.argtypes = [vp, vp, i, i, i, vp]
    L.srx_add_scaled.argtypes = [vp, vp, vp, sz, f, f, vp]
    L.srx_resize_bicubic_tf.argtypes = [vp, vp, i, i, i, i, i, i, vp]
    L.srx_gemm_workspace_bytes.argtypes = [i, i, i, i]
    L.srx_gemm_workspace_bytes.restype = sz
    L.srx_gemm.argtypes = [ctypes.POINTER(GemmDesc), vp, vp, vp, vp, vp, sz, vp]
    for name in EXPORTS:
        getattr(L, name)          # AttributeError if the library is stale
    _lib = L
    return L


def check(rc, what):
    if rc != SRX_OK:
        raise SrxError('%s failed (status %d): %s' % (what, rc, lib().srx_last_error().decode()))
