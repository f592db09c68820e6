"""ctypes binding of libunetk.so (include/unetk.h).

The C ABI takes raw device pointers + sizes + a hipStream_t; PyTorch is used only to own device
memory and streams.  There is NO fallback: if the library is missing, loading raises, and every
op in boxsegliver_amd.ops goes through this module.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_void_p)

_LIB = None
LIB_PATH = os.environ.get("UNETK_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libunetk.so")

ABI_VERSION = 10            # must equal unetk_abi_version() of the loaded library (checked in lib())
UNETK_MAX_CLASSES = 8
W_NONE, W_NUMERICAL, W_PROPORTION, W_PIXELMAP = 0, 1, 2, 3


class ConvDesc(Structure):
    _fields_ = [("N", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32),
                ("x_stride", c_int32), ("y_stride", c_int32), ("precision", c_int32), ("dilation", c_int32)]


FP32, BF16, BF16S = 0, 1, 2      # BF16S: bf16 arithmetic + bf16 storage of activations (include/unetk.h)


class DeconvDesc(Structure):
    _fields_ = [("N", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32),
                ("out_stride", c_int32), ("out_coff", c_int32), ("precision", c_int32)]


class Conv3dDesc(Structure):
    _fields_ = [("N", c_int32), ("D", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32),
                ("kd", c_int32), ("sd", c_int32), ("shw", c_int32), ("x_stride", c_int32), ("y_stride", c_int32),
                ("cin_live8", ctypes.c_uint32 * 2), ("cout_live8", ctypes.c_uint32 * 2)]


class Deconv3dDesc(Structure):
    _fields_ = [("N", c_int32), ("D", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32),
                ("kd", c_int32), ("out_stride", c_int32), ("out_coff", c_int32), ("precision", c_int32)]


class NormDesc(Structure):
    _fields_ = [("N", c_int32), ("HW", c_int32), ("C", c_int32), ("per_sample", c_int32), ("z_stride", c_int32),
                ("guide_ch", c_int32), ("gw_stride", c_int32), ("gw_coff", c_int32), ("affine_only", c_int32),
                ("guide_leaky", c_int32), ("storage", c_int32), ("guide_alpha", c_float), ("dropout_keep", c_float),
                ("dropout_seed", ctypes.c_uint32), ("guide_per_sample", c_int32)]


class HeadDesc(Structure):
    _fields_ = [("N", c_int32), ("HW", c_int32), ("C", c_int32), ("ncls", c_int32),
                ("weight_mode", c_int32), ("numeric_w", c_float * UNETK_MAX_CLASSES),
                ("proportion_decay", c_float), ("storage", c_int32)]


class LitsDesc(Structure):
    _fields_ = [("N", c_int32), ("H", c_int32), ("W", c_int32), ("C", c_int32), ("n_slices", c_int32), ("src_h", c_int32),
                ("src_w", c_int32), ("lab_scale", c_int32), ("seed", ctypes.c_uint32), ("noise_scale", c_float)]


P = c_void_p
_SIGNATURES = {
    "unetk_lits_batch": (c_int, [POINTER(LitsDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "unetk_abi_version": (c_int, []),
    "unetk_nan_watch": (c_int, [P, P, c_int32, P]),
    "unetk_png_unfilter": (c_int, [P, c_int64, c_int, c_int, c_int, c_int, P, c_int64, P, P]),
    "unetk_conv3x3_fwd_affine_ok": (c_int, [POINTER(ConvDesc), c_int]),
    "unetk_conv3x3_fwd_affine": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, c_int, P, c_size_t, P]),
    "unetk_prof_reset": (c_int, [c_int]),
    "unetk_prof_enable": (c_int, [c_int]),
    "unetk_prof_mark": (c_int, []),
    "unetk_prof_read": (c_int, [c_int, c_int, POINTER(c_float)]),
    "unetk_prof_name": (c_int, [c_int, ctypes.c_char_p, c_int]),
    "unetk_error_string": (c_char_p, [c_int]),
    "unetk_conv3x3_pack": (c_int, [P, c_int, c_int, P, P, P]),
    "unetk_pack_item_blocks": (c_int, [c_int, c_int, c_int]),
    "unetk_pack_many": (c_int, [P, c_int, c_int, P]),
    "unetk_conv3x3_pack_bf16": (c_int, [P, c_int, c_int, P, P, P]),
    "unetk_conv3x3_pack_bf16s": (c_int, [P, c_int, c_int, P, P, P]),
    "unetk_conv3x3_stat_rows": (c_int, [POINTER(ConvDesc)]),
    "unetk_conv3x3_fwd": (c_int, [POINTER(ConvDesc), P, P, P, P, P]),
    "unetk_conv3x3_dgrad": (c_int, [POINTER(ConvDesc), P, P, P, P]),
    "unetk_conv3x3_ws_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "unetk_conv3x3_fwd_ws": (c_int, [POINTER(ConvDesc), P, P, P, P, P, c_size_t, P]),
    "unetk_conv3x3_dgrad_ws": (c_int, [POINTER(ConvDesc), P, P, P, P, c_size_t, P]),
    "unetk_conv3x3_dgrad_nbr_rows": (c_int, [POINTER(ConvDesc)]),
    "unetk_conv3x3_dgrad_nbr": (c_int, [POINTER(ConvDesc), P, P, P, P, c_int, P, P, P, P, c_int, P, P]),
    "unetk_conv3x3_wgrad_ws_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "unetk_conv3x3_wgrad": (c_int, [POINTER(ConvDesc), P, P, P, P, c_size_t, P]),
    "unetk_conv3d_pack": (c_int, [P, c_int, c_int, c_int, P, P, P]),
    "unetk_conv3d_out_dims": (c_int, [POINTER(Conv3dDesc), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "unetk_conv3d_stat_rows": (c_int, [POINTER(Conv3dDesc)]),
    "unetk_conv3d_ws_bytes": (c_size_t, [POINTER(Conv3dDesc)]),
    "unetk_conv3d_fwd": (c_int, [POINTER(Conv3dDesc), P, P, P, P, P, c_size_t, P]),
    "unetk_conv3d_dgrad": (c_int, [POINTER(Conv3dDesc), P, P, P, P, c_size_t, P]),
    "unetk_conv3d_wgrad": (c_int, [POINTER(Conv3dDesc), P, P, P, P, c_size_t, P]),
    "unetk_norm_finalize_ws_bytes": (c_size_t, [POINTER(NormDesc), c_int]),
    "unetk_norm_finalize": (c_int, [POINTER(NormDesc), P, c_int, P, P, c_float, c_float, c_int, P, P, P, P, P, P,
                                    P, c_size_t, P]),
    "unetk_norm_apply_relu": (c_int, [POINTER(NormDesc), P, P, P, P, P, P, P, P, P]),
    "unetk_norm_bwd_ws_bytes": (c_size_t, [POINTER(NormDesc)]),
    "unetk_norm_relu_bwd": (c_int, [POINTER(NormDesc), P, P, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                                    c_size_t, P]),
    "unetk_norm_relu_bwd_pre": (c_int, [POINTER(NormDesc), P, P, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int,
                                        P, c_size_t, P]),
    "unetk_norm_apply_relu_pool": (c_int, [POINTER(NormDesc), c_int, P, P, P, P, P, P]),
    "unetk_norm_relu_bwd_pool": (c_int, [POINTER(NormDesc), c_int, P, P, c_int, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "unetk_norm_drop_pool": (c_int, [POINTER(NormDesc), P, P, P, P, P]),
    "unetk_norm_se_bwd_add_drop": (c_int, [POINTER(NormDesc), P, P, P, P, P, P, P, P, P]),
    "unetk_norm_se_bwd_add": (c_int, [POINTER(NormDesc), P, P, P, P, P, P, P, P]),
    "unetk_fc_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, ctypes.c_uint32, P]),
    "unetk_fc_bwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "unetk_conv1d_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "unetk_conv1d_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "unetk_maxpool1d_fwd": (c_int, [P, P, c_int, c_int, c_int, P]),
    "unetk_maxpool1d_bwd": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "unetk_maxpool2_fwd": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, P]),
    "unetk_maxpool2_bwd": (c_int, [P, c_int, P, P, P, c_int, P, c_int, c_int, c_int, c_int, P]),
    "unetk_maxpool2_fwd_bf16": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, P]),
    "unetk_maxpool2_bwd_bf16": (c_int, [P, c_int, P, P, P, c_int, P, c_int, c_int, c_int, c_int, P]),
    "unetk_spatial_mean_fwd": (c_int, [P, P, c_int, c_int64, c_int, P]),
    "unetk_spatial_mean_bwd": (c_int, [P, P, c_int, c_int64, c_int, P]),
    "unetk_guide_moments": (c_int, [P, c_int, c_int64, c_int, c_int, P, P]),
    "unetk_avgpool2_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "unetk_image_gradients": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "unetk_sobel_concat": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "unetk_flip_axpy": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    "unetk_deconv2x2_pack": (c_int, [P, c_int, c_int, P, P, P]),
    "unetk_deconv2x2_pack_bf16": (c_int, [P, c_int, c_int, P, P, P]),
    "unetk_deconv2x2_pack_bf16s": (c_int, [P, c_int, c_int, P, P, P]),
    "unetk_deconv3d_pack_bf16": (c_int, [P, c_int, c_int, c_int, P, P, P]),
    "unetk_deconv2x2_fwd": (c_int, [POINTER(DeconvDesc), P, P, P, P, P]),
    "unetk_deconv2x2_bwd_ws_bytes": (c_size_t, [POINTER(DeconvDesc)]),
    "unetk_deconv2x2_bwd": (c_int, [POINTER(DeconvDesc), P, P, P, P, P, P, P, P, c_size_t, P]),
    "unetk_deconv3d_pack": (c_int, [P, c_int, c_int, c_int, P, P, P]),
    "unetk_deconv3d_fwd": (c_int, [POINTER(Deconv3dDesc), P, P, P, P, P]),
    "unetk_deconv3d_bwd_ws_bytes": (c_size_t, [POINTER(Deconv3dDesc)]),
    "unetk_deconv3d_bwd": (c_int, [POINTER(Deconv3dDesc), P, P, P, P, P, P, P, P, c_size_t, P]),
    "unetk_deconv3d_bwd_parts": (c_int, [POINTER(Deconv3dDesc), P, P, P, P, P, P, P, P, c_size_t, c_int, P]),
    "unetk_deconv2x2_bwd_parts": (c_int, [POINTER(DeconvDesc), P, P, P, P, P, P, P, P, c_size_t, c_int, P]),
    "unetk_head_result_floats": (c_size_t, [POINTER(HeadDesc)]),
    "unetk_head_ws_bytes": (c_size_t, [POINTER(HeadDesc)]),
    "unetk_head_fwd": (c_int, [POINTER(HeadDesc), P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "unetk_head_bwd": (c_int, [POINTER(HeadDesc), P, P, P, P, P, P, c_float, c_float, P, P, P, P, P, c_size_t, P]),
    "unetk_head_predict": (c_int, [P, c_int64, c_int, P, P, P]),
    "unetk_adam_step": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_float, c_float, P]),
    "unetk_momentum_step": (c_int, [P, P, P, c_int64, c_float, c_float, c_int, c_float, c_float, P]),
    "unetk_sumsq": (c_int, [P, c_int64, P, P, c_size_t, P]),
    "unetk_boundary_weights_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "unetk_boundary_weights": (c_int, [P, c_int, c_int, c_int, P, P, c_size_t, P]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))


class UnetkError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if libunetk.so is absent."""
    global _LIB
    if _LIB is None:
        # torch must be imported first: it bundles its own libamdhip64.so.7, and libunetk.so has to bind
        # to THAT runtime (same soname) -- loading /opt/rocm's copy first leaves two HIP runtimes in
        # the process and the second one sees no device.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise UnetkError(
                "libunetk.so not found at {} -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU / PyTorch fallback for the hot path)".format(LIB_PATH))
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.unetk_abi_version() != ABI_VERSION:
            raise UnetkError("{} has ABI version {}, this package binds version {} -- rebuild it".format(
                LIB_PATH, handle.unetk_abi_version(), ABI_VERSION))
        _LIB = handle
    return _LIB


def check(code, what):
    if code != 0:
        msg = lib().unetk_error_string(int(code))
        raise UnetkError("{} failed: {} (code {})".format(what, msg.decode() if msg else "?", code))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
