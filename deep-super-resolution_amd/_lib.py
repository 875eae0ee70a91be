"""ctypes binding of csrc/libdsr_hip.so (C ABI: include/dsr_hip.h).

The HIP library is the product: there is NO fallback.  If the shared object is missing
or a symbol cannot be resolved this module raises, and every op that reaches `lib()` on a
machine without a GPU fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "csrc", "libdsr_hip.so")

ABI_VERSION = 7          # bumped whenever a signature in include/dsr_hip.h changes; checked against the loaded library
BF16, F16 = 0, 1
ACT_NONE, ACT_LEAKY, ACT_PRELU, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_ELU = range(7)
PAD_ZERO, PAD_REFLECT, PAD_REPLICATE = range(3)


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("dtype", "N", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad", "pad_mode")]


class Epilogue(C.Structure):
    _fields_ = [("act", C.c_int), ("slope", C.c_float), ("prelu", C.c_void_p), ("bias", C.c_void_p),
                ("stats_partial", C.c_void_p), ("pixel_shuffle", C.c_int), ("out_nchw_f32", C.c_void_p),
                ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("residual", C.c_void_p)]


_P, _I, _F, _Z, _LL = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_longlong
_DESC = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol declared in include/dsr_hip.h
SIGNATURES = {
    "dsr_last_error": (C.c_char_p, []),
    "dsr_abi_version": (_I, []),
    "dsr_conv_dgrad_masked_supported": (_I, [_DESC]),
    "dsr_conv_dgrad_masked": (_I, [_DESC, _P, _P, _P, _I, _F, _P, _P]),
    "dsr_conv_dgrad_add_supported": (_I, [_DESC]),
    "dsr_conv_dgrad_add": (_I, [_DESC, _P, _P, _P, _P, _P]),
    "dsr_conv_wgrad_batchable": (_I, [_DESC]),
    "dsr_conv_wgrad_batched_workspace": (_Z, [_I, _DESC, C.POINTER(C.c_void_p)]),
    "dsr_conv_wgrad_batched": (_I, [_I, _DESC, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _P, _Z, _P]),
    "dsr_conv_kernel_name": (C.c_char_p, [_DESC, _I, C.POINTER(Epilogue)]),
    "dsr_clock_sample": (_I, [_P, _P]),
    "dsr_resample_u8": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "dsr_noise_gaussian_u8": (_I, [_P, _P, _I, _P, _Z, _P]),
    "dsr_salt_pepper_u8": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "dsr_scale_images_f32": (_I, [_P, _Z, _I, _P]),
    "dsr_patch_batch_u8": (_I, [_I, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                C.POINTER(C.c_int), _I, _I, _I, _P, _P]),
    "dsr_conv_fwd_affine_supported": (_I, [_DESC]),
    "dsr_conv_first_bwd_supported": (_I, [_DESC, _I]),
    "dsr_conv_first_bwd_workspace": (_Z, [_DESC]),
    "dsr_conv_first_bwd": (_I, [_DESC, _P, _P, _P, _I, _F, _P, _P, _P, _Z, _P]),
    "dsr_conv_first2_supported": (_I, [_DESC, _DESC]),
    "dsr_conv_first2_stats_rows": (_I, [_DESC]),
    "dsr_conv_first2_fwd": (_I, [_DESC, _DESC, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P]),
    "dsr_conv_first_bwd_recompute": (_I, [_DESC, _P, _P, _P, _P, _I, _F, _P, _P, _P, _Z, _P]),
    "dsr_conv_dgrad_ps_supported": (_I, [_DESC]),
    "dsr_conv_dgrad_ps_rows": (_I, [_DESC]),
    "dsr_conv_dgrad_ps": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _P]),
    "dsr_conv_dgrad_bn_supported": (_I, [_DESC]),
    "dsr_conv_dgrad_bn_rows": (_I, [_DESC]),
    "dsr_conv_dgrad_bn": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P]),
    "dsr_conv_dgrad_first_bwd_supported": (_I, [_DESC, _DESC, _I]),
    "dsr_conv_dgrad_first_bwd_workspace": (_Z, [_DESC]),
    "dsr_conv_dgrad_first_bwd": (_I, [_DESC, _DESC, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _Z, _P]),
    "dsr_conv_out_size": (_I, [_DESC, C.POINTER(_I), C.POINTER(_I)]),
    "dsr_conv_stats_rows": (_I, [_DESC]),
    "dsr_conv_packed_elems": (_Z, [_DESC, _I]),
    "dsr_conv_pack_weight": (_I, [_DESC, _P, _P, _P, _P]),
    "dsr_conv_pack_weight_multi": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "dsr_conv_fwd": (_I, [_DESC, _P, _P, C.POINTER(Epilogue), _P, _P]),
    "dsr_conv_dgrad_workspace": (_Z, [_DESC]),
    "dsr_conv_dgrad": (_I, [_DESC, _P, _P, _P, _P, _Z, _P]),
    "dsr_conv_wgrad_workspace": (_Z, [_DESC]),
    "dsr_conv_wgrad": (_I, [_DESC, _P, _P, _P, _P, _Z, _P]),
    "dsr_pw_nchw_to_nhwc": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dsr_pw_nhwc_to_nchw": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dsr_pw_pack_weight": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dsr_pw_scratch_rows": (_I, []),
    "dsr_pw_sum_rows": (_I, [_P, _I, _I, _I, _I, _F, _P, _I, _I, _P]),
    "dsr_pw_bn_finalize": (_I, [_P, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P]),
    "dsr_pw_bn_eval_affine": (_I, [_P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P]),
    "dsr_pw_reduce_blocks": (_I, [_Z, C.POINTER(_I)]),
    "dsr_pw_channel_stats": (_I, [_I, _P, _Z, _I, _I, _I, _P, _P]),
    "dsr_pw_bn_act_fwd": (_I, [_I, _P, _P, _P, _P, _P, _Z, _I, _I, _F, _P, _P]),
    "dsr_pw_bn_act_bwd_reduce": (_I, [_I, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _F, _P, _P, _P]),
    "dsr_pw_bn_bwd_finalize": (_I, [_P, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dsr_pw_bn_act_bwd_apply": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _F, _P, _I, _P]),
    "dsr_pw_act_bwd": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _I, _P, _P]),
    "dsr_pw_act_bwd_nchw": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dsr_pw_colsum": (_I, [_I, _P, _Z, _I, _I, _I, _P, _P]),
    "dsr_pw_add": (_I, [_I, _P, _P, _P, _Z, _P]),
    "dsr_pw_axpby_f32": (_I, [_P, _P, _F, _F, _P, _P, _Z, _P]),
    "dsr_pw_diff_loss": (_I, [_P, _P, _P, _Z, _I, _P, _I, _P]),
    "dsr_pw_bce_const": (_I, [_P, _I, _F, _P, _P, _I, _P]),
    "dsr_pw_adam": (_I, [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _P, _F, _P, _P]),
    "dsr_pw_adam_multi": (_I, [_I, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P, _F, _P]),
    "dsr_pw_incr": (_I, [_P, _P]),
    "dsr_cast16": (_I, [_I, _P, _P, _Z, _P]),
    "dsr_flatten": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dsr_linear_fwd_workspace": (_Z, [_I, _Z, _I]),
    "dsr_linear_fwd": (_I, [_I, _P, _P, _P, _I, _F, _P, _I, _Z, _I, _P, _Z, _P]),
    "dsr_linear_dgrad": (_I, [_I, _P, _P, _P, _I, _I, _Z, _P]),
    "dsr_linear_wgrad": (_I, [_I, _P, _P, _P, _I, _I, _Z, _P]),
    "dsr_linear_wgrad_gathered": (_I, [_I, _P, _P, _P, _I, _I, _Z, _I, _F, _P]),
    "dsr_linear_wgrad_adam": (_I, [_I, _P, _P, _I, _I, _Z, _I, _F, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _P]),
    "dsr_dense2_fwd": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "dsr_dense2_bwd": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    "dsr_maxpool2_fwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "dsr_maxpool2_bwd": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dsr_maxpool2_relu_bwd": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dsr_avgpool2_fwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "dsr_avgpool2_bwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "dsr_nearest2x_fwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "dsr_nearest2x_bwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "dsr_bilinear2x_fwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "dsr_bilinear2x_bwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "dsr_resize_norm_fwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "dsr_resize_norm_bwd": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "dsr_box_copy": (_I, [_P, _P] + [_I] * 16 + [_P]),
    "dsr_downsample_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dsr_downsample_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dsr_ssim_blocks": (_I, [_I, _I, _I]),
    "dsr_ssim_f32": (_I, [_P, _P, _I, _I, _I, _F, _P, _P]),
}

_lib = None

# bench.py's roofline leg: a list here makes every launching entry point record (name, start_event, end_event) on the
# stream it launches on (torch's current stream), so that the GPU-busy share of a step can be told from launch gaps.
LAUNCH_LOG = None
_NO_LAUNCH = ("dsr_last_error", "dsr_abi_version", "dsr_conv_dgrad_ps_supported", "dsr_conv_dgrad_ps_rows", "dsr_conv_dgrad_bn_supported", "dsr_conv_dgrad_bn_rows", "dsr_conv_dgrad_first_bwd_supported", "dsr_conv_dgrad_first_bwd_workspace", "dsr_conv_kernel_name", "dsr_conv_wgrad_batchable", "dsr_conv_wgrad_batched_workspace", "dsr_conv_dgrad_add_supported", "dsr_conv_dgrad_masked_supported", "dsr_conv_fwd_affine_supported",
              "dsr_conv_first2_supported", "dsr_conv_first2_stats_rows", "dsr_conv_first_bwd_supported", "dsr_conv_first_bwd_workspace", "dsr_conv_out_size", "dsr_conv_stats_rows",
              "dsr_conv_packed_elems", "dsr_conv_dgrad_workspace", "dsr_conv_wgrad_workspace", "dsr_pw_scratch_rows",
              "dsr_pw_reduce_blocks", "dsr_linear_fwd_workspace", "dsr_ssim_blocks")


class _Lib:
    """The loaded shared object; attribute access returns the bound C function (optionally timed, see LAUNCH_LOG)."""

    def __init__(self, handle):
        self._h = handle
        for name in SIGNATURES:
            fn = getattr(handle, name)
            setattr(self, name, fn if name in _NO_LAUNCH else self._timed(name, fn))

    @staticmethod
    def _timed(name, fn):
        def call(*args):
            log = LAUNCH_LOG
            if log is None:
                return fn(*args)
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            log.append((name, e0, e1))
            return rc
        return call


def lib():
    """The loaded library; raises if it was not built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"{SO_PATH} is missing: the HIP extension is the only implementation of this "
                               "package (no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        h = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = h.dsr_abi_version()
        if got != ABI_VERSION:
            # a stale libdsr_hip.so under a newer binding (or the reverse) passes arguments in the wrong slots: the
            # kernels then read sizes as pointers -- refuse to run instead (round-1 bring-up abort)
            raise RuntimeError(f"{SO_PATH}: C-ABI version {got}, this binding expects {ABI_VERSION}; rebuild with "
                               "`python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = _Lib(h)
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("dsr_hip: " + lib().dsr_last_error().decode())
    return rc
