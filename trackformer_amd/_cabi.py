"""ctypes binding of include/tf_msda.h (libtf_msda.so).

This is the only place the package touches the native library.  There is deliberately no fallback:
if the library is missing, `lib()` raises, and so does every operator built on it.
"""
import ctypes
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TF_MSDA_LIB") or os.path.join(_PKG_DIR, "lib", "libtf_msda.so")

# Every symbol include/tf_msda.h declares (tests/test_cabi.py checks the list against the header).
EXPORTED_SYMBOLS = (
    "tf_msda_abi_version",
    "tf_msda_strerror",
    "tf_msda_last_hip_error",
    "tf_msda_last_kernel",
    "tf_msda_set_tiled",
    "tf_msda_set_option",
    "tf_msda_debug_trace_buffer",
    "tf_msda_forward_fused_f32",
    "tf_msda_forward_f32",
    "tf_msda_forward_f64",
    "tf_msda_forward_f32_dshapes",
    "tf_msda_forward_f64_dshapes",
    "tf_msda_backward_f32",
    "tf_msda_backward_f64",
    "tf_msda_backward_f32_dshapes",
    "tf_msda_backward_f64_dshapes",
    "tf_msda_forward_host_f32",
    "tf_msda_forward_host_f64",
    "tf_msda_backward_host_f32",
    "tf_msda_backward_host_f64",
    # include/tf_fused.h
    "tf_bias_act_f32",
    "tf_add_layernorm_f32",
    "tf_groupnorm_nhwc_f32",
    "tf_groupnorm_relu_nhwc_f32",
    "tf_box_refine_f32",
    "tf_postprocess_pack_f32",
    "tf_upsample_add_nhwc_f32",
    "tf_mask_label_map_f32",
    "tf_conv3x3_merge_packed_f32",
    "tf_groupnorm_stats_nhwc_f32",
    "tf_groupnorm_relu_conv3x3_c1_nhwc_f32",
    "tf_bias_relu_maxpool_f32",
    "tf_stem_conv7x7_f32",
    "tf_linear_split_f32",
    "tf_linear_split_res_f32",
    "tf_conv3x3_split_f32",
    "tf_conv3x3_splitk_f32",
    "tf_conv1x1_strided_split_f32",
    "tf_conv1x1_splitk_f32",
    "tf_linear_packed_bytes",
    "tf_linear_pack_weight_f32",
    "tf_linear_packed_f32",
    "tf_linear_split_add_f32",
    "tf_ffn_fused_f32",
    "tf_linear_res_ln_f32",
    "tf_conv_packed_f32",
    "tf_mha_core_f32",
    "tf_nms_host_f32",
)

ABI_VERSION = 3   # 3: the split-product entry points take w_lo / w_scale / terms (include/tf_fused.h)

_lib = None


class MSDAError(RuntimeError):
    """Raised when libtf_msda.so reports a non-zero status."""


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "trackformer_amd: native library %s not found. Build it with "
            "`python -m trackformer_amd.build` (needs hipcc); there is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).
    # It must be the one already mapped when libtf_msda.so is loaded, otherwise the process ends up
    # with two HIP runtimes and torch's streams / device pointers mean nothing to ours.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.tf_msda_abi_version.restype = ci
    L.tf_msda_abi_version.argtypes = []
    L.tf_msda_strerror.restype = ctypes.c_char_p
    L.tf_msda_strerror.argtypes = [ci]
    L.tf_msda_last_hip_error.restype = ci
    L.tf_msda_last_hip_error.argtypes = []
    L.tf_msda_last_kernel.restype = ctypes.c_char_p
    L.tf_msda_last_kernel.argtypes = []
    L.tf_msda_set_tiled.restype = ci
    L.tf_msda_set_tiled.argtypes = [ci]
    L.tf_msda_set_option.restype = ci
    L.tf_msda_set_option.argtypes = [ctypes.c_char_p, ci]
    L.tf_msda_debug_trace_buffer.restype = None
    L.tf_msda_debug_trace_buffer.argtypes = [vp]
    for suf in ("f32", "f64"):
        for tail in ("", "_dshapes"):
            f = getattr(L, "tf_msda_forward_%s%s" % (suf, tail))
            f.restype = ci
            f.argtypes = [vp] * 5 + [ci] * 7 + [vp]
            b = getattr(L, "tf_msda_backward_%s%s" % (suf, tail))
            b.restype = ci
            b.argtypes = [vp] * 8 + [ci] * 7 + [vp]
    for suf in ("f32", "f64"):   # host tensors: no stream argument, synchronous
        f = getattr(L, "tf_msda_forward_host_" + suf)
        f.restype = ci
        f.argtypes = [vp] * 5 + [ci] * 7
        b = getattr(L, "tf_msda_backward_host_" + suf)
        b.restype = ci
        b.argtypes = [vp] * 8 + [ci] * 7
    L.tf_nms_host_f32.restype = ci
    L.tf_nms_host_f32.argtypes = [vp, vp, ci, ctypes.c_float, vp, vp]
    L.tf_msda_forward_fused_f32.restype = ci
    L.tf_msda_forward_fused_f32.argtypes = [vp, vp, vp, ci, vp, ci, ci, ci, vp] + [ci] * 7 + [vp]
    L.tf_bias_act_f32.restype = ci
    L.tf_bias_act_f32.argtypes = [vp, vp, vp, ctypes.c_int64, ci, ci, vp]
    L.tf_add_layernorm_f32.restype = ci
    L.tf_add_layernorm_f32.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int64, ci, ctypes.c_float, vp]
    L.tf_stem_conv7x7_f32.restype = ci
    L.tf_stem_conv7x7_f32.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    L.tf_bias_relu_maxpool_f32.restype = ci
    L.tf_bias_relu_maxpool_f32.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
    L.tf_box_refine_f32.restype = ci
    L.tf_box_refine_f32.argtypes = [vp, vp, vp, ctypes.c_int64, ci, ctypes.c_float, vp]
    L.tf_postprocess_pack_f32.restype = ci
    L.tf_postprocess_pack_f32.argtypes = [vp, vp, vp, ctypes.c_int64, ci, ctypes.c_float, ctypes.c_float, ci, vp]
    L.tf_groupnorm_stats_nhwc_f32.restype = ci
    L.tf_groupnorm_stats_nhwc_f32.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_int64, vp]
    L.tf_conv3x3_merge_packed_f32.restype = ci
    L.tf_conv3x3_merge_packed_f32.argtypes = [vp, vp, vp, vp, vp, ci, ctypes.c_float, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.tf_mask_label_map_f32.restype = ci
    L.tf_mask_label_map_f32.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ctypes.c_float, vp]
    L.tf_upsample_add_nhwc_f32.restype = ci
    L.tf_upsample_add_nhwc_f32.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]
    L.tf_groupnorm_relu_conv3x3_c1_nhwc_f32.restype = ci
    L.tf_groupnorm_relu_conv3x3_c1_nhwc_f32.argtypes = [vp, vp, vp, vp, ctypes.c_float, vp, vp, ci, ci, ci, ci, ci, ctypes.c_float, vp]
    for fn in (L.tf_groupnorm_nhwc_f32, L.tf_groupnorm_relu_nhwc_f32):
        fn.restype = ci
        fn.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ctypes.c_float, ctypes.c_int64, ctypes.c_int64, vp]
    L.tf_linear_split_f32.restype = ci
    L.tf_linear_split_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, ci, ci, ci, vp]
    L.tf_linear_split_res_f32.restype = ci
    L.tf_linear_split_res_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, ci, ci, ci, vp]
    L.tf_conv3x3_splitk_f32.restype = ci
    L.tf_conv3x3_splitk_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.tf_conv3x3_split_f32.restype = ci
    L.tf_conv3x3_split_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp] + [ci] * 7 + [vp]
    L.tf_conv1x1_splitk_f32.restype = ci
    L.tf_conv1x1_splitk_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.tf_conv1x1_strided_split_f32.restype = ci
    L.tf_conv1x1_strided_split_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp] + [ci] * 7 + [vp]
    L.tf_linear_packed_bytes.restype = ctypes.c_int64
    L.tf_linear_packed_bytes.argtypes = [ci, ci, ci]
    L.tf_linear_pack_weight_f32.restype = ci
    L.tf_linear_pack_weight_f32.argtypes = [vp, vp, ci, ci, ci, vp]
    L.tf_linear_res_ln_f32.restype = ci
    L.tf_linear_res_ln_f32.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_float, vp, ctypes.c_int64, ci, ci, ci, vp]
    L.tf_ffn_fused_f32.restype = ci
    L.tf_ffn_fused_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_float, vp, ctypes.c_int64, ci, ci, ci, vp]
    L.tf_linear_split_add_f32.restype = ci
    L.tf_linear_split_add_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int64, ci, ci, vp]
    L.tf_linear_packed_f32.restype = ci
    L.tf_linear_packed_f32.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int64, ci, ci, ci, ci, vp]
    L.tf_conv_packed_f32.restype = ci
    L.tf_conv_packed_f32.argtypes = [vp, vp, vp, vp, vp, vp] + [ci] * 10 + [vp]
    L.tf_mha_core_f32.restype = ci
    L.tf_mha_core_f32.argtypes = [vp, vp, vp, vp, vp] + [ci] * 9 + [ctypes.c_float, vp]
    if L.tf_msda_abi_version() != ABI_VERSION:
        raise RuntimeError("libtf_msda.so ABI version %d != expected %d (rebuild)" %
                           (L.tf_msda_abi_version(), ABI_VERSION))
    _lib = L
    return _lib


def check(status, what):
    if status != 0:
        L = lib()
        msg = L.tf_msda_strerror(status).decode()
        raise MSDAError("%s failed: %s (status %d, hipError %d)" %
                        (what, msg, status, L.tf_msda_last_hip_error()))
