"""Python front-end of the fused element-wise HIP kernels (include/tf_fused.h).

Used by the inference path only (eval mode, gradients disabled) on GPU tensors; everywhere else the
callers keep the plain PyTorch formulation (these helpers return None when they do not apply)."""
import torch
import torch.nn.functional as F

from . import _cabi


def _stream():
    return torch.cuda.current_stream().cuda_stream


def bias_act_(x, bias, residual=None, relu=True):
    """In place: x = act(x + bias[c] (+ residual)) for a channels_last [N,C,H,W] fp32 GPU tensor.
    Returns x, or None when the tensors do not qualify (caller falls back to ATen ops)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)):
        return None
    C = x.shape[1]
    if C % 4 or bias.dtype != torch.float32 or not bias.is_contiguous():
        return None
    if residual is not None and not (residual.shape == x.shape and residual.dtype == x.dtype
                                     and residual.is_contiguous(memory_format=torch.channels_last)):
        return None
    rc = _cabi.lib().tf_bias_act_f32(x.data_ptr(), bias.data_ptr(),
                                     0 if residual is None else residual.data_ptr(), x.numel(), C,
                                     1 if relu else 0, _stream())
    _cabi.check(rc, "tf_bias_act_f32")
    return x


def add_layernorm(x, res, norm):
    """LayerNorm(x + res) with `norm`'s affine parameters in one pass (res may be None).
    Returns a new tensor, or None when the fused kernel does not apply."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            and norm.elementwise_affine and len(norm.normalized_shape) == 1):
        return None
    C = norm.normalized_shape[0]
    if x.shape[-1] != C or C % 4 or C > 4096:
        return None
    if res is not None:
        if res.shape != x.shape or res.dtype != x.dtype:
            return None
        res = res.contiguous()
    out = torch.empty_like(x)
    rc = _cabi.lib().tf_add_layernorm_f32(x.data_ptr(), 0 if res is None else res.data_ptr(),
                                          norm.weight.data_ptr(), norm.bias.data_ptr(),
                                          out.data_ptr(), x.numel() // C, C, float(norm.eps),
                                          _stream())
    _cabi.check(rc, "tf_add_layernorm_f32")
    return out


def residual_norm(x, res, norm, inference):
    """norm(x + res): fused on the GPU inference path, plain PyTorch otherwise."""
    if inference:
        y = add_layernorm(x, res, norm)
        if y is not None:
            return y
    return norm(x + res)
