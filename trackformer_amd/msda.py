"""Multi-scale deformable attention: the operator, its autograd wrapper and the nn.Module.

Host-side mirror of the reference's operator interface (names, argument meaning, error behaviour):

    reference (under /root/reference/src/trackformer/models/ops/)          here
    --------------------------------------------------------------------   ------------------------------
    MSDA.ms_deform_attn_forward / _backward   src/vision.cpp:4-7            ms_deform_attn_forward / _backward
    MSDeformAttnFunction                      functions/ms_deform_attn_func.py:14-31   MSDeformAttnFunction
    MSDeformAttn                              modules/ms_deform_attn.py:16-89          MSDeformAttn

All compute goes through the C ABI of libtf_msda.so (include/tf_msda.h).  Like the reference's dispatcher
(src/ms_deform_attn.h:16-28,37-49) the operator branches on where `value` lives: device tensors -> the HIP
kernels on the current stream (no host fallback, ever); host tensors -> tf_msda_*_host_* of the same library
(csrc/msda_host.cpp), the real CPU path SURVEY.md section 8(b) asks for where the reference raises
"Not implemented on the CPU".  All tensors of a call must live on the same device; a missing library fails both.
"""
import ctypes
import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.init import constant_, xavier_uniform_

from . import _cabi, fused

import os as _os

# inference fast path of MSDeformAttn.forward (fused projections + fused operator prologue)
FUSED_INFERENCE = _os.environ.get("TF_MSDA_FUSED", "1") != "0"

_HOST_SHAPE_ATTR = "_tf_msda_host_shapes"
_shape_array_cache = {}


def attach_host_shapes(spatial_shapes, host_shapes):
    """Remember the (H_l, W_l) list of a device-resident `spatial_shapes` tensor on the tensor object.

    The reference keeps `spatial_shapes` only as an int64 device tensor (deformable_transformer.py:156)
    and reads it inside the kernel.  Callers that know the shapes on the host (our transformer does)
    attach them so that the operator can pass them by value and validate sum(H*W) == S without a
    device->host copy.  Purely an optimisation: without the attribute the *_dshapes entry points are used.
    """
    setattr(spatial_shapes, _HOST_SHAPE_ATTR, tuple((int(h), int(w)) for h, w in host_shapes))
    return spatial_shapes


def _host_shapes_of(spatial_shapes):
    hs = getattr(spatial_shapes, _HOST_SHAPE_ATTR, None)
    if hs is not None:
        return hs
    if not spatial_shapes.is_cuda:
        return tuple((int(h), int(w)) for h, w in spatial_shapes.tolist())
    return None


def _shape_array(host_shapes):
    arr = _shape_array_cache.get(host_shapes)
    if arr is None:
        flat = [v for hw in host_shapes for v in hw]
        arr = (ctypes.c_int64 * len(flat))(*flat)
        _shape_array_cache[host_shapes] = arr
    return arr


def _suffix(dtype):
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    raise RuntimeError("ms_deform_attn: only float32 and float64 are supported, got %s" % dtype)


def _check_inputs(value, spatial_shapes, sampling_loc, attn_weight, im2col_step):
    if not value.is_contiguous():
        raise RuntimeError("value tensor has to be contiguous")  # cu:29
    for name, t in (("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if value.is_cuda and not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)  # cu:33-34 (HIP device here)
        if t.device != value.device:
            raise RuntimeError("%s is on %s but value is on %s" % (name, t.device, value.device))
        if t.dtype != value.dtype:
            raise RuntimeError("%s has dtype %s but value has %s" % (name, t.dtype, value.dtype))
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("ms_deform_attn: expected value[N,S,M,D], sampling_loc[N,Lq,M,L,P,2], "
                           "attn_weight[N,Lq,M,L,P]")
    N, S, M, D = value.shape
    N2, Lq, M2, L, P, two = sampling_loc.shape
    if (N2, M2, two) != (N, M, 2) or tuple(attn_weight.shape) != (N, Lq, M, L, P):
        raise RuntimeError("ms_deform_attn: inconsistent tensor shapes")
    if spatial_shapes.dim() != 2 or tuple(spatial_shapes.shape) != (L, 2) \
            or spatial_shapes.dtype != torch.int64:
        raise RuntimeError("spatial_shapes must be an int64 tensor of shape [L, 2]")
    step = min(N, int(im2col_step))
    if step <= 0 or N % step != 0:  # cu:44-48
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (N, step))
    return N, S, M, D, L, Lq, P


def _host_shape_tensor(spatial_shapes):
    """Contiguous int64 [L, 2] in host memory (host-tensor calls)."""
    if spatial_shapes.is_cuda:
        raise RuntimeError("spatial_shapes is on %s but value is on the CPU" % spatial_shapes.device)
    return spatial_shapes.contiguous()


def _shape_args(spatial_shapes, value):
    """Returns (tail, pointer-ish, keepalive) for the host- or device-shape entry point."""
    hs = _host_shapes_of(spatial_shapes)
    if hs is not None:
        arr = _shape_array(hs)
        return "", ctypes.cast(arr, ctypes.c_void_p), arr
    if spatial_shapes.device != value.device:
        raise RuntimeError("spatial_shapes must be a CUDA tensor")  # cu:32
    ss = spatial_shapes.contiguous()
    return "_dshapes", ctypes.c_void_p(ss.data_ptr()), ss


def last_kernel():
    """Name of the device kernel this thread's most recent operator call enqueued (tf_msda_last_kernel, include/tf_msda.h):
    what the library dispatched, not what the options suggest -- bench.py labels its roofline with it."""
    return _cabi.lib().tf_msda_last_kernel().decode()


def ms_deform_attn_forward(value, spatial_shapes, sampling_loc, attn_weight, im2col_step=64):
    """value[N,S,M,D], spatial_shapes[L,2] i64, sampling_loc[N,Lq,M,L,P,2], attn_weight[N,Lq,M,L,P]
    -> output[N,Lq,M*D].   Same contract as the reference's MSDA.ms_deform_attn_forward
    (src/cuda/ms_deform_attn_cuda.cu:19-86); `im2col_step` is validated like the reference does and
    otherwise ignored (it never changed results)."""
    N, S, M, D, L, Lq, P = _check_inputs(value, spatial_shapes, sampling_loc, attn_weight,
                                         im2col_step)
    suf = _suffix(value.dtype)
    sampling_loc = sampling_loc.contiguous()
    attn_weight = attn_weight.contiguous()
    lib = _cabi.lib()
    if not value.is_cuda:   # host tensors: the library's host entry point (synchronous)
        out = torch.empty((N, Lq, M * D), dtype=value.dtype)
        shp = _host_shape_tensor(spatial_shapes)
        rc = getattr(lib, "tf_msda_forward_host_" + suf)(
            value.data_ptr(), shp.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(),
            N, S, M, D, L, Lq, P)
        _cabi.check(rc, "ms_deform_attn_forward")
        return out
    with torch.cuda.device(value.device):
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        tail, shp, keep = _shape_args(spatial_shapes, value)
        stream = torch.cuda.current_stream().cuda_stream
        rc = getattr(lib, "tf_msda_forward_%s%s" % (suf, tail))(
            value.data_ptr(), shp, sampling_loc.data_ptr(), attn_weight.data_ptr(), out.data_ptr(),
            N, S, M, D, L, Lq, P, stream)
    del keep
    _cabi.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, sampling_loc, attn_weight, grad_output,
                            im2col_step=64):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight], shaped like the respective inputs
    (reference: src/cuda/ms_deform_attn_cuda.cu:89-168)."""
    N, S, M, D, L, Lq, P = _check_inputs(value, spatial_shapes, sampling_loc, attn_weight,
                                         im2col_step)
    suf = _suffix(value.dtype)
    if grad_output.device != value.device or grad_output.dtype != value.dtype \
            or grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("grad_output must be a tensor of shape [N, Lq, M*D] on value's device with value's dtype")
    sampling_loc = sampling_loc.contiguous()
    attn_weight = attn_weight.contiguous()
    grad_output = grad_output.contiguous()  # autograd may hand over a strided view
    lib = _cabi.lib()
    if not value.is_cuda:
        grad_value, grad_loc, grad_attn = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
        shp = _host_shape_tensor(spatial_shapes)
        rc = getattr(lib, "tf_msda_backward_host_" + suf)(
            value.data_ptr(), shp.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
            grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(), N, S, M, D, L, Lq, P)
        _cabi.check(rc, "ms_deform_attn_backward")
        return [grad_value, grad_loc, grad_attn]
    with torch.cuda.device(value.device):
        grad_value = torch.empty_like(value)  # zero-filled by the library on the stream
        grad_loc = torch.empty_like(sampling_loc)
        grad_attn = torch.empty_like(attn_weight)
        tail, shp, keep = _shape_args(spatial_shapes, value)
        stream = torch.cuda.current_stream().cuda_stream
        rc = getattr(lib, "tf_msda_backward_%s%s" % (suf, tail))(
            value.data_ptr(), shp, sampling_loc.data_ptr(), attn_weight.data_ptr(),
            grad_output.data_ptr(), grad_value.data_ptr(), grad_loc.data_ptr(),
            grad_attn.data_ptr(), N, S, M, D, L, Lq, P, stream)
    del keep
    _cabi.check(rc, "ms_deform_attn_backward")
    return [grad_value, grad_loc, grad_attn]


def ms_deform_attn_forward_fused(value, spatial_shapes, reference_points, qproj, n_heads, n_levels,
                                 n_points):
    """Inference-only fused operator: softmax + sampling-location arithmetic + sampling in one launch.

    value [N,S,M,D]; reference_points [N,Lq,L,2|4]; qproj [N,Lq,3*M*L*P] = the query projected by the
    concatenated (sampling_offsets | attention_weights) Linear.  Equivalent to
    ms_deform_attn.py:69-86 followed by ms_deform_attn_forward; returns [N,Lq,M*D]."""
    hs = _host_shapes_of(spatial_shapes)
    if hs is None:
        raise RuntimeError("ms_deform_attn_forward_fused needs host-side level shapes "
                           "(attach_host_shapes)")
    N, S, M, D = value.shape
    Lq = qproj.shape[1]
    L, P = n_levels, n_points
    if M != n_heads or qproj.shape[-1] != 3 * M * L * P or reference_points.shape[:3] != (N, Lq, L):
        raise RuntimeError("ms_deform_attn_forward_fused: inconsistent tensor shapes")
    reference_points = reference_points.contiguous()
    lib = _cabi.lib()
    with torch.cuda.device(value.device):
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        arr = _shape_array(hs)
        rc = lib.tf_msda_forward_fused_f32(
            value.data_ptr(), ctypes.cast(arr, ctypes.c_void_p), reference_points.data_ptr(),
            reference_points.shape[-1], qproj.data_ptr(), qproj.shape[-1], 0, 2 * M * L * P,
            out.data_ptr(), N, S, M, D, L, Lq, P, torch.cuda.current_stream().cuda_stream)
    _cabi.check(rc, "ms_deform_attn_forward_fused")
    return out


class _CatProjection:
    """(sampling_offsets | attention_weights) as ONE Linear: the two projections share their input, so
    the inference path runs them as a single GEMM (refreshed when the parameters change)."""

    def __init__(self):
        self.key = None
        self.weight = None
        self.bias = None

    def get(self, mod):
        srcs = (mod.sampling_offsets.weight, mod.sampling_offsets.bias,
                mod.attention_weights.weight, mod.attention_weights.bias)
        key = tuple((t.data_ptr(), t._version, t.device, t.dtype) for t in srcs)
        if key != self.key:
            with torch.no_grad():
                self.weight = torch.cat([srcs[0], srcs[2]], 0).contiguous()
                self.bias = torch.cat([srcs[1], srcs[3]], 0).contiguous()
            self.key = key
        return self.weight, self.bias


class MSDeformAttnFunction(Function):
    """Autograd wrapper; mirrors functions/ms_deform_attn_func.py:14-31."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        ctx.host_shapes = _host_shapes_of(value_spatial_shapes)
        output = ms_deform_attn_forward(value, value_spatial_shapes, sampling_locations,
                                        attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, value_spatial_shapes, sampling_locations, attention_weights = ctx.saved_tensors
        if ctx.host_shapes is not None:
            attach_host_shapes(value_spatial_shapes, ctx.host_shapes)
        grad_value, grad_sampling_loc, grad_attn_weight = ms_deform_attn_backward(
            value, value_spatial_shapes, sampling_locations, attention_weights, grad_output,
            ctx.im2col_step)
        return grad_value, None, grad_sampling_loc, grad_attn_weight, None


class _Normed:
    """Marker: MSDeformAttn._forward already applied the caller's residual add and LayerNorm."""
    __slots__ = ("value",)

    def __init__(self, value):
        self.value = value


class MSDeformAttn(nn.Module):
    """Same constructor, parameters (state_dict keys) and forward contract as the reference module
    (modules/ms_deform_attn.py:16-89)."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, im2col_step=64):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, got %d and %d" %
                             (d_model, n_heads))
        self.im2col_step = im2col_step
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._cat_proj = _CatProjection()
        self._reset_parameters()

    def _reset_parameters(self):
        # modules/ms_deform_attn.py:34-47.  The 8-direction table of the reference is the set of
        # unit steps ordered (-1,-1),(-1,0),(-1,1),(0,-1),(0,1),(1,-1),(1,0),(1,1); like the
        # reference it only fits n_heads == 8.
        constant_(self.sampling_offsets.weight.data, 0.)
        dirs = [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1) if (a, b) != (0, 0)]
        if self.n_heads != len(dirs):
            raise ValueError("MSDeformAttn bias initialisation requires n_heads == 8")
        grid = torch.tensor(dirs, dtype=torch.float32).view(self.n_heads, 1, 1, 2)
        grid = grid.repeat(1, self.n_levels, self.n_points, 1)
        scale = torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter((grid * scale).reshape(-1))
        constant_(self.attention_weights.weight.data, 0.)
        constant_(self.attention_weights.bias.data, 0.)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_padding_mask=None, query_attn_mask=None, residual_norm=None, query_pos=None):
        """query[N,Lq,C], reference_points[N,Lq,L,2|4] in [0,1], input_flatten[N,S,C],
        input_spatial_shapes[L,2] (H_l,W_l), input_padding_mask[N,S] (True = padding)
        -> [N,Lq,C]   (modules/ms_deform_attn.py:49-89).
        residual_norm = (residual, nn.LayerNorm) (an extension used by the inference path of the layers): return
        norm(residual + attention output) instead -- the output projection, the add and the norm can then be one launch.
        query_pos (extension, inference path): the query is `query + query_pos`; the add can then ride in the projection."""
        if residual_norm is not None:
            out = self._forward(query, reference_points, input_flatten, input_spatial_shapes, input_padding_mask,
                                query_attn_mask, residual_norm, query_pos)
            if isinstance(out, _Normed):
                return out.value
            return fused.residual_norm(residual_norm[0], out, residual_norm[1], True)
        return self._forward(query, reference_points, input_flatten, input_spatial_shapes, input_padding_mask, query_attn_mask,
                             None, query_pos)

    def _forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_padding_mask, query_attn_mask,
                 residual_norm, query_pos=None):
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        hs = _host_shapes_of(input_spatial_shapes)
        if hs is not None:
            if sum(h * w for h, w in hs) != Len_in:
                raise AssertionError("sum of H_l*W_l does not match the flattened input length")
        else:  # reference behaviour (one device sync), ms_deform_attn.py:62
            assert (input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum() == Len_in

        M, L, P = self.n_heads, self.n_levels, self.n_points
        inference = not self.training and not torch.is_grad_enabled()
        value = fused.module_linear(self.value_proj, input_flatten, inference)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, Len_in, M, self.d_model // M)

        if (FUSED_INFERENCE and hs is not None and query_attn_mask is None and value.is_cuda
                and value.dtype == torch.float32 and not self.training
                and not torch.is_grad_enabled() and reference_points.shape[-1] in (2, 4)
                and (self.d_model // M) % 4 == 0 and P in (1, 2, 4, 8)):
            # inference: one GEMM for both query projections, prologue arithmetic inside the kernel
            w, b = self._cat_proj.get(self)
            qproj = None
            if query_pos is not None:   # opt-in: the positional add inside the projection GEMM
                qproj = fused.linear_add(query, query_pos, w, b)
                if qproj is None:
                    query = query + query_pos
            if qproj is None:
                qproj = fused.linear(query, w, b) if fused.split_linear_enabled() else None
            if qproj is None:
                qproj = F.linear(query, w, b)
            output = ms_deform_attn_forward_fused(value, input_spatial_shapes, reference_points,
                                                  qproj, M, L, P)
            if residual_norm is not None:
                y = fused.linear_residual_norm(output, self.output_proj, residual_norm[0], residual_norm[1])
                if y is not None:
                    return _Normed(y)
            return fused.module_linear(self.output_proj, output, True)

        if query_pos is not None:
            query = query + query_pos
        sampling_offsets = self.sampling_offsets(query).view(N, Len_q, M, L, P, 2)
        attention_weights = self.attention_weights(query).view(N, Len_q, M, L * P)
        attention_weights = F.softmax(attention_weights, -1).view(N, Len_q, M, L, P)
        if query_attn_mask is not None:
            attention_weights = attention_weights.masked_fill(
                query_attn_mask[..., None, None, None], float(0))

        if reference_points.shape[-1] == 2:
            # NB: the divisor is (H_l, W_l) applied to (x, y) offsets, as written in the reference
            # (ms_deform_attn.py:78-79); checkpoints were trained with it.
            sampling_locations = reference_points[:, :, None, :, None, :] \
                + sampling_offsets / input_spatial_shapes[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            sampling_locations = reference_points[:, :, None, :, None, :2] \
                + sampling_offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError('Last dim of reference_points must be 2 or 4, but get {} instead.'
                             .format(reference_points.shape[-1]))
        output = MSDeformAttnFunction.apply(value, input_spatial_shapes, sampling_locations,
                                            attention_weights, self.im2col_step)
        return self.output_proj(output)
