"""ResNet backbone with frozen batch-norm, multi-level outputs and per-level position encodings.

Covers models/backbone.py of the reference (FrozenBatchNorm2d :19-55, BackboneBase :58-88,
Backbone :91-104, Joiner :107-122, build_backbone :125-134) plus the torchvision pieces it pulls in
(torchvision.models.resnet50/101 "v1.5" and IntermediateLayerGetter), which are restated here from
the published architecture because torchvision is not part of this build.  Module / parameter /
buffer names are identical to torchvision's, so reference checkpoints (`backbone.0.body.layerX...`)
load with strict=True.

MI355X notes: in inference (eval mode, grad disabled) every conv + FrozenBN pair runs as ONE conv with
the BN scale folded into the weights and the shift as its bias (the fold is cached and refreshed when
the parameters change), ReLU is applied in place, and activations stay in channels_last so MIOpen
picks NHWC kernels; that removes ~100 elementwise passes over the feature maps per frame.
"""
from collections import OrderedDict
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import nn

from . import fused
from .nested import NestedTensor, all_valid_mask, is_all_valid
from .position_encoding import build_position_encoding


import os as _os

# activations / folded weights in NHWC during inference (MIOpen's fp32 implicit-GEMM kernels are NHWC)
CHANNELS_LAST = _os.environ.get("TF_BACKBONE_NCHW", "0") != "1"


# DEFAULT since round 3 (TF_CONV1X1_SPLIT=0 / set_conv1x1_split(False) switches it off; measured per shape on MI355X in
# profiles/r03_optin_conv_per_layer.txt: 1.03-1.99x the library convolution + bias_act on 22 of 23 bottleneck shapes, 2835 ->
# 1973 us per 800 x 1333 frame with the 3 x 3 route below): the stride-1 1 x 1 convolutions of the
# bottlenecks (32 of ResNet-50's 53 convolutions, ~half of its flops) run as the library's own split-product GEMM
# (fused.linear: hi.hi + hi.mid + mid.hi on the bf16 matrix cores, fp32 accumulate) with the FrozenBN shift, the
# identity branch and the ReLU in its epilogue -- on channels_last activations such a convolution IS a GEMM over the
# N*H*W pixels, and the separate bias_act pass disappears.  Three-term products leave the full-size model inside the
# 1e-3 bar but not at fp32 agreement (tools/experiments/bf16_split_linear.py x3 conv: logits off by ~7e-5).
_conv1x1_split = _os.environ.get("TF_CONV1X1_SPLIT", "1") != "0"


# DEFAULT as well (TF_CONV3X3_SPLIT=0 / set_conv3x3_split(False) switches it off): the bottlenecks' 3 x 3 convolutions (stride 1 and 2) as the
# same split product, an implicit GEMM over the output pixels (fused.conv3x3 -> tf_conv3x3_split_f32), FrozenBN shift and
# ReLU in its epilogue.  (The three strided 1 x 1 projections of the identity branch go through the same kernel with a
# 1 x 1 window under TF_CONV1X1_SPLIT.)  With both routes on, only the 7 x 7 stem stays in MIOpen.
_conv3x3_split = _os.environ.get("TF_CONV3X3_SPLIT", "1") != "0"


# Per-shape exceptions to the two routes above: "cin x cout x kernel x stride" entries (TF_CONV_SPLIT_SKIP="64x64x3x1,256x64x1x1"
# / set_conv_split_skip) keep the library convolution -- tools/bench_conv.py times every bottleneck shape both ways.
def _parse_skip(text):
    out = set()
    for item in text.replace(" ", "").split(","):
        if item:
            parts = item.lower().split("x")
            if len(parts) != 4 or not all(p.isdigit() for p in parts):
                raise ValueError("TF_CONV_SPLIT_SKIP entries are cin x cout x kernel x stride, e.g. 64x64x3x1: %r" % item)
            out.add(tuple(int(p) for p in parts))
    return frozenset(out)


_conv_split_skip = _parse_skip(_os.environ.get("TF_CONV_SPLIT_SKIP", ""))


def set_conv_split_skip(shapes):
    """shapes: iterable of (cin, cout, kernel, stride) that keep the library convolution; returns the previous set."""
    global _conv_split_skip
    prev, _conv_split_skip = _conv_split_skip, frozenset(tuple(int(v) for v in s) for s in shapes)
    return prev


def _split_route_allowed(conv):
    return (conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.stride[0]) not in _conv_split_skip


def set_conv3x3_split(on):
    """Route the bottlenecks' 3 x 3 convolutions through the split-product kernel (process-wide); returns the previous setting."""
    global _conv3x3_split
    prev, _conv3x3_split = _conv3x3_split, bool(on)
    return prev


def set_conv1x1_split(on):
    """Route the stride-1 1 x 1 convolutions through the split-product GEMM (process-wide); returns the previous setting."""
    global _conv1x1_split
    prev, _conv1x1_split = _conv1x1_split, bool(on)
    return prev


def conv1x1_as_gemm(x, w2d, bias, residual, relu, linear_fn):
    """1 x 1 stride-1 convolution of a channels_last activation as a GEMM over its pixels.
    x [N, Cin, H, W] (channels_last), w2d [Cout, Cin], bias [Cout] or None, residual like the output or None;
    linear_fn(x2, w2d, bias, relu, residual2) -> [N*H*W, Cout] or None.  Returns [N, Cout, H, W] (channels_last) or None."""
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        return None
    n, cin, h, w = x.shape
    cout = w2d.shape[0]
    x2 = x.permute(0, 2, 3, 1).reshape(n * h * w, cin)          # a view: NHWC storage
    r2 = None
    if residual is not None:
        if residual.shape != (n, cout, h, w) or not residual.is_contiguous(memory_format=torch.channels_last):
            return None
        r2 = residual.permute(0, 2, 3, 1).reshape(n * h * w, cout)
    y2 = linear_fn(x2, w2d, bias, relu, r2)
    if y2 is None:
        return None
    return y2.view(n, h, w, cout).permute(0, 3, 1, 2)            # NCHW shape over NHWC storage = channels_last


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters, stored as buffers
    (y = x * w * rsqrt(var + 1e-5) + (b - mean * w * rsqrt(var + 1e-5)), backbone.py:45-55)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        state_dict.pop(prefix + 'num_batches_tracked', None)  # plain BatchNorm checkpoints carry it
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)

    def scale_shift(self):
        scale = self.weight * (self.running_var + 1e-5).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


class _FoldCache:
    """conv weight with the following FrozenBN folded in; refreshed when any source tensor changes.  Only the image the
    active route needs is materialised: the [Cout, Cin] / tap-major [Cout, 9 Cin] matrix of the split-product routes (the
    defaults), or the 4-d weight of the library convolution when a layer takes that path (route switched off, shape on the
    skip list, CPU tensors) -- not all of them for every layer (~90 MB of copies for ResNet-50 otherwise)."""

    def __init__(self):
        self.key = None
        self.bias = None
        self._conv = self._bn = None
        self._weight = None        # 4-d, library path
        self._weight2d = None      # [Cout, Cin] of a 1 x 1 convolution (a persistent tensor: fused.linear caches its pieces on it)
        self._weight_taps = None   # [Cout, 9 * Cin] of a 3 x 3 convolution, tap-major

    def _scaled(self):
        with torch.no_grad():
            scale, _ = self._bn.scale_shift()
            return self._conv.weight * scale.reshape(-1, 1, 1, 1)

    @staticmethod
    def _published(t):
        """A tensor built on the caller's current stream is about to be cached for EVERY stream (one lane per sequence, each on
        its own HIP stream, share the model): wait for the building stream first (fused._publish_barrier), then hand it out."""
        if t.is_cuda:
            fused._publish_barrier(t.device)
        return t

    @property
    def weight(self):
        if self._weight is None:
            w = self._scaled()
            self._weight = self._published(w.contiguous(memory_format=torch.channels_last) if CHANNELS_LAST else w.contiguous())
        return self._weight

    @property
    def weight2d(self):
        if self._weight2d is None and self._conv.kernel_size == (1, 1):
            w = self._scaled()
            self._weight2d = fused.inherit_route(self._published(w.reshape(w.shape[0], w.shape[1]).contiguous()), self._conv.weight)
        return self._weight2d

    @property
    def weight_taps(self):
        if self._weight_taps is None and self._conv.kernel_size == (3, 3):
            w = self._scaled()
            # [Cout, 3, 3, Cin] -> [Cout, 9 * Cin]: the storage order of the channels_last weight (tap-major K)
            self._weight_taps = fused.inherit_route(self._published(w.permute(0, 2, 3, 1).reshape(w.shape[0], 9 * w.shape[1]).contiguous()),
                                                    self._conv.weight)
        return self._weight_taps

    def get(self, conv: nn.Conv2d, bn: FrozenBatchNorm2d):
        """-> the folded shift (bias); the weight images are the properties above."""
        srcs = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t.data_ptr(), t._version, t.device, t.dtype) for t in srcs)
        if key != self.key:
            with torch.no_grad():
                _, shift = bn.scale_shift()
                self.bias = self._published(shift.contiguous())
            self._conv, self._bn = conv, bn
            # a six-term route (fused.route_six_terms / audit_activation_range) lives on the image it was given for: the images
            # re-created for changed weights inherit it through the convolution's parameter
            if any(fused._routed(t) for t in (self._weight2d, self._weight_taps) if t is not None):
                fused.route_six_terms(conv.weight)
            self._weight = self._weight2d = self._weight_taps = None
            self.key = key
        return self.bias


def _inference_mode(module: nn.Module) -> bool:
    return (not module.training) and (not torch.is_grad_enabled())


# Round 6 (TF_TRAIN_FOLD=0 / set_train_fold(False): the reference's module graph -- convolution, FrozenBN as mul + add, residual
# add, ReLU, each an autograd op over the whole feature map): the conv + FrozenBN pairs of a TRAINING step.  Measured on the cfg-3
# step (profiles/r06_train_step_breakdown.txt): ~17 ms of a 96 ms step were element-wise passes, most of them these.
#   * no graph is recorded (the previous-frame pass of DETRTrackingBase.forward runs under no_grad in training mode,
#     detr_tracking.py:219-277; the backbone has no dropout): the inference kernels, as in eval mode;
#   * a frozen block whose input carries no graph (stem + layer1: backbone.py:66-69 of the reference freezes everything outside
#     layer2-4): the inference kernels as well -- nothing there is differentiated;
#   * a trainable block: the BN scale folded into the weight as an autograd op on the WEIGHT (Cout x Cin x k x k, not the feature map),
#     the library convolution, then shift / residual / ReLU in ONE in-place pass whose backward is one threshold pass (_BiasAct).
_train_fold = _os.environ.get("TF_TRAIN_FOLD", "1") != "0"
_TRAIN_NOGRAD_LIBRARY = _os.environ.get("TF_TRAIN_FOLD", "1") == "lib"


def set_train_fold(on: bool) -> bool:
    global _train_fold
    prev, _train_fold = _train_fold, bool(on)
    return prev


def _frozen(module: nn.Module) -> bool:
    f = module.__dict__.get("_tf_frozen")
    if f is None or f[0] != _frozen_epoch[0]:
        f = (_frozen_epoch[0], not any(p.requires_grad for p in module.parameters()))
        module.__dict__["_tf_frozen"] = f
    return f[1]


_frozen_epoch = [0]   # bump (backbone.refresh_frozen()) after changing requires_grad of backbone parameters by hand


def refresh_frozen():
    _frozen_epoch[0] += 1


def _fold_mode(module: nn.Module, x) -> int:
    """0: the module graph; 1: the inference kernels (folded BN, fused epilogues, split products); 2: training with the fold as an
    autograd op on the weight and the fused epilogue pass."""
    if not torch.is_grad_enabled():
        return 1 if (not module.training or (_train_fold and x.is_cuda)) else 0
    if not (module.training and _train_fold and x.is_cuda and CHANNELS_LAST):
        return 0   # (eval mode with gradients -- gradient checks, saliency -- keeps the module graph)
    if not x.requires_grad and _frozen(module):
        return 1
    return 2


class _BiasAct(torch.autograd.Function):
    """y <- act(y + shift[c] (+ residual)) in place on a convolution's output (tf_bias_act_f32); backward: one threshold pass,
    the same gradient to the convolution output and to the residual (shift is a buffer of the FrozenBN)."""

    @staticmethod
    def forward(ctx, y, shift, residual, relu):
        out = fused.bias_act_(y, shift, residual, relu)
        if out is None:   # layout / alignment the kernel does not take: the same arithmetic by ATen, still in place
            y.add_(shift.reshape(1, -1, 1, 1))
            if residual is not None:
                y.add_(residual)
            if relu:
                y.relu_()
        ctx.mark_dirty(y)
        ctx.relu = relu
        ctx.has_res = residual is not None
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        if ctx.relu:
            (y,) = ctx.saved_tensors
            g = torch.ops.aten.threshold_backward(g, y, 0)
        return g, None, (g if ctx.has_res else None), None


def _conv_bn(x, conv: nn.Conv2d, bn: nn.Module, cache: _FoldCache, relu: bool, fold: bool,
             residual=None):
    """conv -> frozen BN (-> + residual) (-> ReLU).  Inference on the GPU: one library convolution with
    the BN scale folded into its weights, then ONE fused pass for shift / residual / ReLU."""
    if fold == 2 and isinstance(bn, FrozenBatchNorm2d) and x.is_cuda:
        scale, shift = bn.scale_shift()
        y = F.conv2d(x, conv.weight * scale.reshape(-1, 1, 1, 1), None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if y.requires_grad:
            return _BiasAct.apply(y, shift, residual, relu)
        fold = 0   # (nothing here is differentiated and the block is not frozen either: cannot happen for a trainable weight)
    if fold == 1 and isinstance(bn, FrozenBatchNorm2d) and conv.training and x.is_cuda and not getattr(fused._tls, "one_stream", 0):
        with fused.one_stream():   # a training step: images rebuilt per step, one stream
            return _conv_bn(x, conv, bn, cache, relu, fold, residual)
    if fold == 1 and isinstance(bn, FrozenBatchNorm2d):
        b = cache.get(conv, bn)
        if x.is_cuda:
            split_ok = _split_route_allowed(conv)   # False: this shape keeps the library convolution (TF_CONV_SPLIT_SKIP)
            if _TRAIN_NOGRAD_LIBRARY and conv.training and conv.weight.requires_grad:
                split_ok = False   # (A/B aid: a trainable layer's packed image is rebuilt after every optimiser step)
            if (split_ok and _conv1x1_split and cache.weight2d is not None and residual is None and conv.stride == (1, 1)
                    and conv.padding == (0, 0) and conv.groups == 1 and CHANNELS_LAST
                    and fused.conv1x1_wants_split_k(x.shape[0] * x.shape[2] * x.shape[3], conv.in_channels, conv.out_channels)):
                y = fused.conv3x3(x, cache.weight2d, b, relu, 1)   # few pixels under a long K (layer3 / layer4 conv1): split-K
                if y is not None:
                    return y
            if (split_ok and _conv1x1_split and cache.weight2d is not None and conv.stride == (1, 1) and conv.padding == (0, 0)
                    and conv.groups == 1 and CHANNELS_LAST):
                y = conv1x1_as_gemm(x, cache.weight2d, b, residual, relu,
                                    lambda x2, w2, bb, act, r2: fused.linear(x2, w2, bb, relu=act, residual=r2))
                if y is not None:
                    return y
            if (split_ok and _conv1x1_split and cache.weight2d is not None and residual is None and conv.stride == (2, 2)
                    and conv.padding == (0, 0) and conv.groups == 1 and CHANNELS_LAST):
                y = fused.conv3x3(x, cache.weight2d, b, relu, 2)   # the strided projection of the identity branch
                if y is not None:
                    return y
            if (split_ok and _conv3x3_split and cache.weight_taps is not None and residual is None and conv.padding == (1, 1)
                    and conv.dilation == (1, 1) and conv.groups == 1 and conv.stride in ((1, 1), (2, 2)) and CHANNELS_LAST):
                y = fused.conv3x3(x, cache.weight_taps, b, relu, conv.stride[0])
                if y is not None:
                    return y
            y = F.conv2d(x, cache.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
            if fused.bias_act_(y, b, residual, relu) is not None:
                return y
            y = y + b.reshape(1, -1, 1, 1)
        else:
            y = F.conv2d(x, cache.weight, b, conv.stride, conv.padding, conv.dilation, conv.groups)
        if residual is not None:
            y = y.add_(residual)
        return F.relu_(y) if relu else y
    if fold == 2 and not x.is_cuda:
        fold = 0
    x = bn(conv(x))
    if residual is not None:
        x = x + residual
    return F.relu(x) if relu else x


def _stem_pooled(x, conv1, bn1, maxpool, cache: _FoldCache):
    """Defaults since round 3 (fused.set_stem_pool_fused / TF_STEM_POOL_FUSED=0 switches it off): conv1 (BN scale folded in)
    and then BN shift + ReLU + MaxPool2d(3, 2, 1) in ONE pass over the convolution's output (tf_bias_relu_maxpool_f32;
    bit-identical to the separate passes); (fused.set_stem_conv_split / TF_STEM_CONV_SPLIT=0): conv1 itself as a split
    product (tf_stem_conv7x7_f32).  Returns None when both routes are off or do not apply."""
    if not ((fused.stem_pool_fused_enabled() or fused.stem_conv_split_enabled()) and x.is_cuda and CHANNELS_LAST
            and isinstance(bn1, FrozenBatchNorm2d)
            and isinstance(maxpool, nn.MaxPool2d) and maxpool.kernel_size == 3 and maxpool.stride == 2
            and maxpool.padding == 1 and maxpool.dilation == 1 and not maxpool.ceil_mode):
        return None
    b = cache.get(conv1, bn1)
    w = cache.weight   # the stem's packed image (fused.stem_conv) is cached on this tensor
    y = None
    if (conv1.kernel_size == (7, 7) and conv1.stride == (2, 2) and conv1.padding == (3, 3) and conv1.dilation == (1, 1)
            and conv1.groups == 1):
        # opt-in (fused.set_stem_conv_split): the convolution itself on the matrix cores; shift + ReLU go to whichever pass follows
        fuse_pool = fused.stem_pool_fused_enabled()
        y = fused.stem_conv(x, w, None if fuse_pool else b, relu=not fuse_pool)
        if y is not None and not fuse_pool:
            return maxpool(y)
    if y is None:
        if not fused.stem_pool_fused_enabled():
            return None
        y = F.conv2d(x, w, None, conv1.stride, conv1.padding, conv1.dilation, conv1.groups)
    pooled = fused.bias_relu_maxpool(y, b)
    if pooled is None:   # not applicable after all: the separate passes on the same convolution output
        if fused.bias_act_(y, b, None, True) is None:
            y = F.relu_(y + b.reshape(1, -1, 1, 1))
        pooled = maxpool(y)
    return pooled


class Bottleneck(nn.Module):
    """ResNet v1.5 bottleneck: 1x1 reduce, 3x3 (carries the stride), 1x1 expand (x4), identity/projection."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1,
                 norm_layer=FrozenBatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self._folds = [_FoldCache() for _ in range(4)]

    def forward(self, x):
        fold = _fold_mode(self, x)
        out = _conv_bn(x, self.conv1, self.bn1, self._folds[0], True, fold)
        out = _conv_bn(out, self.conv2, self.bn2, self._folds[1], True, fold)
        if self.downsample is not None:
            x = _conv_bn(x, self.downsample[0], self.downsample[1], self._folds[3], False, fold)
        return _conv_bn(out, self.conv3, self.bn3, self._folds[2], True, fold, residual=x)


class ResNet(nn.Module):
    """torchvision-compatible ResNet trunk (conv1, bn1, relu, maxpool, layer1..4, avgpool, fc)."""

    def __init__(self, layers, num_classes=1000, replace_stride_with_dilation=None,
                 norm_layer=FrozenBatchNorm2d):
        super().__init__()
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError("replace_stride_with_dilation should be None or a 3-element tuple")
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.dilation = 1
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2, replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(256, layers[2], 2, replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(512, layers[3], 2, replace_stride_with_dilation[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * Bottleneck.expansion, num_classes)
        self._stem_fold = _FoldCache()
        for m in self.modules():  # torchvision's default initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        norm_layer = self._norm_layer
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, kernel_size=1, stride=stride,
                          bias=False),
                norm_layer(planes * Bottleneck.expansion))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, previous_dilation, norm_layer)]
        self.inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self.dilation,
                                     norm_layer=norm_layer))
        return nn.Sequential(*layers)

    def stem(self, x):
        fold = _fold_mode(self.conv1, x) if self.training else int(_inference_mode(self))
        if fold == 1:
            pooled = _stem_pooled(x, self.conv1, self.bn1, self.maxpool, self._stem_fold)
            if pooled is not None:
                return pooled
        x = _conv_bn(x, self.conv1, self.bn1, self._stem_fold, True, fold)
        return self.maxpool(x)

    def forward(self, x):
        x = self.stem(x)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


_RESNET_DEPTHS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}


def resnet(name: str, replace_stride_with_dilation=None, norm_layer=FrozenBatchNorm2d) -> ResNet:
    if name not in _RESNET_DEPTHS:
        raise ValueError("unsupported backbone %r (have %s)" % (name, sorted(_RESNET_DEPTHS)))
    return ResNet(_RESNET_DEPTHS[name], replace_stride_with_dilation=replace_stride_with_dilation,
                  norm_layer=norm_layer)


class IntermediateLayerGetter(nn.ModuleDict):
    """Runs a model's children in order and returns the outputs of the named ones (torchvision's
    models._utils.IntermediateLayerGetter): children after the last requested layer are dropped, so
    `avgpool`/`fc` do not appear in the state dict."""

    def __init__(self, model: nn.Module, return_layers: Dict[str, str]):
        if not set(return_layers).issubset(name for name, _ in model.named_children()):
            raise ValueError("return_layers are not present in model")
        remaining = dict(return_layers)
        kept = OrderedDict()
        for name, module in model.named_children():
            kept[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(kept)
        self.return_layers = dict(return_layers)
        self._fold = _FoldCache()

    def forward(self, x):
        out = OrderedDict()
        # (the stem: the mode of ITS convolution -- frozen by the reference, so the inference kernels in a training step too)
        fold = _fold_mode(self["conv1"], x) if (self.training and "conv1" in self) else int(_inference_mode(self))
        fused_stem = fold == 1 and all(k in self for k in ("conv1", "bn1", "relu"))
        # opt-in: the stem's shift + ReLU + pooling in one pass (only when nobody asked for the intermediate maps)
        pooled_stem = (fused_stem and "maxpool" in self
                       and not any(k in self.return_layers for k in ("conv1", "bn1", "relu", "maxpool")))
        skip = ()
        for name, module in self.items():
            if name in skip or (fused_stem and name in ("bn1", "relu")):
                continue
            if fused_stem and name == "conv1":
                pooled = _stem_pooled(x, self["conv1"], self["bn1"], self["maxpool"], self._fold) if pooled_stem else None
                if pooled is not None:
                    x, skip = pooled, ("maxpool",)
                    continue
                x = _conv_bn(x, self["conv1"], self["bn1"], self._fold, True, 1)
            else:
                x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


class BackboneBase(nn.Module):
    def __init__(self, backbone: nn.Module, train_backbone: bool, return_interm_layers: bool):
        super().__init__()
        for name, parameter in backbone.named_parameters():
            if not train_backbone or not any(k in name for k in ("layer2", "layer3", "layer4")):
                parameter.requires_grad_(False)
        if return_interm_layers:
            return_layers = {"layer1": "0", "layer2": "1", "layer3": "2", "layer4": "3"}
            self.strides = [4, 8, 16, 32]
            self.num_channels = [256, 512, 1024, 2048]
        else:
            return_layers = {"layer4": "0"}
            self.strides = [32]
            self.num_channels = [2048]
        self.body = IntermediateLayerGetter(backbone, return_layers=return_layers)

    def forward(self, tensor_list: NestedTensor):
        x = tensor_list.tensors
        if (_inference_mode(self) or (_train_fold and self.training)) and x.is_cuda and CHANNELS_LAST:
            x = x.contiguous(memory_format=torch.channels_last)
        xs = self.body(x)
        m = tensor_list.mask
        assert m is not None
        out: Dict[str, NestedTensor] = {}
        for name, feat in xs.items():
            if is_all_valid(m):  # nearest-downsampling an all-False mask: nothing to compute
                mask = all_valid_mask((m.shape[0],) + tuple(feat.shape[-2:]), feat.device)
            else:
                mask = F.interpolate(m[None].float(), size=feat.shape[-2:]).to(torch.bool)[0]
            out[name] = NestedTensor(feat, mask)
        return out


class Backbone(BackboneBase):
    """ResNet backbone with frozen BatchNorm.  No pretrained download is attempted (there is no
    network; the reference's `pretrained=is_main_process()` fetch, backbone.py:100, is out of scope):
    weights come from a checkpoint or stay at their seeded initialisation."""

    def __init__(self, name: str, train_backbone: bool, return_interm_layers: bool, dilation: bool):
        backbone = resnet(name, replace_stride_with_dilation=[False, False, dilation],
                          norm_layer=FrozenBatchNorm2d)
        super().__init__(backbone, train_backbone, return_interm_layers)
        if dilation:
            self.strides[-1] = self.strides[-1] // 2


class Joiner(nn.Sequential):
    """[backbone, position_embedding] -> (list of NestedTensor features, list of position encodings)."""

    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides = backbone.strides
        self.num_channels = backbone.num_channels

    def forward(self, tensor_list: NestedTensor):
        xs = self[0](tensor_list)
        out: List[NestedTensor] = []
        pos = []
        for x in xs.values():
            out.append(x)
            pos.append(self[1](x).to(x.tensors.dtype))
        return out, pos


def build_backbone(args):
    position_embedding = build_position_encoding(args)
    train_backbone = args.lr_backbone > 0
    return_interm_layers = args.masks or (args.num_feature_levels > 1)
    backbone = Backbone(args.backbone, train_backbone, return_interm_layers, args.dilation)
    return Joiner(backbone, position_embedding)
