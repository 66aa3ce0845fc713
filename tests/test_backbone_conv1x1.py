"""CPU: the mapping of a stride-1 1 x 1 convolution on channels_last activations onto a GEMM over the pixels
(trackformer_amd/backbone.py: conv1x1_as_gemm, the opt-in split-product route of the bottleneck convolutions), with a
plain torch matmul standing in for the GPU kernel: shapes, views, residual and ReLU handling against F.conv2d."""
import torch
import torch.nn.functional as F

from trackformer_amd import backbone


def _torch_linear(x2, w2d, bias, relu, residual2):
    assert x2.is_contiguous() and x2.dim() == 2
    y = x2 @ w2d.t()
    if bias is not None:
        y = y + bias
    if residual2 is not None:
        assert residual2.is_contiguous() and residual2.shape == y.shape
        y = y + residual2
    return torch.relu(y) if relu else y


def test_conv1x1_as_gemm_matches_conv2d():
    g = torch.Generator().manual_seed(0)
    for (n, cin, cout, h, w) in [(2, 64, 256, 7, 9), (1, 256, 64, 5, 4), (3, 32, 32, 1, 1)]:
        x = torch.randn(n, cin, h, w, generator=g).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(cout, cin, 1, 1, generator=g)
        b = torch.randn(cout, generator=g)
        res = torch.randn(n, cout, h, w, generator=g).contiguous(memory_format=torch.channels_last)
        for residual in (None, res):
            for relu in (False, True):
                ref = F.conv2d(x, wt, b)
                if residual is not None:
                    ref = ref + residual
                if relu:
                    ref = torch.relu(ref)
                y = backbone.conv1x1_as_gemm(x, wt.reshape(cout, cin), b, residual, relu, _torch_linear)
                assert y is not None and y.shape == ref.shape
                assert y.is_contiguous(memory_format=torch.channels_last)   # stays NHWC for the next convolution
                assert torch.allclose(y, ref, atol=1e-4, rtol=1e-5)


def test_conv1x1_as_gemm_declines_what_it_cannot_map():
    x = torch.randn(1, 8, 4, 4)                       # NCHW storage: not a pixel-major matrix
    assert backbone.conv1x1_as_gemm(x, torch.randn(4, 8), None, None, False, _torch_linear) is None
    xc = x.contiguous(memory_format=torch.channels_last)
    bad_res = torch.randn(1, 4, 4, 4)                 # residual in the other layout
    assert backbone.conv1x1_as_gemm(xc, torch.randn(4, 8), None, bad_res, False, _torch_linear) is None
    assert backbone.conv1x1_as_gemm(xc, torch.randn(4, 8), None, None, False, lambda *a: None) is None


def test_fold_cache_keeps_a_persistent_2d_weight_for_1x1_convolutions():
    conv = torch.nn.Conv2d(8, 16, 1, bias=False)
    bn = backbone.FrozenBatchNorm2d(16)
    bn.weight.uniform_(0.5, 1.5)
    cache = backbone._FoldCache()
    b = cache.get(conv, bn)
    assert cache._weight is None and cache._weight2d is None      # nothing is materialised before a route asks for it
    w2 = cache.weight2d
    assert w2 is not None and w2.shape == (16, 8) and w2.is_contiguous()
    assert cache._weight is None                       # the split route never pays for the 4-d image
    assert torch.equal(w2, cache.weight.reshape(16, 8))
    cache.get(conv, bn)
    assert cache.weight2d is w2                        # same object: the split pieces cached on it stay valid
    with torch.no_grad():
        bn.weight.mul_(2.0)                            # a source tensor changed: every image is rebuilt
    cache.get(conv, bn)
    assert cache.weight2d is not w2 and torch.allclose(cache.weight2d, 2.0 * w2)
    conv3 = torch.nn.Conv2d(8, 16, 3, bias=False)
    c3 = backbone._FoldCache()
    c3.get(conv3, bn)
    assert c3.weight2d is None


def test_fold_cache_tap_major_weight_of_3x3_convolutions():
    """weight_taps[o, (kh * 3 + kw) * Cin + c] == folded weight[o, c, kh, kw]: the K order tf_conv3x3_split_f32 expects."""
    conv = torch.nn.Conv2d(8, 16, 3, padding=1, bias=False)
    bn = backbone.FrozenBatchNorm2d(16)
    bn.weight.uniform_(0.5, 1.5)
    cache = backbone._FoldCache()
    b = cache.get(conv, bn)
    w = cache.weight
    t = cache.weight_taps
    assert t is not None and t.shape == (16, 72) and t.is_contiguous() and cache.weight2d is None
    assert torch.equal(t.view(16, 3, 3, 8), w.permute(0, 2, 3, 1))
    # and the implicit GEMM over those taps is the convolution (stride 1 and 2, padding 1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8, 7, 9, generator=g)
    for stride in (1, 2):
        ref = F.conv2d(x, w, b, stride=stride, padding=1)
        cols = F.unfold(x, 3, padding=1, stride=stride)                      # [N, Cin * 9, L] with (c, kh, kw) order
        cols = cols.view(2, 8, 9, -1).permute(0, 3, 2, 1).reshape(2, -1, 72)  # -> (tap, c) order
        y = (cols @ t.t() + b).permute(0, 2, 1).reshape(ref.shape)
        assert torch.allclose(y, ref, atol=1e-5)


def test_per_shape_skip_list_of_the_split_routes():
    """TF_CONV_SPLIT_SKIP / set_conv_split_skip: listed (cin, cout, kernel, stride) shapes keep the library convolution."""
    import torch
    from torch import nn
    from trackformer_amd import backbone
    assert backbone._parse_skip("64x64x3x1, 256x64x1x1") == frozenset({(64, 64, 3, 1), (256, 64, 1, 1)})
    assert backbone._parse_skip("") == frozenset()
    import pytest
    with pytest.raises(ValueError):
        backbone._parse_skip("64x64x3")
    prev = backbone.set_conv_split_skip([(64, 64, 3, 1)])
    try:
        assert not backbone._split_route_allowed(nn.Conv2d(64, 64, 3, stride=1, padding=1))
        assert backbone._split_route_allowed(nn.Conv2d(64, 64, 3, stride=2, padding=1))
        assert backbone._split_route_allowed(nn.Conv2d(64, 256, 1))
    finally:
        backbone.set_conv_split_skip(prev)
    # the CPU path is untouched by any of it
    conv, bn = nn.Conv2d(8, 8, 3, padding=1, bias=False), backbone.FrozenBatchNorm2d(8)
    x = torch.randn(1, 8, 5, 5)
    with torch.no_grad():
        y = backbone._conv_bn(x, conv, bn, backbone._FoldCache(), True, True)
    assert torch.allclose(y, torch.relu(bn(conv(x))), atol=1e-6)
