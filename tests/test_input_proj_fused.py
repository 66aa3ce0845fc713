"""CPU: the opt-in fused `input_proj` route (trackformer_amd/fused.py: input_proj_1x1 = split GEMM over the pixels + the
library's channels-innermost GroupNorm) -- its host-side mapping, with torch stand-ins for the two GPU kernels, against
the nn.Sequential(Conv2d, GroupNorm) it replaces (reference: models/deformable_detr.py:73-90)."""
import torch

from trackformer_amd import fused


class _FakeCuda(torch.Tensor):
    pass


def test_input_proj_mapping_matches_sequential(monkeypatch):
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(48, 64, 1)
    gn = torch.nn.GroupNorm(8, 64)
    with torch.no_grad():
        gn.weight.normal_(1, 0.2)
        gn.bias.normal_(0, 0.2)
    x = torch.randn(2, 48, 5, 7).contiguous(memory_format=torch.channels_last)
    ref = gn(conv(x))

    def lin(x2, w2d, bias=None, relu=False, rows=None, residual=None):
        assert x2.shape == (2 * 5 * 7, 48) and x2.is_contiguous() and w2d.shape == (64, 48)
        return x2 @ w2d.t() + bias

    def gnorm(x2, n_img, g):
        hw = x2.shape[0] // n_img
        xr = x2.view(n_img, hw, g.num_groups, -1)
        mean = xr.mean(dim=(1, 3), keepdim=True)
        var = xr.var(dim=(1, 3), unbiased=False, keepdim=True)
        return (((xr - mean) / torch.sqrt(var + g.eps)).reshape(n_img * hw, -1) * g.weight + g.bias)

    monkeypatch.setattr(fused, "linear", lin)
    monkeypatch.setattr(fused, "groupnorm_nhwc", gnorm)
    monkeypatch.setattr(fused, "_input_proj_fused", True)
    # the route is for GPU tensors; exercise the mapping on the host by lifting the device check
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    with torch.no_grad():
        got = fused.input_proj_1x1(x, conv, gn)
    assert got is not None and got.shape == ref.shape
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(got, ref, atol=1e-5)
    # the 2-d weight view is cached on the module
    assert conv._tf_w2d[1].shape == (64, 48)


def test_input_proj_route_is_off_by_default_and_declines_other_convolutions(monkeypatch):
    conv = torch.nn.Conv2d(8, 8, 3, stride=2, padding=1)
    gn = torch.nn.GroupNorm(2, 8)
    x = torch.randn(1, 8, 4, 4)
    assert fused.input_proj_1x1(x, torch.nn.Conv2d(8, 8, 1), gn) is None          # switched off / CPU tensor
    monkeypatch.setattr(fused, "_input_proj_fused", True)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    assert fused.input_proj_1x1(x, conv, gn) is None                                # 3 x 3 stride 2: stays in the library
