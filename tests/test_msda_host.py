"""CPU: host-side behaviour of the operator mirror (error contract, module surface, init)."""
import numpy as np
import pytest
import torch

from tests.util_msda import discontinuity_mask, golden_cases, load_case
from trackformer_amd import msda


def _tiny(dtype=torch.float32):
    value = torch.zeros(1, 4, 1, 4, dtype=dtype)
    shapes = torch.tensor([[2, 2]])
    loc = torch.zeros(1, 1, 1, 1, 1, 2, dtype=dtype)
    attn = torch.zeros(1, 1, 1, 1, 1, dtype=dtype)
    return value, shapes, loc, attn


@pytest.mark.parametrize("path", golden_cases(), ids=lambda p: p.split("msda_")[-1][:-4])
def test_host_operator_matches_reference_goldens(path):
    """tf_msda_{forward,backward}_host_* (csrc/msda_host.cpp; SURVEY 8(b): a real CPU path where the reference raises "Not
    implemented on the CPU", ms_deform_attn.h:27,48) against the vectors of the reference's own ms_deform_attn_core_pytorch
    incl. its autograd gradients (tests/golden/make_golden_msda.py) -- the same bar the oracle is pinned with."""
    z = load_case(path)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in z.items() if k != "shapes"}
    shapes = torch.from_numpy(z["shapes"].astype(np.int64))
    atol, rtol = (1e-12, 1e-10) if z["value"].dtype == np.float64 else (2e-5, 1e-4)
    out = msda.ms_deform_attn_forward(t["value"], shapes, t["loc"], t["attn"], 64)
    np.testing.assert_allclose(out.numpy(), z["out"], atol=atol, rtol=rtol)
    gv, gl, ga = msda.ms_deform_attn_backward(t["value"], shapes, t["loc"], t["attn"], t["grad_out"], 64)
    np.testing.assert_allclose(gv.numpy(), z["grad_value"], atol=atol, rtol=rtol)
    np.testing.assert_allclose(ga.numpy(), z["grad_attn"], atol=atol * 10, rtol=rtol)
    keep = ~discontinuity_mask(z["loc"], z["shapes"])
    np.testing.assert_allclose(gl.numpy()[keep], z["grad_loc"][keep], atol=atol * 10, rtol=rtol)
    assert np.all(gl.numpy()[~keep] == 0)   # cuh:229: a point exactly on the in-range boundary contributes nothing
    # the backward is deterministic (work split by batch x head, no atomics)
    gv2, _, _ = msda.ms_deform_attn_backward(t["value"], shapes, t["loc"], t["attn"], t["grad_out"], 64)
    assert torch.equal(gv, gv2)


def test_host_operator_through_autograd_and_gradcheck():
    """MSDeformAttnFunction on CPU tensors: forward + backward through the host entry points; fp64 gradcheck as the
    reference's test.py:62-83 does on the GPU."""
    g = torch.Generator().manual_seed(3)
    shapes = torch.tensor([[3, 4], [2, 2]])
    N, M, D, Lq, L, P = 1, 2, 4, 3, 2, 2
    S = 16
    value = (torch.rand(N, S, M, D, generator=g, dtype=torch.float64) * 0.01).requires_grad_()
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=torch.float64).requires_grad_()
    attn = torch.rand(N, Lq, M, L, P, generator=g, dtype=torch.float64) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).requires_grad_()
    assert torch.autograd.gradcheck(lambda v, l, a: msda.MSDeformAttnFunction.apply(v, shapes, l, a, 2), (value, loc, attn))


def test_host_operator_error_contract():
    value, shapes, loc, attn = _tiny()
    with pytest.raises(RuntimeError, match="contiguous"):                       # cu:29
        msda.ms_deform_attn_forward(torch.zeros(1, 8, 1, 4)[:, ::2], shapes, loc, attn, 64)
    with pytest.raises(RuntimeError, match="dtype"):
        msda.ms_deform_attn_forward(value, shapes, loc.double(), attn, 64)
    bad = shapes.clone()
    bad[0, 0] = 3                                                              # sum(H * W) != S
    with pytest.raises(RuntimeError):
        msda.ms_deform_attn_forward(value, bad, loc, attn, 64)
    with pytest.raises(RuntimeError, match="float32 and float64"):
        msda.ms_deform_attn_forward(value.half(), shapes, loc.half(), attn.half(), 64)
    out = msda.ms_deform_attn_forward(value, shapes, loc, attn, 64)
    assert out.shape == (1, 1, 4) and not out.is_cuda


def test_device_tensors_never_take_the_host_entry_points(monkeypatch):
    """The host path is selected by `value.is_cuda` alone, before any pointer is taken: a tensor that reports a device goes
    to the tf_msda_forward_f32 family (no fallback when that fails)."""
    calls = []

    class _Lib:
        def __getattr__(self, name):
            def f(*a):
                calls.append(name)
                return 0
            return f
    monkeypatch.setattr(msda._cabi, "lib", lambda: _Lib())
    value, shapes, loc, attn = _tiny()
    msda.ms_deform_attn_forward(value, shapes, loc, attn, 64)
    assert calls == ["tf_msda_forward_host_f32"]


def test_module_state_dict_keys_and_shapes():
    m = msda.MSDeformAttn(d_model=256, n_levels=4, n_heads=8, n_points=4)
    sd = m.state_dict()
    assert list(sd.keys()) == [
        "sampling_offsets.weight", "sampling_offsets.bias", "attention_weights.weight",
        "attention_weights.bias", "value_proj.weight", "value_proj.bias", "output_proj.weight",
        "output_proj.bias"]
    assert sd["sampling_offsets.weight"].shape == (256, 256)
    assert sd["attention_weights.weight"].shape == (128, 256)
    m8 = msda.MSDeformAttn(d_model=288, n_levels=8, n_heads=8, n_points=4)
    assert m8.sampling_offsets.weight.shape == (512, 288)
    assert m8.attention_weights.weight.shape == (256, 288)


def test_module_init_matches_reference_scheme():
    m = msda.MSDeformAttn(d_model=256, n_levels=4, n_heads=8, n_points=4)
    assert torch.count_nonzero(m.sampling_offsets.weight) == 0
    assert torch.count_nonzero(m.attention_weights.weight) == 0
    assert torch.count_nonzero(m.attention_weights.bias) == 0
    assert torch.count_nonzero(m.value_proj.bias) == 0
    b = m.sampling_offsets.bias.view(8, 4, 4, 2)
    # literal table of modules/ms_deform_attn.py:36 scaled by (point index + 1)
    table = torch.tensor([-1, -1, -1, 0, -1, 1, 0, -1, 0, 1, 1, -1, 1, 0, 1, 1.]).view(8, 2)
    for p in range(4):
        for lvl in range(4):
            assert torch.equal(b[:, lvl, p], table * (p + 1))
    with pytest.raises(ValueError):
        msda.MSDeformAttn(d_model=256, n_heads=4)
    with pytest.raises(ValueError):
        msda.MSDeformAttn(d_model=250, n_heads=8)


def test_host_shape_attachment():
    s = torch.tensor([[2, 3], [1, 1]])
    assert msda._host_shapes_of(s) == ((2, 3), (1, 1))
    msda.attach_host_shapes(s, [(5, 6), (7, 8)])
    assert msda._host_shapes_of(s) == ((5, 6), (7, 8))
