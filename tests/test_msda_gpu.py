"""GPU (-m gpu): parity of the HIP MSDeformAttn path, called through the C ABI, against
(a) the committed golden vectors from the reference's ms_deform_attn_core_pytorch,
(b) the C oracle on seeded inputs (small to full cfg-2 / cfg-4 sizes),
(c) size-independent properties at BASELINE.json's full sizes,
plus the reference's own checks restated (ops/test.py: forward/backward equality, gradcheck).

Tolerances: fp32 within 1e-5 abs (+1e-4 rel) of the oracle -- far inside the 1e-3 that north_star
states for boxes/logits and the rtol=1e-2/atol=1e-3 of ops/test.py:30; fp64 within 1e-12.
"""
import numpy as np
import pytest
import torch

from oracle import msda_oracle
from tests.util_msda import (CFG2_SHAPES, discontinuity_mask, golden_cases, load_case,
                             rand_inputs)

pytestmark = pytest.mark.gpu

CASES = golden_cases()


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (torch.cuda.is_available() is False)")
    from trackformer_amd import _cabi
    _cabi.lib()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def _tol(dtype):
    return (1e-12, 1e-10) if dtype in (np.float64, torch.float64) else (1e-5, 1e-4)


def _fwd(value, shapes, loc, attn, host_shapes=True):
    from trackformer_amd import msda
    if host_shapes and msda._host_shapes_of(shapes) is None:
        msda.attach_host_shapes(shapes, shapes.tolist())
    return msda.ms_deform_attn_forward(value, shapes, loc, attn, 64)


def _bwd(value, shapes, loc, attn, grad_out, host_shapes=True):
    from trackformer_amd import msda
    if host_shapes and msda._host_shapes_of(shapes) is None:
        msda.attach_host_shapes(shapes, shapes.tolist())
    return msda.ms_deform_attn_backward(value, shapes, loc, attn, grad_out, 64)


def _to_dev(z, dev):
    t = {k: torch.from_numpy(v).to(dev) for k, v in z.items()}
    return t["value"], t["shapes"], t["loc"], t["attn"], t["grad_out"]


# ------------------------------------------------------------------ (a) golden vectors
@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("msda_")[-1][:-4])
@pytest.mark.parametrize("host_shapes", [True, False], ids=["hostshapes", "devshapes"])
def test_forward_golden(dev, path, host_shapes):
    z = load_case(path)
    value, shapes, loc, attn, _ = _to_dev(z, dev)
    out = _fwd(value, shapes, loc, attn, host_shapes).cpu().numpy()
    atol, rtol = _tol(z["value"].dtype)
    np.testing.assert_allclose(out, z["out"], atol=atol, rtol=rtol)


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("msda_")[-1][:-4])
@pytest.mark.parametrize("host_shapes", [True, False], ids=["hostshapes", "devshapes"])
def test_backward_golden(dev, path, host_shapes):
    z = load_case(path)
    value, shapes, loc, attn, grad_out = _to_dev(z, dev)
    gv, gl, ga = [t.cpu().numpy() for t in _bwd(value, shapes, loc, attn, grad_out, host_shapes)]
    atol, rtol = _tol(z["value"].dtype)
    np.testing.assert_allclose(gv, z["grad_value"], atol=atol * 2, rtol=rtol)
    np.testing.assert_allclose(ga, z["grad_attn"], atol=atol * 10, rtol=rtol)
    keep = ~discontinuity_mask(z["loc"], z["shapes"])
    np.testing.assert_allclose(gl[keep], z["grad_loc"][keep], atol=atol * 10, rtol=rtol)
    assert np.all(gl[~keep] == 0)  # CUDA semantics at the boundary (cuh:359-362)


# ------------------------------------------------------------------ (b) oracle, seeded inputs
ORACLE_CASES = [
    # id, kwargs
    ("tiny", dict(N=2, M=2, D=4, Lq=3, P=2, shapes=[(8, 8), (4, 4), (2, 2)])),
    ("ragged_levels", dict(N=1, M=3, D=8, Lq=65, P=3, shapes=[(1, 1), (1, 7), (9, 1), (3, 5)],
                           loc_mode="wide")),
    ("one_level_one_point", dict(N=2, M=1, D=4, Lq=1, P=1, shapes=[(5, 6)])),
    ("d5_scalar_path", dict(N=2, M=3, D=5, Lq=33, P=2, shapes=[(7, 3), (2, 2)], loc_mode="wide")),
    ("d36_l8", dict(N=2, M=8, D=36, Lq=150, P=4, shapes=[(13, 21), (7, 11), (4, 6), (2, 3)] * 2,
                    loc_mode="wide")),
    ("d32_l4_mid", dict(N=2, M=8, D=32, Lq=777, P=4, shapes=[(25, 42), (13, 21), (7, 11), (4, 6)],
                        loc_mode="local")),
    ("d64", dict(N=1, M=4, D=64, Lq=100, P=4, shapes=[(12, 10), (6, 5)], loc_mode="wide")),
    ("d128_max_lanes", dict(N=1, M=2, D=128, Lq=40, P=2, shapes=[(6, 6)], loc_mode="rand")),
    ("levels16", dict(N=1, M=2, D=8, Lq=20, P=1, shapes=[(3, 2)] * 16, loc_mode="wide")),
]


@pytest.mark.parametrize("name,kw", ORACLE_CASES, ids=[c[0] for c in ORACLE_CASES])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_forward_backward_vs_oracle(dev, name, kw, dtype):
    value, shapes, loc, attn, grad_out = rand_inputs(seed=100 + len(name), dtype=dtype, **kw)
    ref_out = msda_oracle.msda_forward(value.numpy(), shapes.numpy(), loc.numpy(), attn.numpy())
    ref_gv, ref_gl, ref_ga = msda_oracle.msda_backward(value.numpy(), shapes.numpy(), loc.numpy(),
                                                       attn.numpy(), grad_out.numpy())
    d = [t.to(dev) for t in (value, shapes, loc, attn, grad_out)]
    out = _fwd(*d[:4]).cpu().numpy()
    gv, gl, ga = [t.cpu().numpy() for t in _bwd(*d)]
    atol, rtol = _tol(dtype)
    np.testing.assert_allclose(out, ref_out, atol=atol, rtol=rtol)
    np.testing.assert_allclose(gv, ref_gv, atol=atol * 4, rtol=rtol)
    np.testing.assert_allclose(gl, ref_gl, atol=atol * 20, rtol=rtol)
    np.testing.assert_allclose(ga, ref_ga, atol=atol * 10, rtol=rtol)


def test_empty_neighbourhood_all_samples_out_of_range(dev):
    value, shapes, loc, attn, grad_out = rand_inputs(7, N=1, M=2, D=8, Lq=9, P=2,
                                                     shapes=[(4, 4), (2, 2)], device=dev)
    loc = loc + 5.0  # every sample far outside -> output and all gradients are exactly zero
    assert torch.count_nonzero(_fwd(value, shapes, loc, attn)) == 0
    gv, gl, ga = _bwd(value, shapes, loc, attn, grad_out)
    assert torch.count_nonzero(gv) == 0 and torch.count_nonzero(gl) == 0 \
        and torch.count_nonzero(ga) == 0


def test_nonfinite_locations_do_not_fault(dev):
    value, shapes, loc, attn, _ = rand_inputs(8, N=1, M=2, D=8, Lq=9, P=2,
                                              shapes=[(4, 4), (2, 2)], device=dev)
    loc[0, 0, 0, 0, 0, 0] = float("inf")
    loc[0, 1, 0, 0, 0, 1] = float("-inf")
    loc[0, 2, 0, 0, 0, 0] = 1e30
    out = _fwd(value, shapes, loc, attn)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


def test_zero_padding_is_exact_with_nonfinite_neighbours(dev):
    # an Inf stored in a border pixel must not leak (as 0*Inf) into samples whose tap there is invalid
    shapes = torch.tensor([[2, 2]], device=dev)
    value = torch.ones(1, 4, 1, 4, device=dev)
    value[0, 0] = float("inf")                       # pixel (0,0)
    loc = torch.tensor([1.9, 1.9], device=dev).view(1, 1, 1, 1, 1, 2)  # out of range sample
    attn = torch.ones(1, 1, 1, 1, 1, device=dev)
    assert torch.equal(_fwd(value, shapes, loc, attn), torch.zeros(1, 1, 4, device=dev))
    loc2 = torch.tensor([0.75, 0.9], device=dev).view(1, 1, 1, 1, 1, 2)  # taps (1,1),(1,2)x: only in-image (1,1)
    out2 = _fwd(value, shapes, loc2, attn)
    assert torch.isfinite(out2).all()


# ------------------------------------------------------------------ (c) full BASELINE sizes
def _cfg2(dev, Lq, seed=0, loc_mode="rand", N=1):
    return rand_inputs(seed, N=N, M=8, D=32, Lq=Lq, P=4, shapes=CFG2_SHAPES, loc_mode=loc_mode,
                       device=dev)


@pytest.mark.parametrize("Lq,loc_mode", [(22223, "rand"), (22223, "local"), (400, "wide")],
                         ids=["encoder_rand", "encoder_local", "decoder_wide"])
def test_full_size_cfg2_vs_oracle(dev, Lq, loc_mode):
    value, shapes, loc, attn, grad_out = _cfg2(dev, Lq, seed=5, loc_mode=loc_mode)
    out = _fwd(value, shapes, loc, attn).cpu().numpy()
    ref = msda_oracle.msda_forward(value.cpu().numpy(), shapes.cpu().numpy(), loc.cpu().numpy(),
                                   attn.cpu().numpy(), nthreads=8)
    np.testing.assert_allclose(out, ref, atol=1e-5, rtol=1e-4)


def test_full_size_cfg2_backward_vs_oracle(dev):
    value, shapes, loc, attn, grad_out = _cfg2(dev, 22223, seed=6, loc_mode="local", N=2)
    gv, gl, ga = [t.cpu().numpy() for t in _bwd(value, shapes, loc, attn, grad_out)]
    rv, rl, ra = msda_oracle.msda_backward(value.cpu().numpy(), shapes.cpu().numpy(),
                                           loc.cpu().numpy(), attn.cpu().numpy(),
                                           grad_out.cpu().numpy())
    # grad_value accumulates ~64 contributions per element in nondeterministic order
    np.testing.assert_allclose(gv, rv, atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(gl, rl, atol=2e-3, rtol=1e-4)  # values are O(100) (scaled by W_l, H_l)
    np.testing.assert_allclose(ga, ra, atol=1e-4, rtol=1e-4)


def test_full_size_cfg4_multiframe_decoder_vs_oracle(dev):
    # cfg 4: hidden 288 (D=36), 8 decoder levels (2 frames), Lq = 500 + 300
    value, shapes, loc, attn, _ = rand_inputs(9, N=1, M=8, D=36, Lq=800, P=4,
                                              shapes=CFG2_SHAPES * 2, loc_mode="wide", device=dev)
    out = _fwd(value, shapes, loc, attn).cpu().numpy()
    ref = msda_oracle.msda_forward(value.cpu().numpy(), shapes.cpu().numpy(), loc.cpu().numpy(),
                                   attn.cpu().numpy(), nthreads=8)
    np.testing.assert_allclose(out, ref, atol=1e-5, rtol=1e-4)


def test_full_size_properties(dev):
    """Linearity in value and in attn, permutation equivariance over queries, partition of unity."""
    value, shapes, loc, attn, _ = _cfg2(dev, 22223, seed=3, loc_mode="local")
    out = _fwd(value, shapes, loc, attn)
    # linearity in value / attn (exact up to fp32 rounding of the scaling)
    out2 = _fwd(value * 2.0, shapes, loc, attn)
    assert torch.allclose(out2, out * 2.0, atol=1e-6, rtol=1e-6)
    out3 = _fwd(value, shapes, loc, attn * 0.5)
    assert torch.allclose(out3, out * 0.5, atol=1e-6, rtol=1e-6)
    # query permutation equivariance (the tiled encoder kernel sums staged and global taps in a
    # different order when a query is not at its pyramid position, hence not bitwise)
    perm = torch.randperm(loc.shape[1], device=dev)
    outp = _fwd(value, shapes, loc[:, perm].contiguous(), attn[:, perm].contiguous())
    assert torch.allclose(outp, out[:, perm], atol=2e-6, rtol=1e-5)
    # decoder shape (row-gather kernel): bitwise, each query is computed independently in one order
    sub = perm[:400]
    d1 = _fwd(value, shapes, loc[:, sub].contiguous(), attn[:, sub].contiguous())
    d2 = _fwd(value, shapes, loc[:, sub.flip(0)].contiguous(), attn[:, sub.flip(0)].contiguous())
    assert torch.equal(d1, d2.flip(1))
    # constant value + in-range samples + weights summing to 1 -> output == constant
    ones = torch.ones_like(value)
    loc_in = loc.clamp(0.2, 0.8)
    outc = _fwd(ones, shapes, loc_in, attn)
    assert torch.allclose(outc, torch.ones_like(outc), atol=1e-5)
    # forward is deterministic run to run
    assert torch.equal(_fwd(value, shapes, loc, attn), out)


def test_device_shapes_entry_point_matches_host_shapes(dev):
    value, shapes, loc, attn, grad_out = _cfg2(dev, 400, seed=4, loc_mode="wide")
    a = _fwd(value, shapes, loc, attn, host_shapes=True)
    fresh = shapes.clone()  # no host attribute -> *_dshapes kernel path (shapes read on device)
    b = _fwd(value, fresh, loc, attn, host_shapes=False)
    assert torch.equal(a, b)
    ga = _bwd(value, shapes, loc, attn, grad_out, host_shapes=True)
    gb = _bwd(value, shapes.clone(), loc, attn, grad_out, host_shapes=False)
    assert torch.equal(ga[1], gb[1]) and torch.equal(ga[2], gb[2])
    assert torch.allclose(ga[0], gb[0], atol=1e-5)


# ------------------------------------------------------------------ reference's own checks, restated
def test_reference_check_forward_equal_test_py(dev):
    """ops/test.py:23-35 with its generator and tolerance, oracle in place of the torch path."""
    torch.manual_seed(3)
    N, M, D, Lq, L, P = 2, 2, 4, 3, 3, 2
    shapes = torch.as_tensor([(8, 8), (4, 4), (2, 2)], dtype=torch.long, device=dev)
    S = 84
    value = torch.rand(N, S, M, D, device=dev) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2, device=dev)
    attn = torch.rand(N, Lq, M, L, P, device=dev) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    from trackformer_amd.msda import MSDeformAttnFunction
    out = MSDeformAttnFunction.apply(value, shapes, loc, attn, 2)
    ref = msda_oracle.msda_forward(value.cpu().numpy(), shapes.cpu().numpy(), loc.cpu().numpy(),
                                   attn.cpu().numpy())
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-2, atol=1e-3)
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-8


@pytest.mark.parametrize("gv,gl,ga", [(True, False, False), (False, True, False),
                                      (False, False, True), (True, True, True)])
def test_reference_gradcheck_double(dev, gv, gl, ga):
    """ops/test_double_precision.py:97-119: gradcheck with default tolerances in fp64."""
    from trackformer_amd.msda import MSDeformAttnFunction
    torch.manual_seed(3)
    N, M, D, Lq, L, P = 2, 2, 4, 3, 3, 2
    shapes = torch.as_tensor([(12, 8), (6, 4), (3, 2)], dtype=torch.long, device=dev)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = (torch.rand(N, S, M, D, device=dev) * 0.01).double()
    loc = torch.rand(N, Lq, M, L, P, 2, device=dev).double()
    attn = torch.rand(N, Lq, M, L, P, device=dev).double() + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    value.requires_grad = gv
    loc.requires_grad = gl
    attn.requires_grad = ga
    assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, loc, attn, 2),
                                    nondet_tol=1e-12)


def test_autograd_backward_with_noncontiguous_grad_output(dev):
    from trackformer_amd.msda import MSDeformAttnFunction
    value, shapes, loc, attn, grad_out = rand_inputs(12, N=2, M=4, D=8, Lq=19, P=2,
                                                     shapes=[(5, 4), (3, 2)], device=dev)
    value.requires_grad_(True)
    loc.requires_grad_(True)
    attn.requires_grad_(True)
    out = MSDeformAttnFunction.apply(value, shapes, loc, attn, 64)
    # a transposed (strided) upstream gradient, as autograd may hand over
    (out.transpose(1, 2) * grad_out.transpose(1, 2)).sum().backward()
    rv, rl, ra = msda_oracle.msda_backward(value.detach().cpu().numpy(), shapes.cpu().numpy(),
                                           loc.detach().cpu().numpy(), attn.detach().cpu().numpy(),
                                           grad_out.cpu().numpy())
    np.testing.assert_allclose(value.grad.cpu().numpy(), rv, atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(loc.grad.cpu().numpy(), rl, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(attn.grad.cpu().numpy(), ra, atol=1e-4, rtol=1e-4)


def test_error_contract_on_gpu(dev):
    from trackformer_amd import msda
    value, shapes, loc, attn, _ = rand_inputs(1, N=3, M=2, D=4, Lq=5, P=2, shapes=[(4, 4)],
                                              device=dev)
    with pytest.raises(RuntimeError, match="contiguous"):
        msda.ms_deform_attn_forward(value.transpose(2, 3).contiguous().transpose(2, 3), shapes,
                                    loc, attn, 64)
    with pytest.raises(RuntimeError, match="im2col_step"):
        msda.ms_deform_attn_forward(value, shapes, loc, attn, 2)  # 3 % 2 != 0 (cu:46-48)
    bad = msda.attach_host_shapes(shapes.clone(), [(4, 5)])
    with pytest.raises(RuntimeError, match="does not equal S"):
        msda.ms_deform_attn_forward(value, bad, loc, attn, 64)
    with pytest.raises(RuntimeError, match="float32 and float64"):
        msda.ms_deform_attn_forward(value.half(), shapes, loc.half(), attn.half(), 64)


def test_runs_on_the_current_stream_without_sync(dev):
    value, shapes, loc, attn, _ = _cfg2(dev, 400, seed=2)
    ref = _fwd(value, shapes, loc, attn)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = _fwd(value, shapes, loc, attn)
    s.synchronize()
    assert torch.equal(out, ref)


def test_hip_graph_capture(dev):
    value, shapes, loc, attn, _ = _cfg2(dev, 400, seed=2)
    ref = _fwd(value, shapes, loc, attn)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _fwd(value, shapes, loc, attn)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = _fwd(value, shapes, loc, attn)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_dropin_module_name_resolves_to_hip_path(dev):
    import trackformer_amd.dropin as dropin
    MSDA = dropin.install()
    import MultiScaleDeformableAttention as again
    assert again is MSDA
    value, shapes, loc, attn, grad_out = rand_inputs(2, N=2, M=2, D=4, Lq=3, P=2,
                                                     shapes=[(8, 8), (4, 4), (2, 2)], device=dev)
    out = MSDA.ms_deform_attn_forward(value, shapes, loc, attn, 2)
    grads = MSDA.ms_deform_attn_backward(value, shapes, loc, attn, grad_out, 2)
    assert out.shape == (2, 3, 8) and len(grads) == 3
    ref = msda_oracle.msda_forward(value.cpu().numpy(), shapes.cpu().numpy(), loc.cpu().numpy(),
                                   attn.cpu().numpy())
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=1e-6)


def test_module_forward_matches_oracle_composition(dev):
    """MSDeformAttn module on GPU vs the same module arithmetic with the oracle as the operator."""
    from trackformer_amd import msda
    torch.manual_seed(0)
    m = msda.MSDeformAttn(256, 4, 8, 4).to(dev)
    with torch.no_grad():  # move away from the degenerate init so offsets/weights depend on the query
        m.sampling_offsets.weight.normal_(0, 0.02)
        m.attention_weights.weight.normal_(0, 0.1)
    shapes_l = [(12, 20), (6, 10), (3, 5), (2, 3)]
    S = sum(h * w for h, w in shapes_l)
    shapes = msda.attach_host_shapes(torch.tensor(shapes_l, device=dev), shapes_l)
    q = torch.randn(2, 50, 256, device=dev)
    src = torch.randn(2, S, 256, device=dev)
    ref2 = torch.rand(2, 50, 4, 2, device=dev)
    ref4 = torch.rand(2, 50, 4, 4, device=dev) * 0.5 + 0.1
    mask = torch.zeros(2, S, dtype=torch.bool, device=dev)
    mask[1, -7:] = True
    for ref_pts in (ref2, ref4):
        out = m(q, ref_pts, src, shapes, mask)
        mc = msda.MSDeformAttn(256, 4, 8, 4)
        mc.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
        OracleFn = msda_oracle.make_torch_function()
        orig = msda.MSDeformAttnFunction
        msda.MSDeformAttnFunction = OracleFn
        try:
            exp = mc(q.cpu(), ref_pts.cpu(), src.cpu(), torch.tensor(shapes_l), mask.cpu())
        finally:
            msda.MSDeformAttnFunction = orig
        assert torch.allclose(out.cpu(), exp, atol=2e-4, rtol=1e-3)


@pytest.fixture(autouse=True)
def _fp32_linears():
    """This module tests the MSDeformAttn kernels and their fused prologue against fp32 compositions: the
    split-product linears (default on in the inference path, tested in test_linear_split_gpu.py and
    test_full_size_gpu.py) are switched off so that the projections are the same fp32 GEMMs on both sides."""
    from trackformer_amd import fused
    prev = fused.set_split_linear(False)
    yield
    fused.set_split_linear(prev)


# ------------------------------------------------------------------ tiled / LDS-staged encoder kernel
@pytest.fixture(params=[2, 3, 4], ids=["quad", "pquad", "pquad_v1"])
def tiled(dev, request):
    """Select an LDS-window encoder kernel (2: msda_fwd_f32_quad, 3: the persistent kernel -- the default for encoder-shaped
    calls: msda_fwd_f32_pquad2 where it applies, 4: its first version msda_fwd_f32_pquad everywhere) for the duration of a test."""
    from trackformer_amd import _cabi
    lib = _cabi.lib()
    prev = lib.tf_msda_set_tiled(2)
    prev_pq = lib.tf_msda_set_option(b"pquad", 0 if request.param == 2 else 1)
    prev_v2 = lib.tf_msda_set_option(b"pquad_v2", 0 if request.param == 4 else 1)
    yield
    lib.tf_msda_set_tiled(prev)
    lib.tf_msda_set_option(b"pquad", prev_pq)
    lib.tf_msda_set_option(b"pquad_v2", prev_v2)


def _encoder_inputs(dev, shapes, mode, N=1, M=8, D=32, seed=0):
    from tools.bench_msda import make_inputs
    S = sum(h * w for h, w in shapes)
    return make_inputs(N, M, D, S, 4, shapes, mode, dev, seed=seed, encoder_refs=True)


TILED_CASES = [
    ("cfg2_init", CFG2_SHAPES, "init", 1, 8, 32),
    ("cfg2_local_n2", CFG2_SHAPES, "local", 2, 8, 32),
    ("cfg2_uniform_all_fallback", CFG2_SHAPES, "uniform", 1, 8, 32),
    ("mot17_750x1333", [(94, 167), (47, 84), (24, 42), (12, 21)], "local", 1, 8, 32),
    ("small_pyramid", [(25, 42), (13, 21), (7, 11), (4, 6)], "init", 2, 8, 32),
    ("tiny_levels", [(3, 5), (2, 3), (1, 2), (1, 1)], "local", 1, 8, 32),
    ("one_level", [(37, 53)], "local", 1, 8, 32),
    ("coarse_first_falls_back", [(13, 21), (100, 167)], "local", 1, 8, 32),
    ("cfg4_d36_hidden288", CFG2_SHAPES, "local", 1, 8, 36),
    ("d36_small_n2", [(30, 44), (15, 22), (8, 11)], "init", 2, 8, 36),
    ("d64_m4", [(40, 60), (20, 30)], "local", 1, 4, 64),
    ("d16_m8", [(40, 60), (20, 30)], "init", 1, 8, 16),
]


@pytest.mark.parametrize("name,shapes,mode,N,M,D", TILED_CASES, ids=[c[0] for c in TILED_CASES])
def test_encoder_shape_tiled_kernel_vs_oracle(dev, tiled, name, shapes, mode, N, M, D):
    """Lq == S selects the tiled kernel (when the plan fits); any input must still be exact."""
    value, shp, loc, attn, _ = _encoder_inputs(dev, shapes, mode, N=N, M=M, D=D, seed=len(name))
    out = _fwd(value, shp, loc, attn).cpu().numpy()
    ref = msda_oracle.msda_forward(value.cpu().numpy(), shp.cpu().numpy(), loc.cpu().numpy(),
                                   attn.cpu().numpy(), nthreads=8)
    np.testing.assert_allclose(out, ref, atol=1e-5, rtol=1e-4)


def test_tiled_kernel_with_shuffled_queries_and_mixed_windows(dev, tiled):
    """Queries that are NOT at their pyramid position (a permutation) and samples that straddle the
    window border: the heuristics are wrong for every tile, results must not change."""
    value, shp, loc, attn, _ = _encoder_inputs(dev, CFG2_SHAPES, "local", seed=3)
    base = _fwd(value, shp, loc, attn)
    perm = torch.randperm(loc.shape[1], device=dev)
    out = _fwd(value, shp, loc[:, perm].contiguous(), attn[:, perm].contiguous())
    assert torch.allclose(out, base[:, perm], atol=1e-6, rtol=1e-6)
    # push half of the points of every query ~12 px away: in-window and out-of-window taps mix
    far = loc.clone()
    far[:, :, :, :, ::2, 0] += 12.0 / 167
    far[:, :, :, :, ::2, 1] -= 9.0 / 100
    out = _fwd(value, shp, far, attn).cpu().numpy()
    ref = msda_oracle.msda_forward(value.cpu().numpy(), shp.cpu().numpy(), far.cpu().numpy(),
                                   attn.cpu().numpy(), nthreads=8)
    np.testing.assert_allclose(out, ref, atol=1e-5, rtol=1e-4)


def test_tiled_kernel_matches_rowgather_kernel(dev, tiled):
    """Same inputs through the decoder-style kernel (Lq != S forces it) give the same numbers."""
    value, shp, loc, attn, _ = _encoder_inputs(dev, CFG2_SHAPES, "init", seed=5)
    tiled = _fwd(value, shp, loc, attn)
    # drop the last query -> Lq = S - 1 -> row-gather kernel
    rg = _fwd(value, shp, loc[:, :-1].contiguous(), attn[:, :-1].contiguous())
    assert torch.allclose(tiled[:, :-1], rg, atol=2e-6, rtol=1e-5)


PQUAD_VARIANTS = [dict(pquad_npass=1, pquad_wg_per_cu=4), dict(pquad_npass=3, pquad_wg_per_cu=2),
                  dict(pquad_prefetch=2, pquad_wg_per_cu=2), dict(pquad_wide=0), dict(pquad_lds_kb=24),
                  dict(pquad_tile_h=4, pquad_tile_w=16, pquad_wg_per_cu=1), dict(pquad_skew=100),
                  # round 5: msda_fwd_f32_pquad2 (msda_pquad2.h) is the default where it applies (D == 32, two passes, 16-byte
                  # loads) -- the first version at the default plan, and the second under other plans
                  dict(pquad_v2=0), dict(pquad_v2=0, pquad_lds_kb=24), dict(pquad_wg_per_cu=1), dict(pquad_halo_y=2, pquad_halo_x=2),
                  dict(pquad_lds_kb=12), dict(pquad_skew=150, pquad_wg_per_cu=2),
                  # eight-wave workgroups of version 2: one pass of 128 pairs, two workgroups per CU
                  dict(pquad_waves=8, pquad_npass=1, pquad_wg_per_cu=2, pquad_lds_kb=78), dict(pquad_waves=8, pquad_npass=1, pquad_wg_per_cu=1, pquad_lds_kb=30),
                  # round 6: the conflict-free gather (+ its buffer-load path under small windows, + eight waves), the output-store
                  # policies (nt is the default), non-temporal point loads, the rotated head maps, static priorities
                  dict(pquad_cf=1), dict(pquad_cf=1, pquad_lds_kb=12), dict(pquad_cf=1, pquad_waves=8, pquad_npass=1, pquad_wg_per_cu=2, pquad_lds_kb=78),
                  dict(pquad_store=0, pquad_headmix=1, pquad_ldnt=1), dict(pquad_store=2, pquad_headmix=2, pquad_prio=2), dict(pquad_store=3, pquad_prio=1)]


@pytest.mark.parametrize("opts", PQUAD_VARIANTS, ids=["-".join("%s%d" % (k[6:], v) for k, v in o.items())
                                                      for o in PQUAD_VARIANTS])
def test_persistent_encoder_kernel_variants_vs_oracle(dev, opts):
    """msda_fwd_f32_pquad with its knobs away from the defaults (passes per tile, register prefetch, narrow
    loads, a small LDS budget, few workgroups = many tiles per workgroup, start-up skew), plain and fused entry,
    N = 2: the tile loop, the double-buffered per-tile tables and the prefetch must not change a result."""
    from trackformer_amd import _cabi, msda
    lib = _cabi.lib()
    prev = {k: lib.tf_msda_set_option(k.encode(), v) for k, v in opts.items()}
    try:
        shapes_l = [(40, 61), (20, 31), (10, 16), (5, 8)]
        value, shp, loc, attn, _ = _encoder_inputs(dev, shapes_l, "local", N=2, seed=11)
        out = _fwd(value, shp, loc, attn).cpu().numpy()
        ref = msda_oracle.msda_forward(value.cpu().numpy(), shp.cpu().numpy(), loc.cpu().numpy(),
                                       attn.cpu().numpy(), nthreads=8)
        np.testing.assert_allclose(out, ref, atol=1e-5, rtol=1e-4)
        # fused entry on the same geometry: raw offsets / logits + encoder reference points
        N, S, M, D = value.shape
        L, P = len(shapes_l), 4
        g = torch.Generator().manual_seed(5)
        qproj = torch.randn(N, S, 3 * M * L * P, generator=g)
        qproj[..., :2 * M * L * P] *= 2.0
        refp = loc.cpu()[:, :, 0, :, 0, :].contiguous() * 0 + torch.rand(N, S, L, 2, generator=g) * 0.8 + 0.1
        off = qproj[..., :2 * M * L * P].view(N, S, M, L, P, 2)
        a = torch.softmax(qproj[..., 2 * M * L * P:].view(N, S, M, L * P), -1).view(N, S, M, L, P)
        hw = torch.tensor(shapes_l, dtype=torch.float32)[None, None, None, :, None, :]
        floc = refp[:, :, None, :, None, :] + off / hw
        expect = msda_oracle.msda_forward(value.cpu().numpy(), shp.cpu().numpy(), floc.numpy(), a.numpy(), nthreads=8)
        got = msda.ms_deform_attn_forward_fused(value, shp, refp.to(dev), qproj.to(dev), M, L, P)
        np.testing.assert_allclose(got.cpu().numpy(), expect.reshape(got.shape), atol=2e-5, rtol=1e-4)
    finally:
        for k, v in prev.items():
            lib.tf_msda_set_option(k.encode(), v)


@pytest.mark.parametrize("shapes_l,N", [(CFG2_SHAPES, 1), ([(30, 44), (15, 22), (8, 11)], 2), ([(37, 53)], 1)],
                         ids=["cfg4_encoder", "three_levels_n2", "one_level"])
def test_persistent_encoder_kernel_head_dim_36(dev, shapes_l, N):
    """hidden 288 (cfg 4: `multi_frame`, train_multi_frame.yaml:2): head dimension 36 through msda_fwd_f32_pquad's
    144-byte-row variant (3 lanes x 12 channels, windows staged in 16-byte pieces), plain and fused entry."""
    from trackformer_amd import _cabi, msda
    lib = _cabi.lib()
    prev_t = lib.tf_msda_set_tiled(2)
    prev_p = lib.tf_msda_set_option(b"pquad", 1)
    try:
        value, shp, loc, attn, _ = _encoder_inputs(dev, shapes_l, "local", N=N, M=8, D=36, seed=7)
        out = _fwd(value, shp, loc, attn).cpu().numpy()
        ref = msda_oracle.msda_forward(value.cpu().numpy(), shp.cpu().numpy(), loc.cpu().numpy(),
                                       attn.cpu().numpy(), nthreads=8)
        np.testing.assert_allclose(out, ref, atol=1e-5, rtol=1e-4)
        Nn, S, M, D = value.shape
        L, P = len(shapes_l), 4
        g = torch.Generator().manual_seed(9)
        qproj = torch.randn(Nn, S, 3 * M * L * P, generator=g)
        qproj[..., :2 * M * L * P] *= 2.0
        refp = torch.rand(Nn, S, L, 2, generator=g) * 0.8 + 0.1
        off = qproj[..., :2 * M * L * P].view(Nn, S, M, L, P, 2)
        a = torch.softmax(qproj[..., 2 * M * L * P:].view(Nn, S, M, L * P), -1).view(Nn, S, M, L, P)
        hw = torch.tensor(shapes_l, dtype=torch.float32)[None, None, None, :, None, :]
        floc = refp[:, :, None, :, None, :] + off / hw
        expect = msda_oracle.msda_forward(value.cpu().numpy(), shp.cpu().numpy(), floc.numpy(), a.numpy(), nthreads=8)
        got = msda.ms_deform_attn_forward_fused(value, shp, refp.to(dev), qproj.to(dev), M, L, P)
        np.testing.assert_allclose(got.cpu().numpy(), expect.reshape(got.shape), atol=5e-5, rtol=1e-4)
    finally:
        lib.tf_msda_set_tiled(prev_t)
        lib.tf_msda_set_option(b"pquad", prev_p)


BWD_ENC_CASES = [c for c in TILED_CASES if c[0] in (
    "cfg2_init", "cfg2_uniform_all_fallback", "mot17_750x1333", "small_pyramid", "tiny_levels",
    "one_level", "coarse_first_falls_back", "cfg4_d36_hidden288")]


@pytest.mark.parametrize("name,shapes,mode,N,M,D", BWD_ENC_CASES, ids=[c[0] for c in BWD_ENC_CASES])
def test_encoder_shape_backward_vs_oracle(dev, name, shapes, mode, N, M, D):
    """Backward at encoder shapes (Lq == S) incl. degenerate pyramids and far-away sampling points:
    D == 32 runs the full-row atomic scatter (msda_bwd_f32_buf<P, true>), D == 36 the 32-byte one."""
    value, shp, loc, attn, grad_out = _encoder_inputs(dev, shapes, mode, N=N, M=M, D=D, seed=len(name))
    if name == "cfg2_init":   # mix in points far outside their windows
        loc = loc.clone()
        loc[:, :, :, :, ::2, 0] += 12.0 / 167
        loc[:, :, :, :, ::2, 1] -= 9.0 / 100
    gv, gl, ga = [t.cpu().numpy() for t in _bwd(value, shp, loc, attn, grad_out)]
    rv, rl, ra = msda_oracle.msda_backward(value.cpu().numpy(), shp.cpu().numpy(), loc.cpu().numpy(),
                                           attn.cpu().numpy(), grad_out.cpu().numpy())
    np.testing.assert_allclose(gv, rv, atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(gl, rl, atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(ga, ra, atol=1e-4, rtol=1e-4)


# ------------------------------------------------------------------ fused prologue (inference)
@pytest.mark.parametrize("ref_dim", [2, 4])
@pytest.mark.parametrize("Lq,shapes_l,d_model", [(400, CFG2_SHAPES, 256), (1020, [(24, 32), (12, 16), (6, 8), (3, 4)], 256),
                                                 (65, [(13, 21), (7, 11), (4, 6), (2, 3)] * 2, 288)])
def test_fused_prologue_matches_unfused_module(dev, ref_dim, Lq, shapes_l, d_model):
    """MSDeformAttn inference fast path (cat-projection GEMM + fused kernel prologue) vs the reference
    formulation (separate Linears, softmax, location arithmetic, plain operator) on the same module."""
    from trackformer_amd import msda
    torch.manual_seed(1)
    L = len(shapes_l)
    m = msda.MSDeformAttn(d_model, L, 8, 4).to(dev).eval()
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.05)
        m.attention_weights.weight.normal_(0, 0.3)
        m.attention_weights.bias.normal_(0, 0.3)
    S = sum(h * w for h, w in shapes_l)
    shapes = msda.attach_host_shapes(torch.tensor(shapes_l, device=dev), shapes_l)
    q = torch.randn(2, Lq, d_model, device=dev)
    src = torch.randn(2, S, d_model, device=dev)
    ref = torch.rand(2, Lq, L, ref_dim, device=dev) * (0.5 if ref_dim == 4 else 1.0) + 0.05
    with torch.no_grad():
        assert msda.FUSED_INFERENCE
        fused_out = m(q, ref, src, shapes, None)
        msda.FUSED_INFERENCE = False
        try:
            plain_out = m(q, ref, src, shapes, None)
        finally:
            msda.FUSED_INFERENCE = True
    assert torch.allclose(fused_out, plain_out, atol=3e-5, rtol=1e-4)


def test_fused_prologue_entry_point_vs_oracle(dev):
    """C-ABI tf_msda_forward_fused_f32 against oracle(softmax / loc arithmetic done in torch fp32)."""
    from trackformer_amd import msda
    g = torch.Generator().manual_seed(7)
    N, M, D, L, P, Lq = 1, 8, 32, 4, 4, 333
    shapes_l = [(25, 42), (13, 21), (7, 11), (4, 6)]
    S = sum(h * w for h, w in shapes_l)
    value = torch.randn(N, S, M, D, generator=g)
    qproj = torch.randn(N, Lq, 3 * M * L * P, generator=g)
    shapes = torch.tensor(shapes_l)
    for ref_dim in (2, 4):
        ref = torch.rand(N, Lq, L, ref_dim, generator=g) * 0.6 + 0.1
        off = qproj[..., :2 * M * L * P].view(N, Lq, M, L, P, 2)
        attn = torch.softmax(qproj[..., 2 * M * L * P:].view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
        if ref_dim == 2:
            loc = ref[:, :, None, :, None, :] + off / shapes[None, None, None, :, None, :]
        else:
            loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
        expect = msda_oracle.msda_forward(value.numpy(), shapes.numpy(), loc.numpy(), attn.numpy())
        dshapes = msda.attach_host_shapes(shapes.to(dev), shapes_l)
        out = msda.ms_deform_attn_forward_fused(value.to(dev), dshapes, ref.to(dev), qproj.to(dev),
                                                M, L, P)
        np.testing.assert_allclose(out.cpu().numpy(), expect, atol=2e-5, rtol=1e-4)


# ------------------------------------------------------------------ kernels promoted to defaults in round 3
# Written and checked in the SIMT emulator (tests/test_emu_kernels.py) while no GPU was available, confirmed and timed on
# MI355X at the start of round 3 (profiles/r03_optin_pytest_optin.txt, r03_optin_msda_variants.txt).


@pytest.mark.parametrize("Lq,L,N", [(800, 8, 1), (70, 8, 2), (29, 4, 2), (5, 3, 1)], ids=["cfg4_decoder", "l8_n2", "l4_n2", "l3_tiny"])
def test_direct9_decoder_kernel(dev, Lq, L, N):
    """msda_fwd_f32_direct9 (D = 36, 9 lanes per pair): plain + fused entry vs the oracle, and vs msda_fwd_f32_buf."""
    from trackformer_amd import _cabi, msda
    lib = _cabi.lib()
    shapes_l = (CFG2_SHAPES * 2)[:L] if Lq == 800 else ([(13, 21), (7, 11), (4, 6), (2, 3)] * 2)[:L]
    value, shapes, loc, attn, _ = rand_inputs(40 + Lq, N=N, M=8, D=36, Lq=Lq, P=4, shapes=shapes_l, loc_mode="wide", device=dev)
    ref = msda_oracle.msda_forward(value.cpu().numpy(), shapes.cpu().numpy(), loc.cpu().numpy(), attn.cpu().numpy(), nthreads=8)
    prev = lib.tf_msda_set_option(b"direct9", 0)   # msda_fwd_f32_buf, the kernel direct9 replaced
    try:
        base = _fwd(value, shapes, loc, attn)
    finally:
        lib.tf_msda_set_option(b"direct9", prev)
    prev = lib.tf_msda_set_option(b"direct9", 1)
    try:
        out = _fwd(value, shapes, loc, attn)
        np.testing.assert_allclose(out.cpu().numpy(), ref, atol=1e-5, rtol=1e-4)
        assert torch.allclose(out, base, atol=2e-6, rtol=1e-5)
        assert torch.equal(_fwd(value, shapes.clone(), loc, attn, host_shapes=False), out)
        g = torch.Generator().manual_seed(Lq)
        M, P = 8, 4
        qproj = torch.randn(N, Lq, 3 * M * L * P, generator=g)
        for ref_dim in (2, 4):
            refp = torch.rand(N, Lq, L, ref_dim, generator=g) * 0.6 + 0.1
            off = qproj[..., :2 * M * L * P].view(N, Lq, M, L, P, 2)
            a = torch.softmax(qproj[..., 2 * M * L * P:].view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
            if ref_dim == 2:
                floc = refp[:, :, None, :, None, :] + off / torch.tensor(shapes_l, dtype=torch.float32)[None, None, None, :, None, :]
            else:
                floc = refp[:, :, None, :, None, :2] + off / P * refp[:, :, None, :, None, 2:] * 0.5
            expect = msda_oracle.msda_forward(value.cpu().numpy(), shapes.cpu().numpy(), floc.numpy(), a.numpy(), nthreads=8)
            got = msda.ms_deform_attn_forward_fused(value, shapes, refp.to(dev), qproj.to(dev), M, L, P)
            np.testing.assert_allclose(got.cpu().numpy(), expect, atol=2e-5, rtol=1e-4)
    finally:
        lib.tf_msda_set_option(b"direct9", prev)


@pytest.mark.parametrize("name,shapes,mode,N,M,D", [c for c in BWD_ENC_CASES if c[5] == 32], ids=[c[0] for c in BWD_ENC_CASES if c[5] == 32])
def test_backward_sorted_kernel_with_points_outside_their_windows(dev, name, shapes, mode, N, M, D):
    """msda_bwd_f32_sorted2 at the encoder shapes of test_encoder_shape_backward_vs_oracle, with half of the points of the
    cfg-2 case pushed far outside their binned windows (the direct-scatter path)."""
    value, shp, loc, attn, grad_out = _encoder_inputs(dev, shapes, mode, N=N, M=M, D=D, seed=len(name))
    if name == "cfg2_init":
        loc = loc.clone()
        loc[:, :, :, :, ::2, 0] += 12.0 / 167
        loc[:, :, :, :, ::2, 1] -= 9.0 / 100
    rv, rl, ra = msda_oracle.msda_backward(value.cpu().numpy(), shp.cpu().numpy(), loc.cpu().numpy(),
                                           attn.cpu().numpy(), grad_out.cpu().numpy())
    gv, gl, ga = [t.cpu().numpy() for t in _bwd(value, shp, loc, attn, grad_out)]
    np.testing.assert_allclose(gv, rv, atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(gl, rl, atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(ga, ra, atol=1e-4, rtol=1e-4)

