"""GPU (-m gpu): the DEVICE branch of the compiled drop-in module (trackformer_amd/dropin/csrc/msda_ext.cpp:
`MultiScaleDeformableAttention` as a pybind11 torch extension over the C ABI, the form the reference ships its plugin in --
models/ops/src/vision.cpp:4-7, models/ops/setup.py:30-66).  tests/test_dropin_compiled.py covers its host branch on the CPU; here
the same module on device tensors: the reference goldens, device-resident spatial_shapes (the reference's calling convention ->
the *_dshapes entry points, no host synchronisation), a non-default stream, HIP-graph capture, the reference's own autograd
Function on top, and its error behaviour."""
import glob
import os

import numpy as np
import pytest
import torch

from tests.util_msda import discontinuity_mask
from trackformer_amd import dropin, msda

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ext(dev):
    mod = dropin.install(compiled=True)     # the in-tree .so (built by __graft_entry__.build(); rebuilt only if its sources changed)
    assert mod.__file__.endswith(".so") and os.sep + "compiled" + os.sep in mod.__file__
    with open("/proc/self/maps") as f:       # the driver's "native code loaded" record looks at exactly this
        assert any("MultiScaleDeformableAttention" in line and "libtf_msda" not in line for line in f)
    yield mod
    dropin.install()


def _case(dev, dtype=torch.float32, N=2, Lq=300, seed=0, shapes=((25, 42), (13, 21), (7, 11), (4, 6)), M=8, D=32, P=4):
    g = torch.Generator().manual_seed(seed)
    shp = torch.tensor(shapes, dtype=torch.long)
    S, L = int((shp[:, 0] * shp[:, 1]).sum()), len(shapes)
    value = torch.randn(N, S, M, D, generator=g, dtype=dtype)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=dtype) * 1.2 - 0.1
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g, dtype=dtype), -1).view(N, Lq, M, L, P)
    go = torch.randn(N, Lq, M * D, generator=g, dtype=dtype)
    return value.to(dev), shp, loc.to(dev), attn.to(dev), go.to(dev)


@pytest.mark.parametrize("device_shapes", [False, True], ids=["host_shapes", "device_shapes"])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "msda_*.npz"))), ids=os.path.basename)
def test_compiled_module_on_device_reproduces_the_reference_goldens(ext, dev, path, device_shapes):
    z = np.load(path)
    t = lambda k: torch.from_numpy(z[k])
    value, loc, attn = t("value").to(dev), t("loc").to(dev), t("attn").to(dev)
    shapes = t("shapes").long().to(dev) if device_shapes else t("shapes").long()
    f32 = value.dtype == torch.float32
    out = ext.ms_deform_attn_forward(value, shapes, loc, attn, 64)
    assert out.is_cuda and out.dtype == value.dtype
    want = z["out"].reshape(out.shape)
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=(1e-5 if f32 else 1e-12) * max(1.0, float(np.abs(want).max())))
    go = t("grad_out").reshape(out.shape).to(dev)
    gv, gl, ga = [g.cpu().numpy() for g in ext.ms_deform_attn_backward(value, shapes, loc, attn, go, 64)]
    atol, rtol = (1e-5, 1e-4) if f32 else (1e-12, 1e-10)          # the bars of tests/test_msda_gpu.py::test_backward_golden
    np.testing.assert_allclose(gv, z["grad_value"].reshape(gv.shape), atol=atol * 2, rtol=rtol)
    np.testing.assert_allclose(ga, z["grad_attn"].reshape(ga.shape), atol=atol * 10, rtol=rtol)
    keep = ~discontinuity_mask(z["loc"], z["shapes"])             # sampling points ON a pixel boundary: the one-sided derivative is a convention
    np.testing.assert_allclose(gl[keep], z["grad_loc"].reshape(gl.shape)[keep], atol=atol * 10, rtol=rtol)
    assert np.all(gl[~keep] == 0)                                 # CUDA semantics at the boundary (cuh:359-362)
    # the same library entry points as the ctypes form of the module
    assert torch.equal(out, msda.ms_deform_attn_forward(value, shapes, loc, attn, 64))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_device_resident_shapes_take_the_dshapes_entry_without_a_sync(ext, dev, dtype):
    """spatial_shapes on the device (how the reference calls its plugin): the *_dshapes entry points read the levels inside the
    kernel -- same results as with host shapes, and no host synchronisation (sync debug mode raises on any)."""
    value, shp, loc, attn, go = _case(dev, dtype)
    want = ext.ms_deform_attn_forward(value, shp, loc, attn, 64)
    want_g = ext.ms_deform_attn_backward(value, shp, loc, attn, go, 64)
    dshp = shp.to(dev)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = ext.ms_deform_attn_forward(value, dshp, loc, attn, 64)
        grads = ext.ms_deform_attn_backward(value, dshp, loc, attn, go, 64)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    assert float((out - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    for g, w in zip(grads, want_g):
        assert float((g - w).abs().max()) <= (3e-4 if dtype == torch.float32 else 1e-10) * max(1.0, float(w.abs().max()))


def test_compiled_module_runs_on_the_callers_stream(ext, dev):
    """The extension takes torch's CURRENT stream (c10::hip::getCurrentHIPStreamMasqueradingAsCUDA): work enqueued on a side
    stream behind a long kernel is ordered after it, and is not visible to the default stream before the side stream is waited for."""
    value, shp, loc, attn, go = _case(dev, N=1, Lq=22223 // 8)
    want = ext.ms_deform_attn_forward(value, shp, loc, attn, 64)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        v2 = value * 2.0                       # produced ON the side stream: a default-stream launch would race with it
        out = ext.ms_deform_attn_forward(v2, shp, loc, attn, 64)
        done = torch.cuda.Event()
        done.record(side)
    done.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), 2.0 * want.cpu().numpy(), atol=2e-5 * float(want.abs().max()))


def test_compiled_module_is_capturable_in_a_hip_graph(ext, dev):
    """No allocation outside torch's allocator, no synchronisation, the capturing stream: forward + backward replay from a HIP
    graph with new inputs in the static buffers."""
    value, shp, loc, attn, go = _case(dev, N=1, Lq=400)
    dshp = shp.to(dev)
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                 # warm-up on the side stream (first-use plans / caches are built here)
        ext.ms_deform_attn_forward(value, dshp, loc, attn, 64)
        ext.ms_deform_attn_backward(value, dshp, loc, attn, go, 64)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ext.ms_deform_attn_forward(value, dshp, loc, attn, 64)
        gv, gl, ga = ext.ms_deform_attn_backward(value, dshp, loc, attn, go, 64)
    v2, _, l2, a2, g2 = _case(dev, N=1, Lq=400, seed=7)
    value.copy_(v2); loc.copy_(l2); attn.copy_(a2); go.copy_(g2)
    graph.replay()
    torch.cuda.synchronize()
    want = msda.ms_deform_attn_forward(v2, shp, l2, a2, 64)
    wg = msda.ms_deform_attn_backward(v2, shp, l2, a2, g2, 64)
    assert float((out - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    for g, w in zip((gv, gl, ga), wg):
        assert float((g - w).abs().max()) <= 3e-4 * max(1.0, float(w.abs().max()))


def test_autograd_function_over_the_compiled_module(ext, dev):
    """The shape of the reference's MSDeformAttnFunction (functions/ms_deform_attn_func.py:21-41: forward saves the tensors,
    backward calls ms_deform_attn_backward) over the compiled module, in float64, against torch.autograd.gradcheck -- the check
    the reference's own ops/test.py:49-71 runs on its CUDA build."""
    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, value, shapes, loc, attn, step):
            ctx.step = step
            ctx.save_for_backward(value, shapes, loc, attn)
            return ext.ms_deform_attn_forward(value, shapes, loc, attn, step)

        @staticmethod
        def backward(ctx, grad_output):
            value, shapes, loc, attn = ctx.saved_tensors
            gv, gl, ga = ext.ms_deform_attn_backward(value, shapes, loc, attn, grad_output.contiguous(), ctx.step)
            return gv, None, gl, ga, None
    value, shp, loc, attn, _ = _case(dev, torch.float64, N=1, Lq=2, shapes=((6, 4), (3, 2)), M=2, D=4, P=2)
    loc = loc.clamp(0.05, 0.95)
    value.requires_grad_(True); loc.requires_grad_(True); attn.requires_grad_(True)
    assert torch.autograd.gradcheck(lambda v, l, a: Fn.apply(v, shp.to(dev), l, a, 2), (value, loc, attn), eps=1e-6, atol=1e-5, rtol=1e-3)


def test_compiled_module_keeps_the_reference_checks_on_device(ext, dev):
    value, shp, loc, attn, go = _case(dev, N=3, Lq=5, shapes=((4, 5),), M=2, D=4, P=2)
    assert ext.ms_deform_attn_forward(value, shp, loc, attn).shape == (3, 5, 8)
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        ext.ms_deform_attn_forward(value, shp, loc, attn, 2)                                  # cu:37-39
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.ms_deform_attn_forward(value.transpose(1, 2), shp, loc, attn, 64)                 # cu:26-29
    with pytest.raises(RuntimeError, match="one device"):
        ext.ms_deform_attn_forward(value, shp, loc.cpu(), attn, 64)                           # cu:31-34 (AT_ASSERTM ... must be a CUDA tensor)
    with pytest.raises(RuntimeError):                                                          # sum H W != S: the library's own check
        ext.ms_deform_attn_forward(value, torch.tensor([[4, 4]]), loc, attn, 64)
    out = ext.ms_deform_attn_forward(value, shp, loc, attn, 64)                               # ... and the module still works afterwards
    assert torch.isfinite(out).all()
