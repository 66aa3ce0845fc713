"""CPU: the COMPILED form of the drop-in module (trackformer_amd/dropin/csrc/msda_ext.cpp: `MultiScaleDeformableAttention` as a
pybind11 torch extension over the C ABI of libtf_msda.so -- what the reference ships as models/ops/src/vision.cpp:4-7 + setup.py) on
host tensors: the reference's two functions with its signatures, against the oracle's goldens and the ctypes binding, and the
reference's error behaviour (a RuntimeError for what ms_deform_attn_cuda.cu:26-39 asserts).  The device branch is exercised by the
same module on the GPU box (tests/test_dropin_compiled_gpu.py); nothing in the package routes through this form."""
import glob
import os

import numpy as np
import pytest
import torch

from trackformer_amd import dropin, msda

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ext():
    import shutil
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    mod = dropin.install(compiled=True)
    assert mod.__file__.endswith(".so") and os.sep + "compiled" + os.sep in mod.__file__
    yield mod
    dropin.install()          # leave the Python form importable under the name, as before


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "msda_*.npz"))), ids=os.path.basename)
def test_compiled_module_reproduces_the_reference_goldens(ext, path):
    z = np.load(path)
    t = lambda k: torch.from_numpy(z[k])
    value, shapes, loc, attn = t("value"), t("shapes").long(), t("loc"), t("attn")
    out = ext.ms_deform_attn_forward(value, shapes, loc, attn, 64)
    tol = 1e-5 if value.dtype == torch.float32 else 1e-12
    want_out = z["out"].reshape(out.shape)
    np.testing.assert_allclose(out.numpy(), want_out, atol=tol * max(1.0, float(np.abs(want_out).max())))
    assert torch.equal(out, msda.ms_deform_attn_forward(value, shapes, loc, attn, 64))          # same library call: same bits
    go = t("grad_out").reshape(out.shape)
    grads = ext.ms_deform_attn_backward(value, shapes, loc, attn, go, 64)
    want = msda.ms_deform_attn_backward(value, shapes, loc, attn, go, 64)
    np.testing.assert_allclose(grads[0].numpy(), z["grad_value"].reshape(value.shape), atol=10 * tol * max(1.0, float(np.abs(z["grad_value"]).max())))
    assert isinstance(grads, list) and len(grads) == 3 and all(torch.equal(a, b) for a, b in zip(grads, want))
    assert [tuple(g.shape) for g in grads] == [tuple(value.shape), tuple(loc.shape), tuple(attn.shape)]


def test_compiled_module_keeps_the_reference_checks(ext):
    N, S, M, D, Lq, L, P = 3, 20, 2, 4, 5, 1, 2
    value, shapes = torch.randn(N, S, M, D), torch.tensor([[4, 5]])
    loc, attn = torch.rand(N, Lq, M, L, P, 2), torch.rand(N, Lq, M, L, P)
    assert ext.ms_deform_attn_forward(value, shapes, loc, attn).shape == (N, Lq, M * D)          # im2col_step defaults to 64
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        ext.ms_deform_attn_forward(value, shapes, loc, attn, 2)                                  # cu:37-39: batch % min(batch, step) == 0
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.ms_deform_attn_forward(value.transpose(1, 2), shapes, loc, attn, 64)                 # cu:26-29
    with pytest.raises(RuntimeError):
        ext.ms_deform_attn_forward(value, shapes.int(), loc, attn, 64)
    with pytest.raises(RuntimeError, match="inconsistent"):
        ext.ms_deform_attn_forward(value, shapes, loc, attn[:, :, :, :, :1].contiguous(), 64)
    with pytest.raises(RuntimeError, match="grad_output"):
        ext.ms_deform_attn_backward(value, shapes, loc, attn, torch.randn(N, Lq, M * D + 1), 64)
    with pytest.raises(RuntimeError):                                                             # sum H W != S: the library's own check
        ext.ms_deform_attn_forward(value, torch.tensor([[4, 4]]), loc, attn, 64)
