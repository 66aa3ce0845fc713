"""CPU: the C oracle (oracle/msda_ref.c) against the golden vectors produced by the reference's own
ms_deform_attn_core_pytorch (tests/golden/make_golden_msda.py).  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import msda_oracle
from tests.util_msda import discontinuity_mask, golden_cases, load_case

CASES = golden_cases()


def _tol(dtype):
    # fp64: pure roundoff of a different summation order; fp32: a few ulp of O(1) sums
    return (1e-12, 1e-10) if dtype == np.float64 else (2e-5, 1e-4)


def test_golden_files_present():
    assert len(CASES) >= 9


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("msda_")[-1][:-4])
def test_forward_matches_reference(path):
    z = load_case(path)
    out = msda_oracle.msda_forward(z["value"], z["shapes"], z["loc"], z["attn"])
    atol, rtol = _tol(z["value"].dtype)
    np.testing.assert_allclose(out, z["out"], atol=atol, rtol=rtol)


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("msda_")[-1][:-4])
def test_backward_matches_reference(path):
    z = load_case(path)
    gv, gl, ga = msda_oracle.msda_backward(z["value"], z["shapes"], z["loc"], z["attn"],
                                           z["grad_out"])
    atol, rtol = _tol(z["value"].dtype)
    np.testing.assert_allclose(gv, z["grad_value"], atol=atol, rtol=rtol)
    np.testing.assert_allclose(ga, z["grad_attn"], atol=atol * 10, rtol=rtol)
    keep = ~discontinuity_mask(z["loc"], z["shapes"])
    np.testing.assert_allclose(gl[keep], z["grad_loc"][keep], atol=atol * 10, rtol=rtol)
    # at the discontinuity the CUDA semantics (which the oracle restates) give exactly zero
    assert np.all(gl[~keep] == 0)


def test_threads_do_not_change_forward():
    z = load_case(CASES[0])
    a = msda_oracle.msda_forward(z["value"], z["shapes"], z["loc"], z["attn"], nthreads=1)
    b = msda_oracle.msda_forward(z["value"], z["shapes"], z["loc"], z["attn"], nthreads=4)
    assert np.array_equal(a, b)


def test_shape_sum_is_checked():
    z = load_case(CASES[0])
    bad = z["shapes"].copy()
    bad[0, 0] += 1
    with pytest.raises(RuntimeError):
        msda_oracle.msda_forward(z["value"], bad, z["loc"], z["attn"])


# ------------------------------------------------------------------ the restated grid_sample path
@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("msda_")[-1][:-4])
def test_grid_sample_restatement_matches_reference(path):
    """oracle/msda_grid_sample.py (the reference's pure-CPU path restated, bench.py's cpu_baseline
    operator) against outputs AND gradients of the reference's own ms_deform_attn_core_pytorch."""
    import torch
    from oracle.msda_grid_sample import msda_grid_sample
    z = load_case(path)
    value = torch.from_numpy(z["value"]).requires_grad_(True)
    loc = torch.from_numpy(z["loc"]).requires_grad_(True)
    attn = torch.from_numpy(z["attn"]).requires_grad_(True)
    out = msda_grid_sample(value, torch.from_numpy(z["shapes"]), loc, attn)
    atol, rtol = _tol(z["value"].dtype)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], atol=atol, rtol=rtol)
    out.backward(torch.from_numpy(z["grad_out"]).reshape(out.shape))
    np.testing.assert_allclose(value.grad.numpy(), z["grad_value"], atol=atol, rtol=rtol)
    np.testing.assert_allclose(attn.grad.numpy(), z["grad_attn"], atol=atol * 10, rtol=rtol)
    np.testing.assert_allclose(loc.grad.numpy(), z["grad_loc"], atol=atol * 10, rtol=rtol)
