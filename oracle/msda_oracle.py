"""oracle/msda_oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by trackformer_amd/).

ctypes front-end over oracle/msda_ref.c, the plain-C CPU restatement of the reference's
MSDeformAttn arithmetic (ops/src/cuda/ms_deform_im2col_cuda.cuh; line map in msda_ref.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

Parity status: pinned by tests/golden/msda_*.npz, which were produced by the reference's own
`ms_deform_attn_core_pytorch` (ops/functions/ms_deform_attn_func.py:34-54) in the build
container (tests/golden/make_golden_msda.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libmsda_oracle.so")
_lib = None


def build(force=False):
    """Compile msda_ref.c with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "msda_ref.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for suf in ("f32", "f64"):
            getattr(_lib, "oracle_msda_forward_" + suf).restype = ctypes.c_int
            getattr(_lib, "oracle_msda_backward_" + suf).restype = ctypes.c_int
    return _lib


def _prep(value, shapes, loc, attn):
    value = np.ascontiguousarray(value)
    dt = value.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError("oracle supports float32/float64 only")
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and shapes.shape == (L, 2) and attn.shape == (N, Lq, M, L, P)
    return value, shapes, loc, attn, (N, S, M, D, L, Lq, P), ("f32" if dt == np.float32 else "f64")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def msda_forward(value, shapes, loc, attn, nthreads=1):
    """numpy in, numpy out: out[N, Lq, M*D]."""
    value, shapes, loc, attn, dims, suf = _prep(value, shapes, loc, attn)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dtype=value.dtype)
    rc = getattr(_load(), "oracle_msda_forward_" + suf)(
        _p(value), _p(shapes), _p(loc), _p(attn), _p(out),
        *[ctypes.c_int(v) for v in (N, S, M, D, L, Lq, P, int(nthreads))])
    if rc != 0:
        raise RuntimeError("oracle_msda_forward failed rc=%d" % rc)
    return out


def msda_backward(value, shapes, loc, attn, grad_out):
    """Returns (grad_value, grad_loc, grad_attn) as numpy arrays."""
    value, shapes, loc, attn, dims, suf = _prep(value, shapes, loc, attn)
    N, S, M, D, L, Lq, P = dims
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype).reshape(N, Lq, M * D)
    gv = np.zeros_like(value)
    gl = np.zeros_like(loc)
    ga = np.zeros_like(attn)
    rc = getattr(_load(), "oracle_msda_backward_" + suf)(
        _p(value), _p(shapes), _p(loc), _p(attn), _p(grad_out), _p(gv), _p(gl), _p(ga),
        *[ctypes.c_int(v) for v in (N, S, M, D, L, Lq, P)])
    if rc != 0:
        raise RuntimeError("oracle_msda_backward failed rc=%d" % rc)
    return gv, gl, ga


def make_torch_function():
    """An autograd.Function over the C oracle so that tests can run whole nn.Modules on CPU.

    Used by tests only (monkeypatched in place of the HIP operator)."""
    import torch

    class OracleMSDAFunction(torch.autograd.Function):
        @staticmethod
        def forward(ctx, value, shapes, loc, attn, im2col_step=64):
            ctx.save_for_backward(value, shapes, loc, attn)
            out = msda_forward(value.detach().cpu().numpy(), shapes.cpu().numpy(),
                               loc.detach().cpu().numpy(), attn.detach().cpu().numpy())
            return torch.from_numpy(out).to(value.device)

        @staticmethod
        def backward(ctx, grad_output):
            value, shapes, loc, attn = ctx.saved_tensors
            gv, gl, ga = msda_backward(value.detach().cpu().numpy(), shapes.cpu().numpy(),
                                       loc.detach().cpu().numpy(), attn.detach().cpu().numpy(),
                                       grad_output.detach().cpu().contiguous().numpy())
            dev = value.device
            return (torch.from_numpy(gv).to(dev), None, torch.from_numpy(gl).to(dev),
                    torch.from_numpy(ga).to(dev), None)

    return OracleMSDAFunction
