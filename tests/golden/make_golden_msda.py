#!/usr/bin/env python
"""Generate tests/golden/msda_*.npz from the REFERENCE's own pure-PyTorch MSDeformAttn.

Run in the build container only (needs /root/reference; the GPU box does not have it):

    python tests/golden/make_golden_msda.py

The reference function `ms_deform_attn_core_pytorch`
(/root/reference/src/trackformer/models/ops/functions/ms_deform_attn_func.py:34-54) is imported
unmodified; the only stub is an empty `MultiScaleDeformableAttention` module, needed because
func.py:11 imports the CUDA extension at module top.  Gradients come from torch autograd through
that function, exactly as the reference's ops/test.py:38-95 obtains its "pytorch" gradients.

Case generators restate ops/test.py:14-27 (seed 3, value = rand*0.01, loc = rand,
attn = (rand+1e-5) normalised over L*P) and ops/test_double_precision.py:16 (non-square levels),
plus edge cases the CUDA kernels special-case (out-of-range samples, exact borders, integer pixel
centres) and the real head/level geometry of the BASELINE configs (D=32/L=4, D=36/L=8).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src/trackformer/models/ops"


def import_reference_core():
    sys.modules.setdefault("MultiScaleDeformableAttention",
                           types.ModuleType("MultiScaleDeformableAttention"))
    sys.path.insert(0, REF_SRC)
    from functions.ms_deform_attn_func import ms_deform_attn_core_pytorch  # noqa: E402
    return ms_deform_attn_core_pytorch


def gen_inputs(seed, N, M, D, Lq, P, shapes, dtype, loc_mode="rand", value_scale=0.01):
    g = torch.Generator().manual_seed(seed)
    shapes_t = torch.as_tensor(shapes, dtype=torch.long)
    L = len(shapes)
    S = int((shapes_t[:, 0] * shapes_t[:, 1]).sum())
    value = (torch.rand(N, S, M, D, generator=g) * value_scale).to(dtype)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    if loc_mode == "wide":          # many samples outside [0,1]: exercises the in-range rule
        loc = loc * 2.0 - 0.5
    elif loc_mode == "border":      # exact borders / pixel centres / just outside
        H = shapes_t[:, 0].view(1, 1, 1, L, 1).float()
        W = shapes_t[:, 1].view(1, 1, 1, L, 1).float()
        ix = torch.randint(-1, 3, (N, Lq, M, L, P), generator=g).float()
        iy = torch.randint(-1, 3, (N, Lq, M, L, P), generator=g).float()
        # pixel px = loc*W - 0.5  ->  loc = (px + 0.5)/W with px in {-1, 0, W-1, W} (+ tiny jitter on half)
        px = torch.where(ix < 0, -torch.ones_like(W * ix), torch.where(ix == 0, 0 * ix,
             torch.where(ix == 1, W - 1 + 0 * ix, W + 0 * ix)))
        py = torch.where(iy < 0, -torch.ones_like(H * iy), torch.where(iy == 0, 0 * iy,
             torch.where(iy == 1, H - 1 + 0 * iy, H + 0 * iy)))
        jitter = (torch.rand(N, Lq, M, L, P, 2, generator=g) - 0.5) * 0.5
        half = (torch.rand(N, Lq, M, L, P, generator=g) < 0.5).float()
        lx = (px + 0.5 + jitter[..., 0] * half) / W
        ly = (py + 0.5 + jitter[..., 1] * half) / H
        loc = torch.stack([lx, ly], -1)
    elif loc_mode == "grid":        # encoder-like: reference point at a pixel centre + k-pixel offsets
        H = shapes_t[:, 0].view(1, 1, 1, L, 1).float()
        W = shapes_t[:, 1].view(1, 1, 1, L, 1).float()
        ref = torch.rand(N, Lq, 1, 1, 1, 2, generator=g)
        off = torch.randint(-4, 5, (N, Lq, M, L, P, 2), generator=g).float()
        loc = ref + off / torch.stack([H, W], -1)   # reproduces ms_deform_attn.py:79's (H,W) divisor
    loc = loc.to(dtype)
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    grad_out = torch.randn(N, Lq, M * D, generator=g).to(dtype)
    return value, shapes_t, loc, attn, grad_out


CASES = [
    # name, kwargs
    ("test_py_f32", dict(seed=3, N=2, M=2, D=4, Lq=3, P=2, shapes=[(8, 8), (4, 4), (2, 2)],
                         dtype=torch.float32)),
    ("test_py_f64", dict(seed=3, N=2, M=2, D=4, Lq=3, P=2, shapes=[(8, 8), (4, 4), (2, 2)],
                         dtype=torch.float64)),
    ("test_dp_nonsquare_f64", dict(seed=3, N=2, M=2, D=4, Lq=3, P=2,
                                   shapes=[(12, 8), (6, 4), (3, 2)], dtype=torch.float64)),
    ("wide_f32", dict(seed=11, N=2, M=4, D=8, Lq=37, P=3, shapes=[(9, 7), (5, 4), (2, 3)],
                      dtype=torch.float32, loc_mode="wide", value_scale=1.0)),
    ("wide_f64", dict(seed=12, N=1, M=4, D=8, Lq=37, P=3, shapes=[(9, 7), (5, 4), (2, 3)],
                      dtype=torch.float64, loc_mode="wide", value_scale=1.0)),
    ("border_f64", dict(seed=13, N=1, M=2, D=4, Lq=64, P=4, shapes=[(6, 5), (3, 3), (1, 2)],
                        dtype=torch.float64, loc_mode="border", value_scale=1.0)),
    ("cfg2_geom_f32", dict(seed=21, N=1, M=8, D=32, Lq=100, P=4,
                           shapes=[(13, 21), (7, 11), (4, 6), (2, 3)], dtype=torch.float32,
                           loc_mode="grid", value_scale=1.0)),
    ("cfg4_geom_f32", dict(seed=22, N=1, M=8, D=36, Lq=48, P=4,
                           shapes=[(7, 11), (4, 6), (2, 3), (1, 2)] * 2, dtype=torch.float32,
                           loc_mode="wide", value_scale=1.0)),
    ("odd_dims_f32", dict(seed=23, N=3, M=3, D=5, Lq=17, P=1, shapes=[(7, 3), (1, 1)],
                          dtype=torch.float32, loc_mode="wide", value_scale=1.0)),
]


def main():
    core = import_reference_core()
    torch.set_num_threads(1)
    for name, kw in CASES:
        value, shapes, loc, attn, grad_out = gen_inputs(**kw)
        value.requires_grad_(True)
        loc.requires_grad_(True)
        attn.requires_grad_(True)
        out = core(value, shapes, loc, attn)
        gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), grad_out)
        path = os.path.join(HERE, "msda_%s.npz" % name)
        np.savez_compressed(
            path, value=value.detach().numpy(), shapes=shapes.numpy(), loc=loc.detach().numpy(),
            attn=attn.detach().numpy(), grad_out=grad_out.numpy(), out=out.detach().numpy(),
            grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_attn=ga.numpy())
        print("%-24s out%s  |out|max=%.3e  -> %s (%d KB)" % (
            name, tuple(out.shape), float(out.detach().abs().max()), os.path.basename(path),
            os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
