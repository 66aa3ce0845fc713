"""Shared helpers for the MSDeformAttn tests (input generators restating ops/test.py:24-27)."""
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# cfg-2 pyramid of an 800x1333 frame: strides 8,16,32,64
CFG2_SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]


def golden_cases(pattern="msda_*.npz"):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def load_case(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def discontinuity_mask(loc, shapes):
    """True where a sample sits exactly on / outside the in-range boundary (px <= -1 or px >= size).

    There the CUDA kernels define all gradients as 0 (cuh:359-362, :73-77) while F.grid_sample (the
    golden generator) uses a one-sided derivative at px == -1 / px == size exactly; the forward value is
    0 in both.  grad_loc comparisons against the torch goldens skip these measure-zero points."""
    H = shapes[:, 0].reshape(1, 1, 1, -1, 1)
    W = shapes[:, 1].reshape(1, 1, 1, -1, 1)
    px = loc[..., 0] * W - 0.5
    py = loc[..., 1] * H - 0.5
    return (px <= -1) | (py <= -1) | (px >= W) | (py >= H)


def rand_inputs(seed, N, M, D, Lq, P, shapes, dtype=torch.float32, loc_mode="rand", device="cpu",
                value_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    shapes_t = torch.as_tensor(shapes, dtype=torch.long)
    L = len(shapes)
    S = int((shapes_t[:, 0] * shapes_t[:, 1]).sum())
    value = (torch.rand(N, S, M, D, generator=g) * value_scale).to(dtype)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    if loc_mode == "wide":
        loc = loc * 2.0 - 0.5
    elif loc_mode == "local":   # reference point + N(0, 3 px) per level, like a trained encoder
        ref = torch.rand(N, Lq, 1, 1, 1, 2, generator=g)
        sz = torch.stack([shapes_t[:, 1], shapes_t[:, 0]], -1).view(1, 1, 1, L, 1, 2).float()
        loc = ref + torch.randn(N, Lq, M, L, P, 2, generator=g) * 3.0 / sz
    loc = loc.to(dtype)
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    grad_out = torch.randn(N, Lq, M * D, generator=g).to(dtype)
    return (value.to(device), shapes_t.to(device), loc.to(device), attn.to(device),
            grad_out.to(device))
