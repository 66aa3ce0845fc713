#!/usr/bin/env python
"""Feasibility experiment (CPU, no GPU needed): can the dense linears of the path run as bf16 split products on the
matrix cores (fp32 = hi + mid + lo bf16 pieces, fp32 accumulation) without leaving the parity bar?

    python tools/experiments/bf16_split_linear.py [x3|x6|x1] [conv|conv1x1] [full]

Every torch.nn.functional.linear / torch._addmm_activation of the model is replaced by an emulation of the split
product -- the pieces are rounded to bf16 exactly as the hardware would see them, the partial products (exact in
fp32, as on MFMA with fp32 accumulation) are summed in fp32 -- and the CPU parity tests that compare the model and
the tracker with the reference goldens (boxes / logits <= 1e-3, track ids exact) are run on top of it.
  x1: plain bf16 (1 MFMA pass)             -- what "just use bf16" would mean
  x3: hi*hi + hi*mid + mid*hi              -- ~16 mantissa bits
  x6: + mid*mid + hi*lo + lo*hi            -- ~fp32
With `conv` the convolutions (backbone, input projections, mask head) are split the same way; with `conv1x1` only the
stride-1 1 x 1 convolutions (what trackformer_amd/backbone.py's opt-in TF_CONV1X1_SPLIT route sends through the split GEMM,
plus the input projections).  `full` adds the BASELINE-size goldens (tests/test_full_size_cpu.py).
Result of the round-1 run: see DESIGN.md section 6 ("next").
"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
MODE = sys.argv[1] if len(sys.argv) > 1 else "x3"
CONV = len(sys.argv) > 2 and sys.argv[2] in ("conv", "conv1x1")
ONLY_1X1 = len(sys.argv) > 2 and sys.argv[2] == "conv1x1"
FULL = "full" in sys.argv[3:]
_orig_conv2d = F.conv2d
_orig_linear = F.linear
_orig_addmm_act = torch._addmm_activation


def _pieces(t):
    hi = t.to(torch.bfloat16).float()
    mid = (t - hi).to(torch.bfloat16).float()
    lo = (t - hi - mid).to(torch.bfloat16).float()
    return hi, mid, lo


def _split_mm(x, w_t):
    """x [.., K] @ w_t [K, N] with bf16 pieces and fp32 accumulation."""
    xh, xm, xl = _pieces(x)
    wh, wm, wl = _pieces(w_t)
    out = xh @ wh
    if MODE in ("x3", "x6"):
        out = out + xh @ wm + xm @ wh
    if MODE == "x6":
        out = out + xm @ wm + xh @ wl + xl @ wh
    return out


def linear(x, w, b=None):
    if x.dtype != torch.float32 or x.is_cuda:
        return _orig_linear(x, w, b)
    y = _split_mm(x, w.t())
    return y if b is None else y + b


def addmm_activation(bias, x, w_t, *, beta=1, alpha=1, use_gelu=False):
    y = _split_mm(x, w_t) + bias
    return F.gelu(y) if use_gelu else torch.relu(y)


def conv2d(x, w, b=None, *args, **kw):
    if x.dtype != torch.float32 or x.is_cuda:
        return _orig_conv2d(x, w, b, *args, **kw)
    if ONLY_1X1:
        stride = args[0] if len(args) > 0 else kw.get("stride", 1)
        groups = args[3] if len(args) > 3 else kw.get("groups", 1)
        if tuple(w.shape[2:]) != (1, 1) or stride not in (1, (1, 1), [1, 1]) or groups != 1:
            return _orig_conv2d(x, w, b, *args, **kw)
    xh, xm, xl = _pieces(x)
    wh, wm, wl = _pieces(w)
    c = lambda a, k: _orig_conv2d(a, k, None, *args, **kw)
    out = c(xh, wh)
    if MODE in ("x3", "x6"):
        out = out + c(xh, wm) + c(xm, wh)
    if MODE == "x6":
        out = out + c(xm, wm) + c(xh, wl) + c(xl, wh)
    return out if b is None else out + b.view(1, -1, 1, 1)


if __name__ == "__main__":
    if CONV:
        F.conv2d = conv2d
        torch.conv2d = conv2d
    F.linear = linear
    torch.nn.functional.linear = linear
    torch._addmm_activation = addmm_activation
    sel = "model_forward_matches_reference or tracker_sequence_matches_reference or tracker_variants_match_reference"
    files = [os.path.join(REPO, "tests", "test_models_cpu.py")]
    if FULL:
        files.append(os.path.join(REPO, "tests", "test_full_size_cpu.py"))
        sel += " or full_size_model_matches_reference"
    sys.exit(pytest.main(files + ["-q", "--no-header", "-k", sel, "-s"]))
