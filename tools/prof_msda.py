#!/usr/bin/env python
"""Run one MSDeformAttn forward configuration a few times (for rocprofv3 --pmc / --kernel-trace)."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tools.bench_msda import CONFIGS, make_inputs  # noqa: E402
from trackformer_amd import msda  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="cfg2_encoder")
ap.add_argument("--mode", default="init")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--backward", action="store_true")
a = ap.parse_args()
kw = dict(CONFIGS)[a.shape]
S = sum(h * w for h, w in kw["shapes"])
value, shapes, loc, attn, go = make_inputs(mode=a.mode, device="cuda:0",
                                           encoder_refs=(kw["Lq"] == S), **kw)
for _ in range(a.iters):
    if a.backward:
        msda.ms_deform_attn_backward(value, shapes, loc, attn, go, 64)
    else:
        msda.ms_deform_attn_forward(value, shapes, loc, attn, 64)
torch.cuda.synchronize()
