#!/usr/bin/env python
"""Timing of tf_mha_core_f32 (decoder query self-attention) against torch's SDPA on the same inputs."""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from trackformer_amd import fused  # noqa: E402


def time_it(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()
        s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        g.replay()
        b.record(s)
        b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for length, heads, d in ((300, 8, 32), (400, 8, 32), (800, 8, 36)):
    e = heads * d
    qk = torch.randn(1, length, 2 * e, device="cuda")
    v = torch.randn(1, length, e, device="cuda")
    q = qk[..., :e].view(1, length, heads, d).transpose(1, 2)
    k = qk[..., e:].view(1, length, heads, d).transpose(1, 2)
    vv = v.view(1, length, heads, d).transpose(1, 2)
    from trackformer_amd import _cabi
    _set = lambda v: _cabi.lib().tf_msda_set_option(b"mha_mfma", v)
    own = {}
    for mode in (1, 2, 0):   # matrix cores: operands streamed into registers (default) / staged in LDS; the vector kernel
        prev = _set(mode)
        own[mode] = time_it(lambda: fused.mha_core(qk, v, heads))
        _set(prev)
    lib = time_it(lambda: F.scaled_dot_product_attention(q, k, vv))
    print("L=%d heads=%d d=%d: tf_mha_core_f32 %.1f us on the matrix cores (%.1f us with K, V through LDS; %.1f us vector kernel), torch SDPA %.1f us" % (
        length, heads, d, own[1], own[2], own[0], lib))
