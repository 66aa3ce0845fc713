#!/usr/bin/env python
"""The mask head's convolutions at the 800 x 1333 frame (128 queries per chunk) through fused.conv3x3: time per layer for the block
shapes of the halo form (linear_stream_ti) and the block kernel.  python tools/experiments/mask_head_convs.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from trackformer_amd import _cabi, fused  # noqa: E402

LAYERS = [("lay2 288->128 @25x42", 128, 288, 25, 42, 128), ("lay3 128->64 @50x84", 128, 128, 50, 84, 64),
          ("lay4 64->32 @100x167", 128, 64, 100, 167, 32), ("lay5 32->16 @200x334", 128, 32, 200, 334, 16)]


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    dev = torch.device("cuda:0")
    lib = _cabi.lib()
    for name, n, cin, h, w, cout in LAYERS:
        x = torch.randn(n, h, w, cin, device=dev).permute(0, 3, 1, 2)
        taps = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).contiguous()
        b = torch.randn(cout, device=dev)
        row = [name]
        for label, opts in (("halo default", {}), ("halo ti=1", {"linear_stream_ti": 1}), ("halo ti=2", {"linear_stream_ti": 2}),
                            ("halo ti=4", {"linear_stream_ti": 4}), ("block kernel", {"stream": False})):
            prev_stream = fused.set_conv_stream(opts.get("stream", True))
            prev_ti = lib.tf_msda_set_option(b"linear_stream_ti", opts.get("linear_stream_ti", 0))
            try:
                us = timed(lambda: fused.conv3x3(x, taps, b, False, 1))
            finally:
                lib.tf_msda_set_option(b"linear_stream_ti", prev_ti)
                fused.set_conv_stream(prev_stream)
            row.append("%s %.0f us" % (label, us))
        gb = (n * h * w * (cin + cout) * 4) / 1e9
        print("  ".join(row), " | %.2f GB in + out" % gb, flush=True)


if __name__ == "__main__":
    main()
