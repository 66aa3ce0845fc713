#!/usr/bin/env python
"""The round-quantisation experiment VERDICT r04 (weak 3 / next 3) asks for: the 128 -> 128 3 x 3 convolution form of
tf_conv_packed_f32 (64-row x 128-column blocks, 36 K-slices of 32) timed at row counts chosen so that the launch is exactly
256, 512, 528, 768 and 1024 blocks, with the K loop whole and cut in two.  If the cost per block jumps past 512 (= 2 resident
blocks x 256 CUs) the launches of a frame (528, 528, 360, 464 blocks) lose a round to the tail and a persistent / stream-K
launch pays; if it is flat they do not.

    python tools/experiments/conv3_tiles.py
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from trackformer_amd import fused  # noqa: E402
from tools.bench_conv import time_it  # noqa: E402


def main():
    import argparse
    from trackformer_amd import _cabi
    ap = argparse.ArgumentParser()
    ap.add_argument("--ti", type=int, default=0, help="row tiles per block (tf_msda_set_option linear_stream_ti): 2 = 64-row blocks, 4 = 128-row blocks")
    args = ap.parse_args()
    if args.ti:
        _cabi.lib().tf_msda_set_option(b"linear_stream_ti", args.ti)
    rows_per_block = 128 if args.ti >= 4 else 64
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cin = cout = 128
    wt = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).contiguous()
    b = torch.randn(cout, device=dev)
    print("%d-row blocks\nrows  pieces  blocks  us/launch  us/block*256  ns/(block slice)*256" % rows_per_block)
    for pieces in (2, 1):
        fused._conv_ksplit = lambda m, k, c, p=pieces: p
        for blocks in (128, 256, 384, 512, 528, 640, 768, 1024, 1056, 2048):
            rows = blocks // pieces * rows_per_block
            x = torch.randn(1, cin, rows // 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
            with torch.no_grad():
                y = fused.conv3x3(x, wt, b, True, 1)
                assert y is not None
                us = time_it(lambda: fused.conv3x3(x, wt, b, True, 1), 20)
            slices = 36 // pieces
            print("%6d %2d %6d  %8.2f  %8.2f  %8.1f" % (rows, pieces, blocks, us, us / blocks * 256, us / blocks / slices * 256 * 1e3))


if __name__ == "__main__":
    main()
