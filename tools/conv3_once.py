#!/usr/bin/env python
"""A few eager launches of the split-product 3 x 3 convolution at two ResNet-50 shapes of the 800 x 1333 frame (layer1: 64 -> 64
at 200 x 334; layer3: 256 -> 256 at 50 x 84) -- the command tools/pmc_conv3.sh profiles.

    python tools/conv3_once.py [--iters 6]
"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from trackformer_amd import fused  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=6)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for cin, cout, h, w in ((64, 64, 200, 334), (128, 128, 100, 167), (256, 256, 50, 84)):
        x = torch.randn(1, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).contiguous()
        b = torch.randn(cout, device=dev)
        for _ in range(args.iters):
            y = fused.conv3x3(x, wt, b, True, 1)
        torch.cuda.synchronize()
        print(cin, cout, h, w, tuple(y.shape), float(y.abs().mean()))


if __name__ == "__main__":
    main()
