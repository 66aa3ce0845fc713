#!/usr/bin/env python
"""Per-layer timing of the backbone's bottleneck convolutions at the BASELINE frame size (800 x 1333): the library
convolution + the fused shift / residual / ReLU pass (the default) against the split-product routes (opt-in:
backbone.set_conv1x1_split / set_conv3x3_split), same folded weights, same inputs.  One line per distinct shape with its
multiplicity in ResNet-50, and the per-frame totals -- what decides, per shape, which route the backbone takes.

    python tools/bench_conv.py [--iters 20]
"""
import argparse
import os
import sys

import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from trackformer_amd import backbone  # noqa: E402

# (name, cin, cout, kernel, stride, H_in, W_in, residual, how many times per frame)
SHAPES = [
    ("layer1.0.conv1", 64, 64, 1, 1, 200, 334, False, 1), ("layer1.x.conv2", 64, 64, 3, 1, 200, 334, False, 3),
    ("layer1.x.conv3", 64, 256, 1, 1, 200, 334, True, 3), ("layer1.0.downsample", 64, 256, 1, 1, 200, 334, False, 1),
    ("layer1.1-2.conv1", 256, 64, 1, 1, 200, 334, False, 2),
    ("layer2.0.conv1", 256, 128, 1, 1, 200, 334, False, 1), ("layer2.0.conv2", 128, 128, 3, 2, 200, 334, False, 1),
    ("layer2.x.conv3", 128, 512, 1, 1, 100, 167, True, 4), ("layer2.0.downsample", 256, 512, 1, 2, 200, 334, False, 1),
    ("layer2.1-3.conv1", 512, 128, 1, 1, 100, 167, False, 3), ("layer2.1-3.conv2", 128, 128, 3, 1, 100, 167, False, 3),
    ("layer3.0.conv1", 512, 256, 1, 1, 100, 167, False, 1), ("layer3.0.conv2", 256, 256, 3, 2, 100, 167, False, 1),
    ("layer3.x.conv3", 256, 1024, 1, 1, 50, 84, True, 6), ("layer3.0.downsample", 512, 1024, 1, 2, 100, 167, False, 1),
    ("layer3.1-5.conv1", 1024, 256, 1, 1, 50, 84, False, 5), ("layer3.1-5.conv2", 256, 256, 3, 1, 50, 84, False, 5),
    ("layer4.0.conv1", 1024, 512, 1, 1, 50, 84, False, 1), ("layer4.0.conv2", 512, 512, 3, 2, 50, 84, False, 1),
    ("layer4.x.conv3", 512, 2048, 1, 1, 25, 42, True, 3), ("layer4.0.downsample", 1024, 2048, 1, 2, 50, 84, False, 1),
    ("layer4.1-2.conv1", 2048, 512, 1, 1, 25, 42, False, 2), ("layer4.1-2.conv2", 512, 512, 3, 1, 25, 42, False, 2),
]


def time_it(fn, iters):
    for _ in range(3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    total_lib = total_split = total_best = 0.0
    print("%-22s %5s %5s k s %9s  %10s %10s  %6s  x" % ("convolution", "cin", "cout", "HxW", "library us", "split us", "ratio"))
    with torch.no_grad():
        for name, cin, cout, ks, stride, h, w, res, times in SHAPES:
            conv = nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=False).to(dev)
            bn = backbone.FrozenBatchNorm2d(cout).to(dev)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.1)
            bn.running_mean.normal_(0, 0.1)
            bn.running_var.uniform_(0.5, 1.5)
            x = torch.randn(1, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
            ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
            r = torch.randn(1, cout, ho, wo, device=dev).contiguous(memory_format=torch.channels_last) if res else None
            cache = backbone._FoldCache()
            run = lambda: backbone._conv_bn(x, conv, bn, cache, True, True, residual=r)   # noqa: E731
            p1, p3 = backbone.set_conv1x1_split(False), backbone.set_conv3x3_split(False)
            try:
                ref = run().clone()
                t_lib = time_it(run, args.iters)
                backbone.set_conv1x1_split(True)
                backbone.set_conv3x3_split(True)
                got = run().clone()
                t_split = time_it(run, args.iters)
            finally:
                backbone.set_conv1x1_split(p1)
                backbone.set_conv3x3_split(p3)
            err = float((got - ref).abs().max()) / max(1e-30, float(ref.abs().max()))
            total_lib += times * t_lib
            total_split += times * t_split
            total_best += times * min(t_lib, t_split)
            print("%-22s %5d %5d %d %d %4dx%-4d  %10.1f %10.1f  %6.2f  %d   max rel err %.1e" % (
                name, cin, cout, ks, stride, h, w, t_lib, t_split, t_lib / t_split, times, err))
    print("per frame: library %.0f us, split routes %.0f us, best of both per shape %.0f us" % (total_lib, total_split, total_best))
    # the stem: 7 x 7 convolution (+ shift, ReLU, pooling) through the library + bias_act + max_pool2d, and through the own kernels
    from trackformer_amd import fused
    with torch.no_grad():
        conv, bn, pool = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).to(dev), backbone.FrozenBatchNorm2d(64).to(dev), nn.MaxPool2d(3, 2, 1)
        bn.weight.uniform_(0.5, 1.5)
        bn.running_var.uniform_(0.5, 1.5)
        x = torch.randn(1, 3, 800, 1333, device=dev)
        cache = backbone._FoldCache()
        lib = lambda: pool(backbone._conv_bn(x, conv, bn, cache, True, True))   # noqa: E731
        own = lambda: backbone._stem_pooled(x, conv, bn, pool, cache)            # noqa: E731
        t_lib = time_it(lib, args.iters)
        res = []
        for conv_split, pool_fused in ((False, True), (True, False), (True, True)):
            p1, p2 = fused.set_stem_conv_split(conv_split), fused.set_stem_pool_fused(pool_fused)
            try:
                err = float((own() - lib()).abs().max())
                res.append((conv_split, pool_fused, time_it(own, args.iters), err))
            finally:
                fused.set_stem_conv_split(p1)
                fused.set_stem_pool_fused(p2)
        print("stem (conv 7x7 + shift + ReLU + pooling): library + bias_act + max_pool2d %.1f us" % t_lib)
        for conv_split, pool_fused, t, err in res:
            print("  own convolution %s, fused pooling pass %s: %.1f us (max abs difference %.1e)" % (conv_split, pool_fused, t, err))


if __name__ == "__main__":
    main()
