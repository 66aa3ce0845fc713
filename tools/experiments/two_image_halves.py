#!/usr/bin/env python
"""How much do two image-only halves (backbone + input projections + encoder of one 800 x 1333 frame each, HIP graphs) gain from running
CONCURRENTLY on two streams instead of one after the other?  (The pipelined loop runs one such half per frame on the wrapper's side
stream; a second frame of look-ahead could overlap two of them.)   python tools/experiments/two_image_halves.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from trackformer_amd import runtime
    from trackformer_amd.graphed import GraphedDetector
    runtime.configure_inference()
    model, _, post, margs = bench.build_model(bench.CONFIGS["cfg2"], dev)
    model.eval().tracking()
    dets = [GraphedDetector(model, bucket=1) for _ in range(2)]
    img = torch.randn(1, 3, 800, 1333, device=dev)
    with torch.no_grad():
        for d in dets:
            for _ in range(3):
                d(img, None, None)                       # graphs exist (slot 0)
    graphs = [d._enc[(tuple(img.shape), dev)][0]["graph"] for d in dets]
    s = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

    def run(n, concurrent):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for k in range(2):
                with torch.cuda.stream(s[k] if concurrent else s[0]):
                    graphs[k].replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (2 * n) * 1e3

    for _ in range(2):
        print("one after the other: %.3f ms per half;  on two streams: %.3f ms per half" % (run(50, False), run(50, True)), flush=True)


if __name__ == "__main__":
    main()
