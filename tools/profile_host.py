#!/usr/bin/env python
"""cProfile of the host side of bench.py's step loop (where does the non-GPU time go?)."""
import cProfile
import os
import pstats
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from trackformer_amd import runtime  # noqa: E402

runtime.configure_inference()
dev = torch.device("cuda:0")
tracker, model = bench.build_tracker(dev, use_graph=True)
seeder = bench.TrackSeeder(dev, model.hidden_dim)
frames = bench.make_frames(dev)


def step(i):
    seeder.seed(tracker)
    tracker.step(frames[i % len(frames)])


with torch.no_grad():
    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    # wall-clock split: time until the packed D2H returns vs the rest
    t0 = time.perf_counter()
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
