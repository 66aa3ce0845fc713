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
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg2"]
model, criterion, post, margs = bench.build_model(cfg, dev)
model.tracking()
tracker = bench.build_tracker(model, post, use_graph=True)
frames = bench.make_frames(dev, cfg["size"])
seeds = None
if "--no-calibration" not in sys.argv:   # bench.py's default: the association leg has ~100 surviving tracks and ~150 detections
    seeds = bench.calibrate_association(model, frames[0], cfg["tracks"], cfg["size"], dev)
seeder = bench.TrackSeeder(dev, margs.hidden_dim, cfg["tracks"], cfg["size"], seeds=seeds)


def step(i):
    seeder.seed(tracker)
    tracker.step(frames[i % len(frames)])


with torch.no_grad():
    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40):
        step(i)
    torch.cuda.synchronize()
    print("ms/step", (time.perf_counter() - t0) / 40 * 1e3)
    # how long the GPU part alone takes: replay-only loop of the detector
    img = frames[0]['img']
    t0 = time.perf_counter()
    for i in range(40):
        seeder.seed(tracker)
    print("ms/seed (bench artefact: 100 Track objects rebuilt per step)", (time.perf_counter() - t0) / 40 * 1e3)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(40):
        step(i)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
st.sort_stats("cumulative").print_stats(25)
