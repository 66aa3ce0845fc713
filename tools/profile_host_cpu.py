#!/usr/bin/env python
"""Host side of Tracker.step alone, on a machine without a GPU: a stand-in detector returns outputs of the real shapes
(300 object queries + the track queries, hidden 256) with scores laid out like bench.py's calibrated association leg
(~93 of 100 track queries stay above the threshold, ~100 object queries become detections), the real PostProcess and the real
Tracker run on them.  Reports ms of host work per frame and a cProfile of it.  What it cannot show: the cost of enqueueing
GPU work (the stand-in's tensors are CPU tensors).

    python tools/profile_host_cpu.py [--tracks 100] [--detections 100] [--frames 200] [--profile]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from trackformer_amd import config  # noqa: E402
from trackformer_amd.deformable_detr import DeformablePostProcess as PostProcess  # noqa: E402
from trackformer_amd.tracker import Track, Tracker  # noqa: E402


class StandInDetector(torch.nn.Module):
    num_queries = 300
    overflow_boxes = False

    def __init__(self, n_det, hidden=256, frames=8, seed=0):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        g = torch.Generator().manual_seed(seed)
        self.n_det, self.hidden = n_det, hidden
        # well separated boxes so that NMS keeps nearly everything (as in the calibrated bench)
        self.obj_boxes = [torch.cat([torch.rand(300, 2, generator=g) * 0.9 + 0.05, torch.rand(300, 2, generator=g) * 0.01 + 0.005], 1)
                          for _ in range(frames)]
        self.obj_hs = [torch.randn(300, hidden, generator=g) for _ in range(frames)]
        self.noise = [torch.rand(1024, generator=g) for _ in range(frames)]
        self.i = 0

    def forward(self, img, target, prev_features):
        k = self.i % len(self.obj_boxes)
        self.i += 1
        n = 0 if target is None else target[0]['track_query_boxes'].shape[0]
        logits = torch.full((1, n + 300, 1), -3.0)
        boxes = torch.empty(1, n + 300, 4)
        hs = torch.empty(1, n + 300, self.hidden)
        if n:
            logits[0, :n, 0] = torch.where(self.noise[k][:n] < 0.93, 2.0, -2.0)
            boxes[0, :n] = target[0]['track_query_boxes']
            hs[0, :n] = target[0]['track_query_hs_embeds']
        logits[0, n:n + self.n_det, 0] = 1.0
        boxes[0, n:] = self.obj_boxes[k]
        hs[0, n:] = self.obj_hs[k]
        return {'pred_logits': logits, 'pred_boxes': boxes, 'hs_embed': hs}, target, None, None, hs[None]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=100)
    ap.add_argument("--detections", type=int, default=100)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--gc", choices=["default", "off", "freeze"], default="default")
    args = ap.parse_args()
    torch.set_num_threads(1)   # the shared CPU of the build container: one thread, CPU time instead of wall time
    det = StandInDetector(args.detections)
    tracker = Tracker(det, {'bbox': PostProcess()}, config.tracker_cfg(), False)
    tracker.reset()
    g = torch.Generator().manual_seed(1)
    n = args.tracks
    pos = torch.cat([torch.rand(n, 2, generator=g) * 1200 + 20, torch.rand(n, 2, generator=g) * 10 + 5], 1)
    pos[:, 2:] += pos[:, :2]
    scores, hs, ind = torch.full((n,), 0.9), torch.randn(n, 256, generator=g), torch.arange(n).view(n, 1)

    def seed():   # bench.py's TrackSeeder: the same n tracks before every step (a steady state)
        tracker.tracks = [Track((pos, i), (scores, i), i, (hs, i), i) for i in range(n)]
        tracker.inactive_tracks = []
        tracker.track_num = n

    blob = {'img': torch.zeros(1, 3, 8, 8), 'orig_size': torch.tensor([[800, 1333]]), 'size': torch.tensor([[800, 1333]]),
            'dets': torch.zeros(1, 0, 4)}

    def run(frames):
        t_seed = t_step = 0.0
        for _ in range(frames):
            t0 = time.process_time()
            seed()
            t1 = time.process_time()
            tracker.step(blob)
            t_step += time.process_time() - t1
            t_seed += t1 - t0
        return t_seed / frames * 1e3, t_step / frames * 1e3

    import gc
    if args.gc == "off":
        gc.disable()
    elif args.gc == "freeze":
        gc.collect()
        gc.freeze()
    with torch.no_grad():
        run(20)
        alive = len(tracker.tracks)
        runs = [run(args.frames) for _ in range(9)]
        s, t = min(r[0] for r in runs), min(r[1] for r in runs)
        print("host ms per frame (CPU time, best of 9 x %d frames): step %.3f (+ seeding %.3f, bench artefact); %d tracks alive after a step"
              % (args.frames, t, s, alive))
        if args.profile:
            pr = cProfile.Profile()
            pr.enable()
            run(args.frames)
            pr.disable()
            pstats.Stats(pr).sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
