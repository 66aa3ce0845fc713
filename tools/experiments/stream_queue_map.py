"""Round 6 experiment: which HIP streams share a hardware queue decides the rate of the pipelined frame loop.

PyTorch hands out streams from a pool of 32 per priority, created together; the HIP runtime spreads a process's streams over
GPU_MAX_HW_QUEUES (default 4) hardware queues in creation order, so pool stream i sits on queue (i + c) mod 4.  Streams on one
hardware queue execute in submission order -- a false dependency between the decoder half of frame t (the sequence's stream) and
the image-only half of frame t + 1 (GraphedDetector's side stream) when the two happen to collide.  bench.py's legs create
streams in different orders, and their rates came out QUANTISED (284 / 308 / 331 / 368 frames/s for the same schedule).

This tool pins the relation: the 32 pool streams are taken once, the sequence runs on pool[0], the side streams on chosen pool
entries; every case runs cfg 2's pipelined loop (bench.run_tracking's body) with HBM and with host frames.

    python tools/experiments/stream_queue_map.py [--frames 160]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402


def loop(tracker, seeder, frames, steps, depth, image_ready):
    handle, issued, done, ahead = None, 0, 0, 0
    while done < steps:
        nxt = frames[issued % len(frames)] if issued < steps else None
        if handle is not None:
            while ahead < depth and issued + ahead < steps:
                if not tracker.step_prepare(frames[(issued + ahead) % len(frames)], image_ready=image_ready):
                    break
                ahead += 1
            tracker.step_finish(handle)
            handle = None
            done += 1
        if nxt is not None:
            seeder.seed(tracker)
            handle = tracker.step_async(nxt)
            issued += 1
            ahead = max(0, ahead - 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=160)
    ap.add_argument("--cases", default="")
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    from trackformer_amd import runtime
    from trackformer_amd.graphed import GraphedDetector
    runtime.configure_inference(verbose=False)
    cfg = bench.CONFIGS["cfg2"]
    model, _, post, margs = bench.build_model(cfg, device)
    model.tracking()
    seeds = bench.calibrate_association(model, bench.make_frames(device, cfg["size"], n=1)[0], cfg["tracks"], cfg["size"], device)
    seeder = bench.TrackSeeder(device, margs.hidden_dim, cfg["tracks"], cfg["size"], seeds=seeds)
    from trackformer_amd.runtime import bind_streams, pool_stream
    bind_streams(device)                                             # (every pool stream used once, in a fixed order)
    pool = [pool_stream(device, i) for i in range(32)]               # the normal-priority pool, by index
    high = [pool_stream(device, i, -1) for i in range(4)]
    nq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    print("GPU_MAX_HW_QUEUES", nq, "priority range", getattr(torch.cuda.Stream, "priority_range", lambda: "n/a")(), flush=True)
    # (name, main stream, side streams, slots, look-ahead)
    same, d1, d2, d1b = pool[nq], pool[1], pool[2], pool[1 + nq]    # relative to pool[0]: same queue, two other queues, d1's queue again
    print("stream ids: pool", [s.stream_id for s in pool[:10]], "high", [s.stream_id for s in high], flush=True)
    cases = [
        ("1 side stream, pool[0] / pool[1]", pool[0], (d1,), 2, 1),
        ("high[0], 1 side stream pool[1]", high[0], (d1,), 2, 1),
        ("high[0], sides pool[1], pool[2], 2 ahead", high[0], (d1, d2), 4, 2),
    ]
    # the absolute position: side streams (pool[r], pool[r + nq]) for every residue r, sequence on each of the high-priority streams;
    # depth 2 (the pipelined loop) and depth 1 (what the deferred step() loop does)
    for r in range(nq):
        cases.append(("high[0], sides pool[%d], pool[%d], 2 ahead" % (r, r + nq), high[0], (pool[r], pool[r + nq]), 4, 2))
    for r in range(nq):
        cases.append(("high[0], sides pool[%d], pool[%d], 1 ahead" % (r, r + nq), high[0], (pool[r], pool[r + nq]), 4, 1))
    for h in range(1, 4):
        cases.append(("high[%d], sides pool[1], pool[%d], 2 ahead" % (h, 1 + nq), high[h], (pool[1], pool[1 + nq]), 4, 2))
    for r in range(nq):
        cases.append(("normal pool[2], sides pool[%d], pool[%d], 2 ahead" % (r, r + nq), pool[2], (pool[r], pool[r + nq]), 4, 2))
    null = torch.cuda.default_stream(device)     # what a caller who never sets a stream runs on
    cases.append(("default (null) stream, sides pool[1], pool[5], 2 ahead", null, (pool[1], pool[5]), 4, 2))
    cases.append(("default (null) stream, sides pool[1], pool[5], 1 ahead", null, (pool[1], pool[5]), 4, 1))
    cases.append(("default (null) stream, 1 side stream pool[1] (NARROW)", null, (pool[1],), 2, 1))
    if a.cases:
        keep = {int(x) for x in a.cases.split(",")}
        cases = [c for i, c in enumerate(cases) if i in keep]
    frames = {"hbm": bench.make_frames(device, cfg["size"]), "host": bench.make_frames(device, cfg["size"], host=True)}
    from trackformer_amd import config
    from trackformer_amd.tracker import Tracker
    for name, main_stream, sides, slots, depth in cases:
        row = []
        for kind in ("hbm", "host"):
            det = GraphedDetector(model, bucket=1)
            det.SLOTS, det.LOOKAHEAD, det.SIDE_STREAMS = slots, depth, len(sides)
            det._last_alloc = slots - 1
            det._side[device] = tuple(sides)
            tracker = Tracker(det, post, config.tracker_cfg(), False)
            tracker.reset()
            with torch.no_grad(), torch.cuda.stream(main_stream):
                loop(tracker, seeder, frames[kind], 16, depth, kind == "hbm")
                torch.cuda.synchronize()
                best = 0.0
                for _ in range(3):
                    t0 = time.perf_counter()
                    loop(tracker, seeder, frames[kind], a.frames, depth, kind == "hbm")
                    torch.cuda.synchronize()
                    best = max(best, a.frames / (time.perf_counter() - t0))
            row.append("%s %.1f" % (kind, best))
            del tracker, det
        print("%-78s %s" % (name, "   ".join(row)), flush=True)


if __name__ == "__main__":
    main()
