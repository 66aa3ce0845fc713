#!/usr/bin/env python
"""bench.py -- TrackFormer-Deformable per-frame inference throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md section 8d "cfg 2"): DeformableDETRTracking
(ResNet-50, 4 feature levels, hidden 256, 6+6 layers), synthetic 800x1333 frames already resident in
HBM, 300 object queries + exactly 100 track queries per frame, batch 1, fp32, weights = seeded default
initialisation (no checkpoints / datasets exist offline).  One step = one `Tracker.step(blob)`:
detector forward, post-processing, the single packed device->host copy and the host-side association
(thresholds, NMS, id bookkeeping).  Before every step the tracker is re-seeded with the same 100
synthetic tracks so that each step has exactly 300+100 queries.

Multi-GPU: the path shards by video sequence (engine.py:289-303 of the reference); every rank tracks
its own sequence on its own GPU, there is no collective in the data path ("scaling": "weak").  RCCL is
used only for the barriers around the timed region and the max-over-ranks of the elapsed time.

The JSON line also carries
  roofline      -- the dominant custom kernel, MSDeformAttn forward at the encoder call shape
                   (N=1, S=Lq=22223, M=8, D=32, L=4, P=4): algorithmic bytes (79.65 MB) /
                   average launch duration measured here with HIP events on the launch stream
                   (K launches replayed from one HIP graph so the host cannot be the bottleneck).
  cpu_baseline  -- the same frame on the host CPU: identical nn.Modules on CPU with the C oracle
                   (oracle/msda_ref.c, a port of the reference kernels' arithmetic) as the operator,
                   rank 0, N=1 only, a bounded sample.
"""
import argparse
import json
import os
import sys
import time

# RCCL / cross-process device memory on this stack need dmabuf IPC (see the environment notes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

IMG_H, IMG_W = 800, 1333
NUM_TRACK_QUERIES = 100
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-graph", action="store_true", help="run the detector eagerly (no HIP graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=3)
    ap.add_argument("--host-frames", action="store_true",
                    help="keep the frames in pinned host memory (PCIe copy inside the timed region); "
                         "the reported headline value always uses HBM-resident frames")
    ap.add_argument("--split-linear", action="store_true",
                    help="encoder / decoder linears as bf16 split products on the matrix cores (tf_linear_split_f32; "
                         "same as TF_SPLIT_LINEAR=1).  Verified against the goldens, off by default until measured "
                         "end to end")
    ap.add_argument("--sequences", type=int, default=4,
                    help="independent video sequences tracked concurrently per GPU (one host thread "
                         "and HIP stream each); frames of one sequence stay strictly sequential")
    return ap.parse_args()


def init_distributed(n_gpus):
    from trackformer_amd import dist_utils as du
    rank, local_rank, world = du.env_world()
    if world != n_gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d" % (n_gpus, world, n_gpus))
    if world > 1:
        torch.cuda.set_device(local_rank)
        du.init_from_env(backend="nccl", device=torch.device("cuda", local_rank))  # nccl == RCCL
    return rank, local_rank, world


def barrier(world):
    from trackformer_amd import dist_utils as du
    du.barrier()


def build_tracker(device, use_graph, model=None):
    from trackformer_amd import config, factory
    from trackformer_amd.deformable_detr import DeformablePostProcess
    from trackformer_amd.tracker import Tracker
    post = {'bbox': DeformablePostProcess()}
    if model is None:
        args = config.make_args('deformable', 'tracking', 'mot17', device=str(device))
        torch.manual_seed(42)   # cfgs/train.yaml:112
        model, _, post = factory.build_model(args)
        model.to(device)
        model.tracking()
    detector = model
    if use_graph:
        from trackformer_amd.graphed import GraphedDetector
        detector = GraphedDetector(model)
    tracker = Tracker(detector, post, config.tracker_cfg(), False)
    tracker.reset()
    return tracker, model


class TrackSeeder:
    """Re-seeds the tracker with the same 100 synthetic tracks before every step."""

    def __init__(self, device, hidden_dim, seed=0):
        from trackformer_amd.box_ops import box_cxcywh_to_xyxy
        g = torch.Generator().manual_seed(seed)
        n = NUM_TRACK_QUERIES
        centres = torch.rand(n, 2, generator=g) * 0.8 + 0.1
        sizes = torch.rand(n, 2, generator=g) * 0.18 + 0.02
        self.pos = box_cxcywh_to_xyxy(torch.cat([centres, sizes], 1)) * torch.tensor(
            [IMG_W, IMG_H, IMG_W, IMG_H], dtype=torch.float32)
        self.scores = torch.full((n,), 0.9)
        self.hs = torch.randn(n, hidden_dim, generator=g).to(device)
        self.obj_ind = torch.arange(n).view(n, 1)

    def seed(self, tracker):
        from trackformer_amd.tracker import Track
        tracker.tracks = [Track(self.pos[i], self.scores[i], i, self.hs[i], self.obj_ind[i])
                          for i in range(NUM_TRACK_QUERIES)]
        tracker.inactive_tracks = []
        tracker.track_num = NUM_TRACK_QUERIES


def make_frames(device, n=4, host=False):
    frames = []
    for i in range(n):
        g = torch.Generator().manual_seed(i)
        img = torch.randn(1, 3, IMG_H, IMG_W, generator=g)
        img = img.pin_memory() if host else img.to(device)
        frames.append({'img': img, 'orig_size': torch.tensor([[IMG_H, IMG_W]]),
                       'size': torch.tensor([[IMG_H, IMG_W]]), 'dets': torch.zeros(1, 0, 4)})
    return frames


def measure_roofline(device, launches=50):
    """HIP-event timing of the MSDeformAttn forward kernel at the cfg-2 encoder shape."""
    from tools.bench_msda import CFG2_SHAPES, algorithmic_bytes, make_inputs
    from trackformer_amd import msda
    S = sum(h * w for h, w in CFG2_SHAPES)
    dims = dict(N=1, S=S, M=8, D=32, L=4, Lq=S, P=4)
    # sampling locations as the seeded default-initialised model produces them (zero offset weights,
    # 8-direction bias grid; ms_deform_attn.py:34-41): reference point + k/(H_l, W_l)
    value, shapes, loc, attn, _ = make_inputs(1, 8, 32, S, 4, CFG2_SHAPES, "init", device,
                                              encoder_refs=True)
    stream = torch.cuda.Stream(device)
    stream.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(stream):
        for _ in range(3):
            msda.ms_deform_attn_forward(value, shapes, loc, attn, 64)
        graph = torch.cuda.CUDAGraph()
        stream.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            for _ in range(launches):
                msda.ms_deform_attn_forward(value, shapes, loc, attn, 64)
        graph.replay()
        stream.synchronize()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(stream)
        graph.replay()
        end.record(stream)
        end.synchronize()
    us = start.elapsed_time(end) * 1e3 / launches
    alg = algorithmic_bytes(**dims)
    achieved = alg / (us * 1e-6) / 1e9
    # HBM traffic of the same kernel/shape from the PMC counters: collected offline (rocprofv3 --pmc
    # needs its own passes) and committed together with the method; see the file's "_how"
    mode = os.environ.get("TF_MSDA_TILED", "2")[:1] or "2"   # the library's kernel choice for this shape
    kernel = {"0": "msda_fwd_f32_direct", "1": "msda_fwd_f32_win"}.get(mode, "msda_fwd_f32_quad")
    traffic = None
    try:
        with open(os.path.join(REPO, "profiles", "r01_msda_fwd_quad_traffic.json")) as f:
            traffic = json.load(f)[kernel]["hbm_traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return {"bound": "hbm", "kernel": kernel + " (encoder shape, Lq=S=22223)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "algorithmic_bytes": alg, "avg_launch_us": round(us, 2), "launches": launches}


def measure_cpu_baseline(frames):
    """The same frame on the host: identical modules on CPU, C oracle as the MSDeformAttn operator."""
    from oracle import msda_oracle
    from trackformer_amd import config, factory, msda
    from trackformer_amd.tracker import Tracker
    cores = torch.get_num_threads()
    msda_oracle.build()
    args = config.make_args('deformable', 'tracking', 'mot17', device='cpu')
    torch.manual_seed(42)
    model, _, post = factory.build_model(args)
    model.tracking()
    saved = msda.MSDeformAttnFunction

    class HostOp(torch.autograd.Function):   # checker used as the timed CPU baseline, never shipped
        @staticmethod
        def forward(ctx, value, shapes, loc, attn, step):
            out = msda_oracle.msda_forward(value.numpy(), shapes.numpy(), loc.numpy(),
                                           attn.numpy(), nthreads=cores)
            return torch.from_numpy(out)

    msda.MSDeformAttnFunction = HostOp
    try:
        tracker = Tracker(model, post, config.tracker_cfg(), False)
        tracker.reset()
        seeder = TrackSeeder(torch.device('cpu'), args.hidden_dim)
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(frames):
                seeder.seed(tracker)
                g = torch.Generator().manual_seed(i)
                blob = {'img': torch.randn(1, 3, IMG_H, IMG_W, generator=g),
                        'orig_size': torch.tensor([[IMG_H, IMG_W]]),
                        'size': torch.tensor([[IMG_H, IMG_W]]), 'dets': torch.zeros(1, 0, 4)}
                tracker.step(blob)
        dt = time.perf_counter() - t0
    finally:
        msda.MSDeformAttnFunction = saved
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frame(s) of the same workload (800x1333, 300+100 queries) through the "
                      "same nn.Modules on CPU (torch CPU kernels, %d threads) with oracle/msda_ref.c "
                      "as the MSDeformAttn operator; %.1f s" % (frames, cores, dt)}


def main():
    args = parse_args()
    rank, local_rank, world = init_distributed(args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    from trackformer_amd import runtime
    runtime.configure_inference(tune=os.environ.get("TF_TUNE", "0") == "1",
                                miopen_find=os.environ.get("TF_MIOPEN_FIND", "1") == "1",
                                verbose=(rank == 0))

    from trackformer_amd import _cabi, fused
    _cabi.lib()   # fail loudly if the HIP library is missing
    if args.split_linear:
        fused.set_split_linear(True)

    import threading
    n_seq = max(1, args.sequences)
    steps_per_seq = [args.steps // n_seq + (1 if i < args.steps % n_seq else 0) for i in range(n_seq)]
    warm_per_seq = max(3, (args.warmup + n_seq - 1) // n_seq)   # >= 3: HIP-graph capture happens here

    # One tracker (own HIP stream, own graph buffers) per sequence; the detector weights are shared.
    trackers, model = [], None
    for i in range(n_seq):
        tracker, m = build_tracker(device, use_graph=not args.no_graph, model=model)
        model = m
        trackers.append(tracker)
    seeder = TrackSeeder(device, model.hidden_dim)
    frames = make_frames(device, host=args.host_frames)
    streams = [torch.cuda.Stream(device) for _ in range(n_seq)]

    stagger = [0.0]

    def run(seq, n):
        torch.cuda.set_device(device)
        with torch.no_grad(), torch.cuda.stream(streams[seq]):
            if stagger[0]:   # de-phase the sequences: one does host-side association while the
                time.sleep(seq * stagger[0])   # other's forward occupies the GPU
            for i in range(n):
                seeder.seed(trackers[seq])
                trackers[seq].step(frames[(seq + i) % len(frames)])
            streams[seq].synchronize()

    # warm-up sequentially (MIOpen find, caches, graph capture are not re-entrant), then time
    for seq in range(n_seq):
        run(seq, warm_per_seq)
    torch.cuda.synchronize()
    tw = time.perf_counter()
    run(0, 3)
    stagger[0] = (time.perf_counter() - tw) / 3 / n_seq if n_seq > 1 else 0.0
    torch.cuda.synchronize()
    barrier(world)
    t0 = time.perf_counter()
    if n_seq == 1:
        run(0, steps_per_seq[0])
    else:
        threads = [threading.Thread(target=run, args=(seq, steps_per_seq[seq]))
                   for seq in range(n_seq)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    torch.cuda.synchronize()
    barrier(world)
    elapsed = time.perf_counter() - t0

    from trackformer_amd import dist_utils as du
    elapsed = du.max_over_ranks(elapsed, device)

    roofline = cpu_baseline = None
    if rank == 0:
        if not args.no_roofline:
            roofline = measure_roofline(device)
        if world == 1 and not args.no_cpu_baseline:
            cpu_baseline = measure_cpu_baseline(args.cpu_frames)

    if world > 1:
        barrier(world)
        import torch.distributed as dist
        dist.destroy_process_group()

    if rank == 0:
        total_frames = args.steps * world
        value = total_frames / elapsed
        line = {
            "metric": "frames/sec, TrackFormer-Deformable inference (ResNet-50, 1333x800, 300 object + "
                      "100 track queries, bs 1 per GPU); roofline = ms_deform_attn HBM GB/s",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "per_gpu": round(value / world, 3),
            "config": {"workload": "BASELINE cfg 2: Tracker.step on 800x1333 synthetic frames, "
                                   "DeformableDETRTracking R50 4 levels, 300 obj + 100 track "
                                   "queries, seeded random-init weights, frames "
                                   + ("in pinned host memory (PCIe copy timed)" if args.host_frames
                                      else "resident in HBM"),
                       "global_batch": world, "parallelism": "sequence-sharded x%d" % world,
                       "sequences_per_gpu": n_seq, "hip_graph": not args.no_graph,
                       "linears": "bf16 split product (hi.hi + hi.mid + mid.hi, f32 accumulate)"
                                  if fused.split_linear_enabled() else "f32 (hipBLASLt)"},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line))


if __name__ == "__main__":
    main()
