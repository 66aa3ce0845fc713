#!/usr/bin/env python
"""Training-step timing for BASELINE cfg 3 (Deformable TrackFormer, bs 2 per GPU, 800x1333).

    python tools/bench_train.py [--steps 6] [--warmup 2] [--height 800 --width 1333]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/bench_train.py            # DDP over RCCL

One step = trackformer_amd.engine.train_step: forward of the previous frames without gradients,
Hungarian matching, track-query augmentation, forward, set-prediction loss (1 + 5 auxiliary
matchings), backward (incl. the MSDeformAttn backward kernels), gradient all-reduce (DDP), clipping,
AdamW.  Synthetic data as SURVEY.md section 8d describes it (30 boxes per image, previous frame =
image + noise, previous boxes jittered by 1 %).  Prints one JSON line: images/s over all ranks.
"""
import argparse
import json
import os
import sys
import time

# RCCL / cross-process device memory on this stack need dmabuf IPC (see the environment notes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def synthetic_batch(device, batch, h, w, boxes_per_image=30, seed=0):
    g = torch.Generator().manual_seed(seed)
    samples, targets = [], []
    for i in range(batch):
        img = torch.randn(3, h, w, generator=g)
        cxcy = torch.rand(boxes_per_image, 2, generator=g) * 0.8 + 0.1
        wh = torch.rand(boxes_per_image, 2, generator=g) * 0.15 + 0.03
        boxes = torch.cat([cxcy, wh], 1)
        prev_boxes = boxes * (1 + 0.01 * (torch.rand(boxes.shape, generator=g) * 2 - 1))
        ids = torch.arange(boxes_per_image)
        labels = torch.zeros(boxes_per_image, dtype=torch.long)
        size = torch.tensor([h, w])
        t = {'boxes': boxes, 'labels': labels, 'track_ids': ids, 'image_id': torch.tensor([i]),
             'size': size, 'orig_size': size,
             'prev_image': img + 0.01 * torch.randn(3, h, w, generator=g),
             'prev_target': {'boxes': prev_boxes, 'labels': labels.clone(), 'track_ids': ids.clone(),
                             'image_id': torch.tensor([i]), 'size': size, 'orig_size': size}}

        def dev(x):
            return {k: dev(v) for k, v in x.items()} if isinstance(x, dict) else x.to(device)
        samples.append(img.to(device))
        targets.append(dev(t))
    return samples, targets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    args = ap.parse_args()

    from trackformer_amd import config, dist_utils as du, engine, factory
    rank, local_rank, world = du.env_world()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        du.init_from_env(backend="nccl", device=device)
    torch.backends.cudnn.benchmark = True

    margs = config.make_args('deformable', 'tracking', 'mot17', device=str(device))
    torch.manual_seed(42)
    model, criterion, _ = factory.build_model(margs)
    model.to(device)
    model.train()
    criterion.train()
    optimizer, _ = engine.build_optimizer(model, margs)
    ddp = engine.wrap_ddp(model, device)
    samples, targets = synthetic_batch(device, args.batch, args.height, args.width, seed=rank)

    def step():
        import copy
        tg = [dict(t, prev_target=dict(t['prev_target'])) for t in targets]   # forward mutates targets
        return engine.train_step(ddp, criterion, optimizer, samples, tg,
                                 clip_max_norm=margs.clip_max_norm)

    for _ in range(args.warmup):
        step()
    engine.settle_heap()
    du.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    torch.cuda.synchronize()
    du.barrier()
    elapsed = du.max_over_ranks(time.perf_counter() - t0, device=device)
    if rank == 0:
        print(json.dumps({
            "metric": "images/sec, Deformable TrackFormer training step (fwd prev frame + match + fwd + "
                      "loss + bwd + all-reduce + AdamW), bs %d per GPU, %dx%d" % (args.batch, args.width, args.height),
            "value": round(args.steps * args.batch * world / elapsed, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "dtype": "f32", "data": "synthetic",
            "last_loss": round(float(loss), 4)}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
