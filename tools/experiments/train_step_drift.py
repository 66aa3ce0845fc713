#!/usr/bin/env python
"""cfg 3: per-step wall time over 24 steps (does the step time drift?), with the cyclic collector on / off and allocator statistics."""
import gc
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tools.bench_train import synthetic_batch  # noqa: E402
from trackformer_amd import config, engine, factory  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
margs = config.make_args('deformable', 'tracking', 'mot17', device=str(dev))
torch.manual_seed(42)
model, criterion, _ = factory.build_model(margs)
model.to(dev).train()
criterion.train()
optimizer, _ = engine.build_optimizer(model, margs)
samples, targets = synthetic_batch(dev, 2, 800, 1333, seed=0)
if "--nogc" in sys.argv:
    gc.disable()
if "--lr0" in sys.argv:
    for g in optimizer.param_groups:
        g["lr"] = 0.0
        g["weight_decay"] = 0.0
for i in range(24):
    tg = [dict(t, prev_target=dict(t['prev_target'])) for t in targets]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, _ = engine.train_step(model, criterion, optimizer, samples, tg, clip_max_norm=margs.clip_max_norm)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nq = [int(t['track_query_hs_embeds'].shape[0]) for t in tg]
    if i == 1 and "--settle" in sys.argv:
        engine.settle_heap()
    print("step %2d  %.1f ms  loss %.3f  track queries %s  alloc %.2f GB reserved %.2f GB  gc objects %d" % (
        i, dt * 1e3, float(loss), nq, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, len(gc.get_objects())))
