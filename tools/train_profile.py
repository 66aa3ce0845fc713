#!/usr/bin/env python
"""Where one cfg-3 training step (bs 2, 800 x 1333) spends its time: the step's stages timed with a device synchronisation at
every stage boundary (so a stage's figure is host + GPU time of that stage), the same step without the synchronisations, and a
cProfile of the host side.  Kernel-level: rocprofv3 --kernel-trace on tools/bench_train.py + tools/train_breakdown.py.

    python tools/train_profile.py [--steps 3] [--cprofile]
"""
import argparse
import collections
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tools.bench_train import synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cprofile", action="store_true")
    ap.add_argument("--per-step", type=int, default=0, help="print the stage times of this many single steps with their track-query counts")
    args = ap.parse_args()
    from trackformer_amd import config, deformable_detr, engine, factory
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    margs = config.make_args('deformable', 'tracking', 'mot17', device=str(dev))
    torch.manual_seed(42)
    model, criterion, _ = factory.build_model(margs)
    model.to(dev).train()
    criterion.train()
    optimizer, _ = engine.build_optimizer(model, margs)
    samples, targets = synthetic_batch(dev, 2, 800, 1333, seed=0)
    acc = collections.OrderedDict()
    sync_on = [True]

    def timed(name, fn):
        def wrapped(*a, **k):
            if not sync_on[0]:
                return fn(*a, **k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        return wrapped

    # stages: the detector forward (called twice per step: previous frame without gradients, then the frame itself), the matcher
    # (once for the previous frame inside the model, six times inside the criterion), the criterion, backward, clipping, AdamW
    base_forward = deformable_detr.DeformableDETR.forward
    calls = [0]

    def fwd(self, *a, **k):
        calls[0] += 1
        name = "forward, previous frame (no grad)" if not torch.is_grad_enabled() else "forward, this frame"
        return timed(name, base_forward)(self, *a, **k)
    deformable_detr.DeformableDETR.forward = fwd
    model._matcher.forward = timed("matcher (previous frame + 6 in the criterion)", model._matcher.forward)
    model.add_track_queries_to_targets = timed("add_track_queries_to_targets", model.add_track_queries_to_targets)
    crit_forward = criterion.forward
    criterion.forward = timed("criterion (incl. its matchers)", crit_forward)

    def step():
        tg = [dict(t, prev_target=dict(t['prev_target'])) for t in targets]
        outputs, tg2, *_ = model(samples, tg)
        loss_dict = criterion(outputs, tg2)
        losses = sum(loss_dict[k] * criterion.weight_dict[k] for k in loss_dict if k in criterion.weight_dict)
        timed("loss value to the host (finite check)", lambda: float(losses.detach()))()
        optimizer.zero_grad()
        timed("backward", losses.backward)()
        timed("clip_grad_norm_", lambda: torch.nn.utils.clip_grad_norm_(model.parameters(), margs.clip_max_norm))()
        timed("AdamW step", optimizer.step)()

    sync_on[0] = False
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    free = (time.perf_counter() - t0) / args.steps
    sync_on[0] = True
    acc.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    synced = (time.perf_counter() - t0) / args.steps
    print("# one cfg-3 training step (bs 2, 800x1333, fp32): %.1f ms free-running = %.1f images/s; %.1f ms with a synchronisation at every stage boundary" % (
        free * 1e3, 2 / free, synced * 1e3))
    tot = 0.0
    for k, v in acc.items():
        print("%8.2f ms  %s" % (v / args.steps * 1e3, k))
        if "matcher" not in k and "add_track" not in k:
            tot += v / args.steps
    print("%8.2f ms  sum of the stages (matcher / add_track_queries counted inside their callers)" % (tot * 1e3))
    for i in range(args.per_step):
        acc.clear()
        nq = []
        orig = model.add_track_queries_to_targets

        def spy(tg, *a, **k):
            r = orig(tg, *a, **k)
            nq.extend(int(t['track_query_hs_embeds'].shape[0]) for t in tg)
            return r
        model.add_track_queries_to_targets = spy
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        model.add_track_queries_to_targets = orig
        print("step %2d  %.1f ms  track queries %s  " % (i, (time.perf_counter() - t0) * 1e3, nq) + "  ".join("%s %.1f" % (k.split(",")[0].split(" (")[0][:22], v * 1e3) for k, v in acc.items()))
    if args.cprofile:
        import cProfile
        import pstats
        sync_on[0] = False
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
