"""Training step and optimiser set-up of the reference, without its sacred / visdom scaffolding.

`build_optimizer` restates src/train.py:93-120 (three AdamW parameter groups: default, backbone,
linear projections of the sampling offsets / reference points; MultiStepLR at `lr_drop`);
`train_step` is the loop body of engine.py:119-158 (forward -- which itself runs the previous frame
and adds the track queries, detr_tracking.py:219-277 -- criterion, weighted loss sum, backward,
gradient clipping, optimiser step).  Data-parallel training wraps the model in
torch.nn.parallel.DistributedDataParallel(find_unused_parameters=True) as train.py:87-88 does; the
gradient all-reduce then runs over RCCL (backend "nccl" on ROCm).
"""
import math

import torch

from . import dist_utils


def _matches(name, keywords):
    return any(k in name for k in keywords)


def build_optimizer(model, args):
    """-> (AdamW, MultiStepLR) with the reference's per-group learning rates."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    special = list(args.lr_backbone_names) + list(args.lr_linear_proj_names) + ['layers_track_attention']
    groups = [
        {"params": [p for n, p in named if not _matches(n, special)], "lr": args.lr},
        {"params": [p for n, p in named if _matches(n, args.lr_backbone_names)],
         "lr": args.lr_backbone},
        {"params": [p for n, p in named if _matches(n, args.lr_linear_proj_names)],
         "lr": args.lr * args.lr_linear_proj_mult},
    ]
    if getattr(args, "track_attention", False):
        groups.append({"params": [p for n, p in named if _matches(n, ['layers_track_attention'])],
                       "lr": args.lr_track})
    # (the same update rule as the reference's torch.optim.AdamW; on the GPU as ONE multi-tensor launch per parameter group and
    # state tensor instead of a Python loop over ~340 parameters)
    on_gpu = all(p.is_cuda for g in groups for p in g["params"])
    optimizer = torch.optim.AdamW(groups, lr=args.lr, weight_decay=args.weight_decay, **({"fused": True} if on_gpu else {}))
    scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, [args.lr_drop])
    return optimizer, scheduler


def wrap_ddp(model, device):
    """DistributedDataParallel when torch.distributed is initialised, the bare model otherwise."""
    if not dist_utils.is_distributed():
        return model
    ids = [device.index] if device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids,
                                                     find_unused_parameters=True)


def train_step(model, criterion, optimizer, samples, targets, clip_max_norm=0.1,
               check_finite=True):
    """One optimisation step; returns (weighted loss as a 0-dim tensor, loss_dict).

    `samples`: NestedTensor / list of [3,H,W] images on the model's device; `targets`: list of dicts
    with boxes, labels, track_ids, prev_image, prev_target (tracking), already on the device."""
    outputs, targets, *_ = model(samples, targets)
    loss_dict = criterion(outputs, targets)
    weight_dict = criterion.weight_dict
    losses = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)
    if check_finite and not math.isfinite(float(losses.detach())):   # engine.py:143-146 (one host sync)
        raise FloatingPointError("loss is %r: %r" % (float(losses.detach()),
                                                     {k: float(v) for k, v in loss_dict.items()}))
    optimizer.zero_grad()
    losses.backward()
    if clip_max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_max_norm)
    optimizer.step()
    return losses.detach(), loss_dict


def settle_heap():
    """Call after the first training steps (model, optimiser state and the libraries' caches exist).  A step creates tens of thousands
    of short-lived Python objects (autograd nodes, target dicts); CPython's cyclic collector then runs its oldest-generation pass every
    few steps, and that pass walks EVERY tracked object of the process -- ~300 000 with the model, the optimiser and torch in memory:
    ~80 ms, inside whatever stage happens to allocate (measured on MI355X: a 100 ms step becomes 185 ms in 9 of 24 steps,
    profiles/r06_train_step_breakdown.txt).  gc.freeze() moves what is alive now into the permanent generation."""
    from . import runtime
    runtime.settle_heap()


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch=0, clip_max_norm=0.1, settle_after=2, log=None):
    """The loop of engine.py:119-158 of the reference around train_step: moves each batch to `device`, steps, and returns the mean of
    every loss over the epoch.  `settle_after`: steps after which settle_heap() runs (0: never)."""
    model.train()
    criterion.train()
    sums, n = {}, 0

    def to_dev(x):
        if isinstance(x, dict):
            return {k: to_dev(v) for k, v in x.items()}
        return x.to(device) if torch.is_tensor(x) else x
    for i, (samples, targets) in enumerate(data_loader):
        samples = samples.to(device) if hasattr(samples, "to") else [s.to(device) for s in samples]
        targets = [to_dev(t) for t in targets]
        loss, loss_dict = train_step(model, criterion, optimizer, samples, targets, clip_max_norm=clip_max_norm)
        for k, v in loss_dict.items():
            sums[k] = sums.get(k, 0.0) + float(v)
        sums["loss"] = sums.get("loss", 0.0) + float(loss)
        n += 1
        if settle_after and i + 1 == settle_after and epoch == 0:
            settle_heap()
        if log is not None:
            log(epoch, i, float(loss))
    return {k: v / max(n, 1) for k, v in sums.items()}
