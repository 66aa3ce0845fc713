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
    optimizer = torch.optim.AdamW(groups, lr=args.lr, weight_decay=args.weight_decay)
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
