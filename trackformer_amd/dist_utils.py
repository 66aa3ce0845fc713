"""Multi-GPU plumbing: one process per GPU, sequences sharded across ranks, no data-path collective.

Mirrors how the reference distributes tracking evaluation (engine.py:289-303: sequence i goes to rank
i % world_size; results are merged with a pickled all_gather, util/misc.py:91-132) and provides the
barrier / max-over-ranks timing helpers bench.py uses.  Backend "nccl" is RCCL on ROCm; the same code
runs on "gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(backend=None, device=None):
    """Initialise the default process group when WORLD_SIZE > 1 (env:// rendezvous on 127.0.0.1)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def barrier():
    if is_distributed():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    """The slowest rank's value (what the whole job waits for)."""
    if not is_distributed():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not is_distributed():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shard_sequences(sequences, rank=None, world=None):
    """Round-robin partition of independent sequences: sequence i -> rank i % world."""
    if rank is None or world is None:
        rank = dist.get_rank() if is_distributed() else 0
        world = dist.get_world_size() if is_distributed() else 1
    return [s for i, s in enumerate(sequences) if i % world == rank]


def gather_results(local_results):
    """Every rank gets the list of all ranks' (picklable) results, in rank order."""
    if not is_distributed():
        return [local_results]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local_results)
    return out


def track_sequences(make_tracker, sequences, device):
    """Track this rank's share of `sequences` (each an iterable of blobs) and merge the per-sequence
    results of all ranks: {sequence index: tracker results}.  No collective inside the loop."""
    rank = dist.get_rank() if is_distributed() else 0
    world = dist.get_world_size() if is_distributed() else 1
    tracker = make_tracker(device)
    local = {}
    for idx, seq in enumerate(sequences):
        if idx % world != rank:
            continue
        tracker.reset()
        with torch.no_grad():
            for blob in seq:
                tracker.step(blob)
        local[idx] = tracker.get_results()
    merged = {}
    for part in gather_results(local):
        merged.update(part)
    return merged
