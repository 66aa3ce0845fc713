"""Multi-GPU plumbing: one process per GPU, sequences sharded across ranks, no data-path collective.

Mirrors how the reference distributes tracking evaluation (engine.py:289-303: sequence i goes to rank
i % world_size; results are merged with a pickled all_gather, util/misc.py:91-132) and provides the
barrier / max-over-ranks timing helpers bench.py uses.  Backend "nccl" is RCCL on ROCm; the same code
runs on "gloo" for the CPU tests.
"""
import os

from collections import deque

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


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def _gpu_numa_node(device_index):
    """NUMA node the GPU hangs off (sysfs), or None when the platform does not say."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read())
        return node if node >= 0 else None
    except (OSError, AttributeError, ValueError, RuntimeError, AssertionError):
        return None


def _node_cpus(node):
    try:
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            return _parse_cpulist(f.read())
    except (OSError, ValueError):
        return None


def _cpu_share(local_rank, local_world, allowed, node_of):
    """The CPU slice of `local_rank`.  node_of: NUMA node of every local rank's GPU (None where unknown).  Every rank computes the
    same table from sysfs, so the ranks agree without a collective: when ALL nodes are known AND every node in use has at least
    one allowed CPU per rank on it, the ranks whose GPUs share a node split that node's CPUs by their order among themselves
    (whatever the GPU -> node mapping looks like: contiguous, interleaved, uneven).  Otherwise -- a node unknown, or a node
    whose allowed CPUs do not go round (a container cpuset that covers one node only) -- EVERY rank takes the even split of
    the allowed CPUs by local rank: the decision is made on the whole table, never per rank, so slices cannot overlap."""
    if all(n is not None for n in node_of):
        pools, ok = {}, True
        for node in sorted(set(node_of)):
            cpus = _node_cpus(node)
            pools[node] = sorted(set(allowed) & cpus) if cpus else []
            ok = ok and len(pools[node]) >= sum(1 for n in node_of if n == node)
        if ok:
            mine = node_of[local_rank]
            pool = pools[mine]
            peers = [r for r in range(local_world) if node_of[r] == mine]
            k = peers.index(local_rank)
            return pool[len(pool) * k // len(peers):len(pool) * (k + 1) // len(peers)]
    return allowed[len(allowed) * local_rank // local_world:len(allowed) * (local_rank + 1) // local_world]


def pin_rank_to_cpus(local_rank, local_world, max_threads=16):
    """One process per GPU on one host: give every rank its own slice of the host cores, on its GPU's NUMA node where
    sysfs knows it -- at 8 ranks x (tracker threads + MIOpen find + the matcher's scipy) an unpinned host oversubscribes
    its cores and the slowest rank sets the job's time.  Caps torch's intra-op threads to the slice.  Returns the CPU list
    (empty: the platform has no sched_setaffinity, nothing was changed)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return []
    allowed = sorted(os.sched_getaffinity(0))
    have = torch.cuda.is_available() and torch.cuda.device_count() >= local_world
    node_of = [_gpu_numa_node(r) if have else None for r in range(local_world)]
    share = _cpu_share(local_rank, local_world, allowed, node_of)
    if not share:
        return []
    # (call this before anything starts a thread pool: sched_setaffinity(0) moves the calling thread, and the threads created
    # afterwards inherit its mask -- bench.py and engine.py call it right after reading the rank environment)
    os.sched_setaffinity(0, share)
    torch.set_num_threads(max(1, min(max_threads, len(share))))
    return share


def ranks_seen(device=None):
    """What every rank of the job runs on, gathered over the job's own backend (RCCL for "nccl"): the N > 1 bench line
    carries it so that a scaling run verifies itself (N distinct devices, one process each)."""
    info = {"rank": dist.get_rank() if is_distributed() else 0, "pid": os.getpid(),
            "backend": dist.get_backend() if is_distributed() else None, "cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    if device is not None and torch.cuda.is_available() and torch.device(device).type == "cuda":
        props = torch.cuda.get_device_properties(device)
        info.update(device=str(device), name=props.name, uuid=str(getattr(props, "uuid", "")),
                    pci_bus_id=getattr(props, "pci_bus_id", None))
    return gather_results(info)


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


# streams of interleaved lanes, as indices into PyTorch's normal-priority stream pool (runtime.pool_stream): lane k's tracker runs on
# stream LANE_MAINS[k], its GraphedDetector's side stream is LANE_SIDES[k] (TF_LANE_MAINS / TF_LANE_SIDES: comma-separated, A/B aid)
# (a second side stream: + 16).  Three lanes of cfg 2 on MI355X: 424 frames/s with this layout, the best of 13 under the binding order
# of runtime.bind_streams (334 - 424, tools/gpu_runs/gpu_r06_61.sh; under the interleaved order: 48 layouts, 232 - 392, gpu_r06_56.sh).
LANE_MAINS = tuple(int(x) for x in os.environ.get("TF_LANE_MAINS", "5,4,3,0,1,2,6,7").split(","))
LANE_SIDES = tuple(int(x) for x in os.environ.get("TF_LANE_SIDES", "1,2,6,10,12,9,13,8").split(","))
# ... and the single side stream of a NARROW wrapper (mask-head / multi-frame models; TF_LANE_SIDES_NARROW)
LANE_SIDES_NARROW = tuple(int(x) for x in os.environ.get("TF_LANE_SIDES_NARROW", "1,14,6,10,12,9,13,8").split(","))
# ... and the lanes' own streams for a multi-frame model (no look-ahead with several lanes: only these streams matter; cfg 4, three
# lanes: 144 - 216 frames/s over 22 layouts, tools/gpu_runs/gpu_r06_58.sh; a mask-head model does best on LANE_MAINS: cfg 5 124.7)
LANE_MAINS_NARROW = tuple(int(x) for x in os.environ.get("TF_LANE_MAINS_NARROW", "4,3,10,8,9,0,1,2").split(","))


def sequence_stream(device, lanes=1, lane=0, narrow=False):
    """A stream for one sequence's tracker.  ONE sequence in flight: HIGH priority (TF_SEQ_STREAM_PRIORITY overrides).  What runs on
    it -- the decoder half of a frame, the post-processing, the result rows' way to the host: ~150 launches of a few microseconds
    that the host waits for before it can associate -- competes with the image-only halves of the coming frames, which
    GraphedDetector runs on normal-priority side streams and which nobody waits for yet.  cfg 2 on MI355X with two frames of
    look-ahead: 349.6 -> 374.7 frames/s (tools/gpu_runs/gpu_r06_48.sh).  SEVERAL sequences interleaved: normal priority -- every
    lane's decoder half in front of every lane's image-only half starves the latter (three lanes: 426 frames/s normal, 253 high;
    cfg 5: 130 / 102).  The streams are fixed members of PyTorch's pool (runtime.pool_stream: which streams decides the rate)."""
    from .runtime import bind_streams, placement_tuned, pool_stream
    if not placement_tuned():   # (another number of hardware queues than the tables were measured with: round 5's way)
        return torch.cuda.Stream(device, priority=int(os.environ.get("TF_SEQ_STREAM_PRIORITY", "0")))
    bind_streams(device)
    default = "-1" if lanes == 1 else "0"
    priority = int(os.environ.get("TF_SEQ_STREAM_PRIORITY", default))
    if lanes == 1:
        # stream 0 of the high-priority pool: the placement GraphedDetector's side streams were measured against
        return pool_stream(device, int(os.environ.get("TF_SEQ_STREAM_INDEX", "0")), priority)
    table = LANE_MAINS_NARROW if narrow else LANE_MAINS   # (narrow: a multi-frame model)
    return pool_stream(device, table[lane % len(table)], priority)


def track_sequences(make_tracker, sequences, device, interleave=1):
    """Track this rank's share of `sequences` (each an iterable of blobs) and merge the per-sequence
    results of all ranks: {sequence index: tracker results}.  No collective inside the loop.
    interleave > 1: that many of the rank's sequences are in flight at once, in ONE thread -- each on its own tracker
    (make_tracker is called `interleave` times; the trackers may share the detector's weights) and, on a GPU, its own
    stream: Tracker.step_async enqueues a frame's forward and returns, step_finish runs its association, so one sequence's
    host work overlaps another's GPU work (cfg 2 on MI355X with ~100 live tracks: 149 -> 263 frames/s at 3).  Frames
    of one sequence stay strictly sequential and the results are those of interleave = 1.
    Inside a sequence the lane looks ahead (Tracker.look_ahead frames): Tracker.step_prepare(blob) enqueues the image-only half of
    the coming frames' forward before step_finish associates the frame in flight (round 5: one sequence 236 -> 351 frames/s; a no-op for
    the models it does not apply to)."""
    rank = dist.get_rank() if is_distributed() else 0
    world = dist.get_world_size() if is_distributed() else 1
    mine = [(idx, seq) for idx, seq in enumerate(sequences) if idx % world == rank]
    lanes = max(1, min(int(interleave), len(mine) or 1))
    trackers = [make_tracker(device) for _ in range(lanes)]
    for k, t in enumerate(trackers):   # (GraphedDetector: the schedule of the prepared image-only halves depends on the number of lanes)
        set_lanes = getattr(getattr(t, "obj_detector", None), "set_lanes", None)
        if set_lanes is not None and type(t.obj_detector).__name__ == "GraphedDetector":
            set_lanes(lanes, k)
    on_gpu = torch.cuda.is_available() and torch.device(device).type == "cuda"
    narrow = bool(getattr(getattr(trackers[0], "obj_detector", None), "multi_frame_attention", False))
    streams = [sequence_stream(device, lanes, k, narrow) for k in range(lanes)] if on_gpu else [None] * lanes
    local = {}
    todo = iter(mine)
    lane_seq = [None] * lanes      # (sequence index, frame iterator) of the lane
    pending = [None] * lanes       # handle of the frame in flight
    # blobs of the lane's sequence already taken from its iterator, oldest first: [0 .. prepared[k]) have had step_prepare
    ahead = [deque() for _ in range(lanes)]
    prepared = [0] * lanes
    # look-ahead policy (measured on MI355X, profiles/r06_bench_cfg4_*.json): with several sequences in flight the other lanes
    # already fill the GPU while one associates; for the multi-frame models -- whose image-only half is two encoder passes -- a
    # second stream per lane then costs more than it gains (3 lanes: 191 frames/s without, 170 with; 1 lane: 111 -> 151 with)
    look_ahead = lanes == 1 or not getattr(getattr(trackers[0], "obj_detector", None), "multi_frame_attention", False)
    # frames ahead (Tracker.look_ahead: 2 with HIP graphs and a single-frame model -- the image-only halves of frames t + 1 and
    # t + 2 share the chip, round 6 --, otherwise 1)
    depth = int(getattr(trackers[0], "look_ahead", 1)) if look_ahead and hasattr(trackers[0], "step_prepare") else 0

    def advance(k):
        """Finish the lane's frame in flight, then launch its next frame (of the same or, at its end, the next sequence)."""
        if pending[k] is not None:
            if lane_seq[k] is not None:
                while len(ahead[k]) < depth:
                    blob = next(lane_seq[k][1], None)
                    if blob is None:
                        break
                    ahead[k].append(blob)
                while prepared[k] < len(ahead[k]):
                    # GPU: the next frames' backbone + encoder; host: this frame's association
                    if not trackers[k].step_prepare(ahead[k][prepared[k]]):
                        break
                    prepared[k] += 1
            trackers[k].step_finish(pending[k])
            pending[k] = None
        while True:
            if lane_seq[k] is None:
                nxt = next(todo, None)
                if nxt is None:
                    return False
                trackers[k].reset()
                lane_seq[k] = (nxt[0], iter(nxt[1]))
                ahead[k].clear()
                prepared[k] = 0
            if ahead[k]:
                blob = ahead[k].popleft()
                prepared[k] = max(0, prepared[k] - 1)
            else:
                blob = next(lane_seq[k][1], None)
            if blob is not None:
                pending[k] = trackers[k].step_async(blob)
                return True
            local[lane_seq[k][0]] = trackers[k].get_results()
            lane_seq[k] = None

    if on_gpu:
        # frames the caller already put on the device were produced on ITS stream: the lanes' streams start behind that work
        # (frames produced while iterating are enqueued on the lane's stream: the iterators are advanced under it)
        caller = torch.cuda.current_stream(device)
        for st in streams:
            st.wait_stream(caller)
    with torch.no_grad():
        busy = [True] * lanes
        while any(busy):
            for k in range(lanes):
                if not busy[k]:
                    continue
                if streams[k] is not None:
                    with torch.cuda.stream(streams[k]):
                        busy[k] = advance(k)
                else:
                    busy[k] = advance(k)
    if on_gpu:
        for st in streams:   # (and whatever the caller does next on its stream comes after the lanes' work)
            caller.wait_stream(st)
    merged = {}
    for part in gather_results(local):
        merged.update(part)
    return merged
