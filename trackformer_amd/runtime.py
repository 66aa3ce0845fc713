"""Process-wide inference setup for MI355X: library-kernel selection for the dense layers.

The dense part of the path (ResNet-50 convolutions, the encoder/decoder linears) runs on MIOpen and
hipBLASLt/rocBLAS through PyTorch-ROCm.  Their default heuristics leave a lot on the table at this
model's shapes (fp32, M = 22 223 tokens, K,N in {128, 256, 1024}; batch-1 NHWC convolutions):

  * `torch.backends.cudnn.benchmark = True` lets MIOpen time its solvers per convolution
    configuration on first use instead of taking the immediate-mode guess (13.0 -> 11.0 ms/frame);
  * PyTorch TunableOp picks the fastest hipBLASLt / rocBLAS solution per GEMM shape (53.6 -> 32 us for
    the 22223x256x256 projections, 122 TFLOP/s fp32 on the FFN GEMMs; 13.0 -> 10.7 ms/frame).  The
    selections for the BASELINE cfg-2 shapes, tuned on an MI355X with this image, ship in
    trackformer_amd/tuning/ and are only looked up (no tuning at run time) unless `tune=True`.

Numerics: every candidate is an fp32 kernel of the same library; results differ by summation order
only (parity tests run with this setup enabled).
"""
import os
import sys

import torch

_TUNING_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning")
_configured = False


def configure_inference(tune=False, miopen_find=True, tunable_file=None, verbose=False, host_threads=4):
    """Idempotent.  Call once per process before the first forward.
    host_threads: cap of torch's intra-op CPU threads (None: leave it).  The association leg of Tracker.step works on host
    tensors of a few hundred boxes (NMS masks, IoU matrices, the packed copy); with torch's default of one thread per
    core a 250 x 250 `triu` pays the fork / join of 128 threads: measured on the MI355X box with ~100 live tracks and
    ~150 detections per frame, 52 ms per step at the default against the GPU forward's 3 ms (profiles/r03_host_profile_*)."""
    global _configured
    if _configured:
        return
    _configured = True
    if host_threads is not None and torch.get_num_threads() > host_threads:
        torch.set_num_threads(int(host_threads))
    if not torch.cuda.is_available():
        return
    if miopen_find:
        torch.backends.cudnn.benchmark = True
    try:
        import torch.cuda.tunable as tunable
    except ImportError:
        return
    path = tunable_file or os.path.join(_TUNING_DIR, "tunableop_gfx950_cfg2.csv")
    tunable.enable(True)
    tunable.tuning_enable(bool(tune))
    if tune:
        tunable.set_max_tuning_duration(30)
        tunable.set_max_tuning_iterations(100)
        out = os.environ.get("TF_TUNABLEOP_OUT")
        if out:
            tunable.set_filename(out, insert_device_ordinal=True)
    elif hasattr(tunable, "write_file_on_exit"):
        tunable.write_file_on_exit(False)
    else:   # this PyTorch writes its (unchanged) table at exit: keep it out of the working directory
        tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"),
                                          "tf_tunableop_results.csv"), insert_device_ordinal=True)
    if os.path.exists(path):
        ok = tunable.read_file(path)
        if verbose:
            print("trackformer_amd: TunableOp selections %s from %s" %
                  ("loaded" if ok else "REJECTED (validator mismatch)", path), file=sys.stderr)


def configure_training():
    """The training step (engine.train_step) runs its linears in hipBLASLt / rocBLAS and its convolutions in MIOpen, forward and
    backward, in fp32: MIOpen times its solvers per configuration on first use (as for inference).  PyTorch TunableOp is NOT
    switched on here: tuned on an MI355X for the GEMM shapes of the BASELINE cfg-3 step (tools/gpu_runs/gpu_r06_34.sh) it picked
    solutions that are slower inside the step than hipBLASLt's own heuristics -- 95.7 against 84.8 ms per step (the 44 446 x 256 ->
    1024 forward GEMM: 190 us for the tuner's choice, 152 us for the default's)."""
    if torch.cuda.is_available():
        torch.backends.cudnn.benchmark = True


def settle_heap():
    """Call once the model, the trackers and the HIP graphs exist (after the warm-up frames).  The association leg creates and
    drops a few hundred small objects per frame (Track, deque, per-row tensors), which makes CPython's cyclic collector run
    every few frames; its older-generation passes walk EVERY tracked object of the process -- with a model in memory that is
    the whole module / parameter graph, for nothing.  gc.freeze() moves what is alive now into the permanent generation:
    later collections only look at what the frames themselves allocate.  (tools/profile_host_cpu.py --gc: the collector is
    ~7 % of the association leg even without a model in the process.)"""
    import gc
    gc.collect()
    gc.freeze()



def placement_tuned():
    """The stream placement of round 6 (pool_stream / bind_streams, dist_utils.sequence_stream, GraphedDetector's WIDE schedule) was
    measured with the HIP runtime's default of 4 hardware queues per priority; with another GPU_MAX_HW_QUEUES the same tables read
    155 - 292 frames/s instead of 375 (profiles/r06_stream_queue_map.txt, section 14).  False then: the callers keep round 5's
    schedule (one side stream, the pool's next streams, normal priority: 349 frames/s wherever the streams sit)."""
    return os.environ.get("GPU_MAX_HW_QUEUES", "4").strip() == "4"


_BOUND = set()


def bind_streams(device):
    """Touch every stream of PyTorch's two pools for `device` once, in a fixed order (normal 0 ... 31, then high 0 ... 31).
    The HIP runtime binds a stream to one of its hardware queues when the stream first does something, and which streams end up
    together decides the rate of the pipelined frame loop (pool_stream): with the binding left to whoever uses a stream first, three
    interleaved lanes on the SAME streams ran at 427 frames/s in a process that started with them and at 294 in one that had
    tracked a single sequence before (bench.py's legs; tools/gpu_runs/gpu_r06_54.sh).  Once per device and process; a few hundred
    microseconds."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if device in _BOUND:
        return
    _BOUND.add(device)
    scratch = torch.zeros(64, device=device)
    torch.cuda.synchronize(device)
    # the order matters too (tools/gpu_runs/gpu_r06_61.sh): all normal-priority streams first, then the high-priority ones -- a single
    # sequence reads the same 375 frames/s as with the two pools interleaved, three lanes reach 424 instead of 392; high-priority
    # streams first: a single sequence drops to 235; the interleaved order backwards: 350 / <= 323
    order = os.environ.get("TF_BIND_ORDER", "normal_first")
    pairs = [(i, priority) for i in range(32) for priority in (0, -1)]
    if order == "normal_first":
        pairs = [(i, 0) for i in range(32)] + [(i, -1) for i in range(32)]
    elif order == "high_first":
        pairs = [(i, -1) for i in range(32)] + [(i, 0) for i in range(32)]
    elif order == "reverse":
        pairs = pairs[::-1]
    for i, priority in pairs:
        with torch.cuda.stream(pool_stream(device, i, priority)):
            scratch[2 * i + (priority != 0)].zero_()
    torch.cuda.synchronize(device)


def pool_stream(device, index, priority=0):
    """THE stream number `index` of PyTorch's stream pool for `device` and `priority` (0 normal, -1 high; 32 streams per pool, all
    created together at the device's first stream request and handed out round-robin; StreamId = index << 5 | type).

    Why pick by index (round 6, MI355X, ROCm 7.2): which hardware queue a HIP stream sits on is fixed when the stream is created,
    and the rate of the pipelined frame loop -- the decoder half of frame t on the sequence's stream next to the image-only halves
    of frames t + 1 and t + 2 on two side streams -- depends on WHICH streams: same schedule, same kernels, 378 frames/s with the
    sequence on high-priority stream 0 and the side streams on normal streams (1, 5); 234 with the side streams on (2, 6); 233 with
    the sequence on high-priority stream 1 (tools/experiments/stream_queue_map.py, profiles/r06_stream_queue_map.txt).  Taking
    "the next stream of the pool" made every leg of bench.py depend on how many streams the process had drawn before it
    (284 / 308 / 331 / 368 frames/s for one schedule, tools/gpu_runs/gpu_r06_44.sh).  Falls back to the pool's next stream when the
    index never shows up (another StreamId layout)."""
    device = torch.device(device)
    stream = None
    for _ in range(64):
        stream = torch.cuda.Stream(device, priority=priority)
        if (int(stream.stream_id) >> 5) == index:
            break
    return stream
