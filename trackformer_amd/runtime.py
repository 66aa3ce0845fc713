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

