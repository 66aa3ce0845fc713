"""TEST INFRASTRUCTURE: runs the package's GPU inference path -- the Python glue around every HIP kernel (fused MSDeformAttn
entry, split-product linears, bias_act, residual + LayerNorm, the query self-attention kernel, the opt-in convolution /
GroupNorm / box-refinement routes) -- on CPU tensors, with the SIMT emulator's build of the kernel sources
(tests/emu/) loaded in place of libtf_msda.so.  Host pointers stand in for device pointers; `Tensor.is_cuda` and the few
torch.cuda calls the glue makes are patched for the duration of the context.  This is how the host side of the opt-in
routes (views, layouts, weight caches) was checked end to end against the reference goldens before any of it ran on
hardware.  Nothing under trackformer_amd/ knows about it."""
import contextlib
import os

import torch

from tests.emu import build_emu


class _NullDevice:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Stream:
    cuda_stream = 0

    def synchronize(self):   # the emulator runs every launch to completion before returning
        pass


@contextlib.contextmanager
def gpu_path_on_emulator(host_threads=8):
    from trackformer_amd import _cabi
    so = build_emu.build()
    if so is None:
        raise RuntimeError("no host clang++ to build the emulated library with")
    os.environ.setdefault("HIPEMU_POISON", "1")
    os.environ["HIPEMU_THREADS"] = str(host_threads)     # read at every launch: workgroups spread over host threads
    saved = (_cabi.LIB_PATH, _cabi._lib, torch.cuda.device, torch.cuda.current_stream,
             torch.cuda.is_current_stream_capturing)
    had_prop = "is_cuda" in torch.Tensor.__dict__
    old_prop = torch.Tensor.__dict__.get("is_cuda")
    try:
        _cabi.LIB_PATH, _cabi._lib = so, None
        lib = _cabi.lib()                                # same prototypes as for the real library
        calls = {}
        for name in _cabi.EXPORTED_SYMBOLS:              # count the calls per entry point: tests assert which routes ran
            fn = getattr(lib, name)
            if not name.startswith("tf_") or name in ("tf_msda_strerror",):
                continue

            def counted(*a, _fn=fn, _name=name):
                calls[_name] = calls.get(_name, 0) + 1
                return _fn(*a)
            setattr(lib, name, counted)
        lib.calls = calls
        torch.cuda.device = _NullDevice
        torch.cuda.current_stream = lambda *a, **k: _Stream()
        torch.cuda.is_current_stream_capturing = lambda: False
        torch.Tensor.is_cuda = property(lambda self: True)
        yield _cabi._lib
    finally:
        if had_prop:
            torch.Tensor.is_cuda = old_prop
        else:
            try:
                del torch.Tensor.is_cuda
            except AttributeError:
                pass
        (_cabi.LIB_PATH, _cabi._lib, torch.cuda.device, torch.cuda.current_stream,
         torch.cuda.is_current_stream_capturing) = saved
        os.environ.pop("HIPEMU_THREADS", None)
