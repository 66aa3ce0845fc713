#!/usr/bin/env python
"""Phase trace of the stream GEMM (trackformer_amd/csrc/linear_stream.hip built with -DTF_STREAM_TRACE by
tools/build_stream_trace.py): where a K-slice of a block spends its time.

    python tools/build_stream_trace.py && python tools/stream_trace.py      (on the GPU box)

Wave 0 of the first 64 blocks stamps s_memtime at three points of each slice -- where the LDS counter is drained anyway -- and keeps
the stamps of slices 4 .. 7 in registers until the block ends (version 2; version 1 stamped six points straight to memory and doubled
the launch time: profiles/r04_stream_phase_trace.txt); printed: the median over blocks and slices of each phase, in clock ticks and as
a share of the slice, for a
3 x 3 convolution of layer2 (128 -> 128 at 100 x 167), one of layer1 (64 -> 64 at 200 x 334) and the 1024 -> 256 linear at 22 223
rows; beside it the launch's duration by HIP events, which calibrates the tick."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["TF_MSDA_LIB"] = os.path.join(REPO, "tools", "bin", "ablate", "libtf_msda_stream_trace.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from trackformer_amd import _cabi, fused  # noqa: E402

PHASES = ["matrix phase: next weights requested, LDS fragments read, MFMAs issued", "staging: next slice split, LDS writes, global loads issued",
          "barrier"]


def trace(name, fn, iters=5):
    lib = _cabi.lib()
    lib.tf_debug_stream_trace_buffer.restype = ctypes.c_int
    lib.tf_debug_stream_trace_buffer.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3
    b, s, p = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.tf_debug_stream_trace_buffer(None, ctypes.byref(b), ctypes.byref(s), ctypes.byref(p))
    buf = torch.zeros(b.value * s.value * p.value, dtype=torch.int64, device="cuda:0")
    fn()                                              # weight image, warm-up
    torch.cuda.synchronize()
    assert lib.tf_debug_stream_trace_buffer(buf.data_ptr(), None, None, None) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    lib.tf_debug_stream_trace_buffer(None, None, None, None)
    us = e0.elapsed_time(e1) * 1e3 / iters
    t = buf.cpu().numpy().reshape(b.value, s.value, p.value).astype(np.float64)
    blocks = np.nonzero((t > 0).all(axis=(1, 2)))[0]  # blocks that ran all traced slices
    if len(blocks) == 0:
        print("== %s: %.1f us per launch; no block ran the traced slices (K too short)" % (name, us))
        return us, None
    # slice sl: points 0, 1, 2 and the start of slice sl + 1 (point 0 of the next row) -> three phases for all but the last traced slice
    d = []
    for bl in blocks:
        for sl in range(s.value - 1):
            d.append([t[bl, sl, 1] - t[bl, sl, 0], t[bl, sl, 2] - t[bl, sl, 1], t[bl, sl + 1, 0] - t[bl, sl, 2]])
    d = np.array(d)
    med = np.median(d, axis=0)
    print("== %s: %.1f us per launch (events, incl. launch gaps), %d traced blocks" % (name, us, len(blocks)))
    for ph, m in zip(PHASES, med):
        print("   %-75s %8.0f ticks  %5.1f %%" % (ph, m, 100 * m / med.sum()))
    print("   %-75s %8.0f ticks" % ("slice", med.sum()))
    return us, med


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    fused.set_conv_stream("all")
    for cin, cout, h, w in ((128, 128, 100, 167), (64, 64, 200, 334), (256, 256, 50, 84)):
        x = torch.randn(1, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).contiguous()
        b = torch.randn(cout, device=dev)
        trace("conv 3 x 3 %d -> %d at %d x %d" % (cin, cout, h, w), lambda: fused.conv3x3(x, wt, b, True, 1))
    x = torch.randn(22223, 1024, device=dev)
    wl = (torch.randn(256, 1024, device=dev) / 32).contiguous()
    bl = torch.randn(256, device=dev)
    trace("linear 22223 x 1024 -> 256", lambda: fused.linear(x, wl, bl))
    x2 = torch.randn(22223, 256, device=dev)
    w2 = (torch.randn(1024, 256, device=dev) / 16).contiguous()
    b2 = torch.randn(1024, device=dev)
    trace("linear 22223 x 256 -> 1024", lambda: fused.linear(x2, w2, b2))


if __name__ == "__main__":
    main()
