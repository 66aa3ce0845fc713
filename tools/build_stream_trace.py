#!/usr/bin/env python
"""The phase-trace build of the stream GEMM (tools only -- libtf_msda.so never contains it):

    python tools/build_stream_trace.py

builds tools/bin/ablate/libtf_msda_stream_trace.so = the library with linear_stream.hip compiled with -DTF_STREAM_TRACE (see the macro
in trackformer_amd/csrc/linear_stream.hip).  tools/stream_trace.py loads it in place of libtf_msda.so (TF_MSDA_LIB) and prints the
per-slice phase durations."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from trackformer_amd import build as tfbuild  # noqa: E402


def main():
    tfbuild.build_all()
    out_dir = os.path.join(REPO, "tools", "bin", "ablate")
    os.makedirs(out_dir, exist_ok=True)
    obj_dir = os.path.join(tfbuild.LIB_DIR, "obj")
    others = [os.path.join(obj_dir, f) for f in sorted(os.listdir(obj_dir)) if f.endswith(".o") and f != "linear_stream.o"]
    hipcc = tfbuild._hipcc()
    flags = ["--offload-arch=" + tfbuild.GFX_ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + tfbuild.INCLUDE, "-Wno-pass-failed"]
    obj = os.path.join(out_dir, "linear_stream_trace.o")
    so = os.path.join(out_dir, "libtf_msda_stream_trace.so")
    subprocess.check_call([hipcc] + flags + ["-DTF_STREAM_TRACE=1", "-c", os.path.join(tfbuild.CSRC, "linear_stream.hip"), "-o", obj])
    subprocess.check_call([hipcc, "--offload-arch=" + tfbuild.GFX_ARCH, "-shared", "-fPIC", obj] + others + ["-o", so])
    os.remove(obj)
    print("built", os.path.relpath(so, REPO))


if __name__ == "__main__":
    main()
