#!/usr/bin/env python
"""Builds a timing VARIANT of the library for tools/bin/msda_bench (tools only -- libtf_msda.so never contains it):

    python tools/build_variant.py <tag> [--source msda_hip.hip] -DNAME=VALUE [-DNAME=VALUE ...]

-> tools/bin/ablate/libtf_msda_<tag>.so: the library with ONE source (default msda_pquad.hip) compiled with the given macros
(e.g. the TF_PQUAD_ABLATE phase mask, TF_P2_STAGE_LDS, ... of msda_pquad.hip / msda_pquad2.h; TF_BWD_ABLATE / TF_BWD_PASSES of
msda_hip.hip).  Run the harness against it with
    LD_PRELOAD=tools/bin/ablate/libtf_msda_<tag>.so tools/bin/msda_bench --sets 4 --patterns pert pquad
or  TF_MSDA_LIB=tools/bin/ablate/libtf_msda_<tag>.so python tools/bench_msda.py ..."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from trackformer_amd import build as tfbuild  # noqa: E402


def main():
    tag, defines = sys.argv[1], sys.argv[2:]
    source = "msda_pquad.hip"
    if "--source" in defines:
        i = defines.index("--source")
        source = defines[i + 1]
        defines = defines[:i] + defines[i + 2:]
    stem = os.path.splitext(source)[0]
    tfbuild.build_all()
    out_dir = os.path.join(REPO, "tools", "bin", "ablate")
    os.makedirs(out_dir, exist_ok=True)
    obj_dir = os.path.join(tfbuild.LIB_DIR, "obj")
    others = [os.path.join(obj_dir, f) for f in sorted(os.listdir(obj_dir)) if f.endswith(".o") and f != stem + ".o"]
    hipcc = tfbuild._hipcc()
    flags = ["--offload-arch=" + tfbuild.GFX_ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + tfbuild.INCLUDE, "-Wno-pass-failed"]
    obj = os.path.join(out_dir, "%s_%s.o" % (stem, tag))
    so = os.path.join(out_dir, "libtf_msda_%s.so" % tag)
    subprocess.check_call([hipcc] + flags + defines + ["-c", os.path.join(tfbuild.CSRC, source), "-o", obj])
    subprocess.check_call([hipcc, "--offload-arch=" + tfbuild.GFX_ARCH, "-shared", "-fPIC", obj] + others + ["-o", so])
    os.remove(obj)
    print("built", os.path.relpath(so, REPO))


if __name__ == "__main__":
    main()
