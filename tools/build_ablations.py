#!/usr/bin/env python
"""Timing ablations of msda_fwd_f32_pquad (tools only -- libtf_msda.so never contains them).

    python tools/build_ablations.py [mask ...]          # default: 1 2 3 4 8 16 32 7 47

Builds tools/bin/ablate/libtf_msda_abl<mask>.so: the library with msda_pquad.hip compiled with
-DTF_PQUAD_ABLATE=<mask> (see the macro's comment in trackformer_amd/csrc/msda_pquad.hip: 1 no LDS gathers, 2 no LDS-DMA
staging, 4 no bounding boxes, 8 no stores, 16 no fused prologue arithmetic, 32 no buffer-load fallback).  Run the
harness against one with
    LD_PRELOAD=tools/bin/ablate/libtf_msda_abl1.so tools/bin/msda_bench --sets 4 --patterns pert pquad
(the preloaded library's symbols win over the rpath'd libtf_msda.so).  The harness then reports wrong results -- by
design -- and the launch time without that phase: T(full) - T(ablated) is the phase's marginal cost under the real
overlap of the resident workgroups."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from trackformer_amd import build as tfbuild  # noqa: E402


def main():
    masks = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 8, 16, 32, 7, 47]
    tfbuild.build_all()                      # the other translation units' objects
    out_dir = os.path.join(REPO, "tools", "bin", "ablate")
    os.makedirs(out_dir, exist_ok=True)
    obj_dir = os.path.join(tfbuild.LIB_DIR, "obj")
    others = [os.path.join(obj_dir, f) for f in sorted(os.listdir(obj_dir)) if f.endswith(".o") and f != "msda_pquad.o"]
    hipcc = tfbuild._hipcc()
    flags = ["--offload-arch=" + tfbuild.GFX_ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + tfbuild.INCLUDE, "-Wno-pass-failed"]

    def one(mask):
        obj = os.path.join(out_dir, "msda_pquad_abl%d.o" % mask)
        so = os.path.join(out_dir, "libtf_msda_abl%d.so" % mask)
        subprocess.check_call([hipcc] + flags + ["-DTF_PQUAD_ABLATE=%d" % mask, "-c", os.path.join(tfbuild.CSRC, "msda_pquad.hip"), "-o", obj])
        subprocess.check_call([hipcc, "--offload-arch=" + tfbuild.GFX_ARCH, "-shared", "-fPIC", obj] + others + ["-o", so])
        os.remove(obj)
        return so

    with ThreadPoolExecutor(max_workers=min(len(masks), os.cpu_count() or 1)) as ex:
        for so in ex.map(one, masks):
            print("built", os.path.relpath(so, REPO))


if __name__ == "__main__":
    main()
