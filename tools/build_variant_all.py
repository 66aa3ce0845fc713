#!/usr/bin/env python
"""Builds a VARIANT of the whole library -- every translation unit compiled with the given macros -- for A/B timing:

    python tools/build_variant_all.py <tag> -DNAME=VALUE [...]      ->  tools/bin/ablate/libtf_msda_<tag>.so

(tools/build_variant.py rebuilds ONE source; a macro of msda_common.h such as TF_STORE_AUX touches all of them.)  Use it with
TF_MSDA_LIB=<path> python bench.py ... or LD_PRELOAD=<path> tools/bin/msda_bench ..."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from trackformer_amd import build as tfbuild  # noqa: E402


def main():
    tag, defines = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(REPO, "tools", "bin", "ablate")
    obj_dir = os.path.join(out_dir, "obj_" + tag)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = tfbuild._hipcc()
    flags = ["--offload-arch=" + tfbuild.GFX_ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + tfbuild.INCLUDE, "-Wno-pass-failed"] + defines
    srcs = tfbuild._TARGETS[0][1]
    objs = [os.path.join(obj_dir, os.path.splitext(s)[0] + ".o") for s in srcs]
    with ThreadPoolExecutor(max_workers=len(srcs)) as pool:
        list(pool.map(lambda so: subprocess.check_call([hipcc] + flags + ["-c", os.path.join(tfbuild.CSRC, so[0]), "-o", so[1]]), zip(srcs, objs)))
    so = os.path.join(out_dir, "libtf_msda_%s.so" % tag)
    subprocess.check_call([hipcc, "--offload-arch=" + tfbuild.GFX_ARCH, "-shared", "-fPIC", "-pthread"] + objs + ["-o", so])
    for o in objs:
        os.remove(o)
    os.rmdir(obj_dir)
    print("built", os.path.relpath(so, REPO))


if __name__ == "__main__":
    main()
