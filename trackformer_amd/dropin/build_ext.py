"""Builds the COMPILED form of the drop-in -- trackformer_amd/dropin/compiled/MultiScaleDeformableAttention.<abi>.so from
csrc/msda_ext.cpp (a pybind11 torch extension over the C ABI of libtf_msda.so, the form of the reference's plugin:
models/ops/setup.py:30-66, models/ops/src/vision.cpp:4-7).  Host C++ only (no kernels): compiled with g++ against the torch headers
of the running interpreter, linked to libtf_msda.so by a relative rpath.

    python -m trackformer_amd.dropin.build_ext          # or trackformer_amd.dropin.build_ext.build()
"""
import hashlib
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
REPO = os.path.dirname(PKG)
SRC = os.path.join(HERE, "csrc", "msda_ext.cpp")
OUT_DIR = os.path.join(HERE, "compiled")
NAME = "MultiScaleDeformableAttention"


def target():
    return os.path.join(OUT_DIR, NAME + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build(force=False, verbose=False):
    """-> path of the extension module.  Rebuilt when the content hash of its source, the C-ABI header, this recipe and the
    torch version differs from the stamp beside it (not by mtime: the gpurun snapshot does not keep mtime order).  The library
    itself is linked dynamically: a rebuilt libtf_msda.so needs no rebuild here, the module checks tf_msda_abi_version() when
    it is imported."""
    import torch
    from torch.utils import cpp_extension as ce
    out = target()
    lib = os.path.join(PKG, "lib", "libtf_msda.so")
    if not os.path.exists(lib):
        raise RuntimeError("build libtf_msda.so first (python -m trackformer_amd.build)")
    h = hashlib.sha256(torch.__version__.encode())
    for d in (SRC, os.path.join(REPO, "include", "tf_msda.h"), os.path.abspath(__file__)):
        with open(d, "rb") as f:
            h.update(f.read())
    key = h.hexdigest()
    stamp = out + ".stamp"
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == key:
        return out
    os.makedirs(OUT_DIR, exist_ok=True)
    rocm = ce.ROCM_HOME or "/opt/rocm"
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", out,
           "-I" + os.path.join(REPO, "include"), "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(rocm, "include"),
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=" + NAME, "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-deprecated-declarations",
           "-L" + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-ltorch_python",
           "-L" + os.path.join(PKG, "lib"), "-ltf_msda",
           "-Wl,-rpath,$ORIGIN/../../lib", "-Wl,-rpath," + tlib]
    cmd[5:5] = ["-I" + p for p in ce.include_paths()]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(key + "\n")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
