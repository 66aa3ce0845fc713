"""Builds the native pieces in-tree (no JIT cache, so the .so files travel with the repo snapshot).

    python -m trackformer_amd.build          # compile everything for gfx950
    python -m trackformer_amd.build --force

hipcc cross-compiles gfx950 code objects without a GPU present.
"""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
INCLUDE = os.path.join(REPO_DIR, "include")
GFX_ARCH = "gfx950"

# (output, [sources], extra flags)
_TARGETS = [
    ("libtf_msda.so", ["msda_hip.hip", "fused_ops.hip", "linear_split.hip"], []),
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (expected on PATH or at /opt/rocm/bin/hipcc)")
    return exe


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_all(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    built = []
    headers = [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    headers += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]   # kernel includes
    for out_name, srcs, extra in _TARGETS:
        out = os.path.join(LIB_DIR, out_name)
        src_paths = [os.path.join(CSRC, s) for s in srcs]
        if not force and not _stale(out, src_paths + headers):
            continue
        cmd = [_hipcc(), "--offload-arch=" + GFX_ARCH, "-O3", "-std=c++17", "-shared", "-fPIC",
               "-I" + INCLUDE, "-Wno-pass-failed"] + extra + src_paths + ["-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        built.append(out)
    built += _build_tools(force, verbose)
    return built


def _build_tools(force, verbose):
    """tools/bin/msda_bench, tools/bin/linear_bench: standalone parity + timing harnesses of the forward kernels and of
    the split-product linear (no Python on the GPU box).  They link against the in-tree libtf_msda.so through a
    relative rpath."""
    out_dir = os.path.join(REPO_DIR, "tools", "bin")
    lib = os.path.join(LIB_DIR, "libtf_msda.so")
    built = []
    for name in ("msda_bench", "linear_bench"):
        src = os.path.join(REPO_DIR, "tools", name + ".cpp")
        if not os.path.exists(src):
            continue
        os.makedirs(out_dir, exist_ok=True)
        out = os.path.join(out_dir, name)
        if not force and not _stale(out, [src, lib]):
            continue
        cmd = [_hipcc(), "--offload-arch=" + GFX_ARCH, "-O2", "-std=c++17", "-I" + INCLUDE, src,
               "-L" + LIB_DIR, "-ltf_msda", "-Wl,-rpath,$ORIGIN/../../trackformer_amd/lib", "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        built.append(out)
    return built


if __name__ == "__main__":
    res = build_all(force="--force" in sys.argv, verbose=True)
    print("built:" if res else "up to date", *res)
