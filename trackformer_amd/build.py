"""Builds the native pieces in-tree (no JIT cache).

    python -m trackformer_amd.build          # compile everything for gfx950
    python -m trackformer_amd.build --force

hipcc cross-compiles gfx950 code objects without a GPU present.  The built .so / tools are git-ignored
(the history stays source-only) but travel to the GPU box with the `gpurun` snapshot; a fresh clone
has to run this once (`_cabi.lib()` has no fallback and does not build behind the caller's back).
A target is rebuilt when the hash of its sources, headers, compiler flags and architecture differs
from the stamp written next to it -- not by mtime, so a flag or arch change rebuilds too.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
INCLUDE = os.path.join(REPO_DIR, "include")
GFX_ARCH = "gfx950"

# (output, [sources], extra flags)
_TARGETS = [
    ("libtf_msda.so", ["msda_hip.hip", "msda_pquad.hip", "fused_ops.hip", "linear_split.hip", "linear_stream.hip", "mha_core.hip", "ffn_fused.hip", "stem_conv.hip", "msda_host.cpp"], []),
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (expected on PATH or at /opt/rocm/bin/hipcc)")
    return exe


def _key(deps, flags):
    h = hashlib.sha256()
    h.update(("\0".join(flags) + "\0" + GFX_ARCH).encode())
    for d in sorted(deps):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(out, key):
    stamp = out + ".stamp"
    if not os.path.exists(out) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != key


def _stamp(out, key):
    with open(out + ".stamp", "w") as f:
        f.write(key + "\n")


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_all(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    built = []
    headers = [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    headers += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]   # kernel includes
    for out_name, srcs, extra in _TARGETS:
        out = os.path.join(LIB_DIR, out_name)
        src_paths = [os.path.join(CSRC, s) for s in srcs]
        flags = ["--offload-arch=" + GFX_ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE,
                 "-Wno-pass-failed"] + extra
        key = _key(src_paths + headers, flags)
        if not force and not _stale(out, key):
            continue
        # one hipcc per translation unit, in parallel, then one link
        obj_dir = os.path.join(LIB_DIR, "obj")
        os.makedirs(obj_dir, exist_ok=True)
        objs = [os.path.join(obj_dir, os.path.splitext(s)[0] + ".o") for s in srcs]
        with ThreadPoolExecutor(max_workers=len(srcs)) as pool:
            list(pool.map(lambda so: _run([_hipcc()] + flags + ["-c", so[0], "-o", so[1]], verbose),
                          zip(src_paths, objs)))
        _run([_hipcc(), "--offload-arch=" + GFX_ARCH, "-shared", "-fPIC", "-pthread"] + objs + ["-o", out], verbose)
        _stamp(out, key)
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
    for name in ("msda_bench", "linear_bench", "ffn_bench"):
        src = os.path.join(REPO_DIR, "tools", name + ".cpp")
        if not os.path.exists(src):
            continue
        os.makedirs(out_dir, exist_ok=True)
        out = os.path.join(out_dir, name)
        flags = ["--offload-arch=" + GFX_ARCH, "-O2", "-std=c++17", "-I" + INCLUDE]
        headers = [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
        key = _key([src] + headers, flags)
        if not force and not _stale(out, key) and os.path.getmtime(out) >= os.path.getmtime(lib):
            continue
        _run([_hipcc()] + flags + [src, "-L" + LIB_DIR, "-ltf_msda",
                                   "-Wl,-rpath,$ORIGIN/../../trackformer_amd/lib", "-o", out], verbose)
        _stamp(out, key)
        built.append(out)
    return built


if __name__ == "__main__":
    res = build_all(force="--force" in sys.argv, verbose=True)
    print("built:" if res else "up to date", *res)
