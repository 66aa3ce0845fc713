#!/usr/bin/env python
"""Static audit of the gfx950 kernels (no GPU needed): compiles each translation unit of libtf_msda.so to assembly and
reports, per kernel, what a profile would otherwise have to reveal:

    python tools/isa_audit.py [--all] [--stream KERNEL_SUBSTRING]

  vgpr / scratch    allocated registers, scratch bytes (spills)
  stores / waited   global / buffer stores, and how many of them are directly preceded by `s_waitcnt vmcnt(0)` -- on
                    gfx9-family hardware vmcnt counts stores too, so such a store waits for every earlier store to reach
                    L2 (how the serialised GEMM epilogue of round 2 was found: profiles/r02_static_isa_gemm_epilogue_waits.txt)
  loops             backward `s_cbranch_execnz` (waterfall / divergent loops)

Without --all only kernels with >= 4 waited stores or scratch are listed.  --stream prints the compact instruction stream
(L load, G buffer load, DMA buffer_load..lds, r/w LDS, M MFMA, S store, B barrier, br branch, W[..] waits) of a kernel."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "trackformer_amd", "csrc")
UNITS = ["msda_hip.hip", "msda_pquad.hip", "fused_ops.hip", "linear_split.hip", "linear_stream.hip", "mha_core.hip", "ffn_fused.hip", "stem_conv.hip"]


def assembly(unit, cache):
    out = os.path.join(cache, unit + ".s")
    src = os.path.join(CSRC, unit)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"),
                               "-Wno-pass-failed", "--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
    return out


def kernels(path):
    name, buf, meta = None, [], {}
    for line in open(path):
        m = re.match(r"^(_Z\S+):\s", line)
        if m:
            name, buf, meta = m.group(1), [], {}
            continue
        if name is None:
            continue
        t = line.split(";")[0].strip()
        if t.startswith(".amdhsa_next_free_vgpr"):
            meta["vgpr"] = int(t.split()[-1])
        elif t.startswith(".amdhsa_private_segment_fixed_size"):
            meta["scratch"] = int(t.split()[-1])
        elif ".end_amdhsa_kernel" in line:
            yield name, [x for x in buf if x and not x.startswith(".")], meta
            name = None
        else:
            buf.append(t)


def demangle(n):
    try:
        out = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        return out or n
    except OSError:
        return n


def compact(ins):
    out = []
    for x in ins:
        op = x.split()[0]
        if op == "s_waitcnt":
            out.append("W[" + x.split(None, 1)[1].replace(" ", "") + "]")
        elif op.startswith("global_load"):
            out.append("L")
        elif op.startswith("buffer_load"):
            out.append("DMA" if x.rstrip().endswith("lds") else "G")
        elif "mfma" in op:
            out.append("M")
        elif op.startswith("ds_write"):
            out.append("w")
        elif op.startswith("ds_read"):
            out.append("r")
        elif re.match(r"(global|buffer|flat)_store", op):
            out.append("S")
        elif op == "s_barrier":
            out.append("B")
        elif op.startswith("s_cbranch"):
            out.append("br")
    return " ".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--stream", default=None)
    args = ap.parse_args()
    cache = os.path.join(tempfile.gettempdir(), "tf_isa_audit")
    os.makedirs(cache, exist_ok=True)
    print("%-16s %5s %7s %6s %6s %5s  %s" % ("unit", "vgpr", "scratch", "stores", "waited", "loops", "kernel"))
    for unit in UNITS:
        for name, ins, meta in kernels(assembly(unit, cache)):
            stores = [i for i, x in enumerate(ins) if re.match(r"(global|buffer|flat)_store", x)]
            waited = 0
            for i in stores:
                for j in range(i - 1, max(-1, i - 8), -1):
                    if ins[j].startswith("s_waitcnt") and "vmcnt(0)" in ins[j]:
                        waited += 1
                        break
                    if re.match(r"(global|buffer|flat)_store", ins[j]):
                        break
            loops = sum(1 for x in ins if x.startswith("s_cbranch_execnz"))
            pretty = demangle(name).replace("(anonymous namespace)::", "").replace("void ", "", 1).split("(")[0]
            if args.stream:
                if args.stream in pretty or args.stream in name:
                    print("== %s\n%s\n" % (pretty, compact(ins)))
                continue
            if args.all or waited >= 4 or meta.get("scratch"):
                print("%-16s %5s %7s %6d %6d %5d  %s" % (unit, meta.get("vgpr"), meta.get("scratch"), len(stores), waited, loops, pretty))


if __name__ == "__main__":
    main()
