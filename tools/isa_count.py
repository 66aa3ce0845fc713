#!/usr/bin/env python
"""Static instruction mix of the gfx950 kernels in a hipcc -S listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude --cuda-device-only -S \
        trackformer_amd/csrc/msda_hip.hip -o /tmp/msda.s
    python tools/isa_count.py /tmp/msda.s [substring-of-kernel-name ...]

Counts are per basic-block label (static: a loop body counts once), so that the cost of each phase of
a kernel can be read off without a GPU: VALU (v_*), packed VALU, SALU, LDS (ds_*), vector memory
(buffer_/global_), scalar memory, waits/barriers."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "vpk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    kern, label = None, None
    data = collections.OrderedDict()
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, label = m.group(1), "entry"
            data[kern] = collections.OrderedDict()
            continue
        if kern is None:
            continue
        if line.startswith(".Lfunc_end"):
            kern = None
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            label = m.group(1)
            continue
        m = re.match(r"^\s+([a-z]\w+)", line)
        if m and not line.lstrip().startswith("."):
            data[kern].setdefault(label, collections.Counter())[classify(m.group(1))] += 1
    for kern, blocks in data.items():
        if pats and not any(p in kern for p in pats):
            continue
        tot = collections.Counter()
        print(kern)
        for label, c in blocks.items():
            tot.update(c)
            if "-v" in pats or len(pats) > 0:
                print("  %-14s %s" % (label, dict(c)))
        print("  TOTAL          %s" % dict(tot))


if __name__ == "__main__":
    main()
