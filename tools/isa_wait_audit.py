#!/usr/bin/env python
"""Static audit of the compiled gfx950 kernels for ONE defect class (found by hand in split_conv3_kernel, DESIGN.md section
4.4): inside a loop, vector-memory loads that are waited for right where they were issued -- an `s_waitcnt vmcnt(k)` that
covers loads issued since the last matrix instruction, with no v_mfma in between -- so that a prefetch meant to be covered
by the MFMAs of the current K-slice exposes its full memory latency instead.  The usual cause: arithmetic on the loaded value
written next to the load (`x = ok ? v : 0`, `x = a + b`), which hipcc schedules right behind it.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itrackformer_amd/csrc --cuda-device-only -S <file.hip> -o /tmp/k.s
    python tools/isa_wait_audit.py /tmp/k.s [substring-of-kernel-name ...]

Per kernel and loop: the loads issued, the waits that cover fresh loads before any MFMA, and how many MFMAs the loop has.
Only loops that contain MFMAs are reported (the GEMM family); `fresh` > 0 is a finding."""
import re
import sys


def audit(path, pats):
    kern, lines = None, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1)
            lines[kern] = []
            continue
        if kern is None:
            continue
        if line.startswith(".Lfunc_end"):
            kern = None
            continue
        lines[kern].append(line.rstrip("\n"))
    for kern, body in lines.items():
        if pats and not any(p in kern for p in pats):
            continue
        # blocks: label -> its lines (up to the next label); a loop = its header block + every block the compiler marked
        # "in Loop: Header=<it>", walked in layout order starting at the header (hipcc rotates loops: the MFMA block often
        # precedes the header in the listing)
        reported = False
        blocks, order, cur = {}, [], None
        for l in body:
            m = re.match(r"^(\.LBB\w+):(.*)", l)
            if m:
                cur = m.group(1)
                blocks[cur] = [l]
                order.append(cur)
            elif cur is not None:
                blocks[cur].append(l)
        for name in order:
            if "Loop Header" not in blocks[name][0]:
                continue
            tag = "Header=" + name[2:]          # .LBB56_6 -> Header=BB56_6
            members = [b for b in order if b == name or tag in blocks[b][0]]
            k = members.index(name)
            loop = [l for b in members[k:] + members[:k] for l in blocks[b]]
            n_mfma = sum("v_mfma" in l for l in loop)
            if not n_mfma:
                continue
            # walk the loop twice (the second pass sees the loads the first one left in flight)
            inflight_fresh, findings, loads = 0, [], 0
            for rnd in range(2):
                for l in loop:
                    op = l.strip().split()[0] if l.strip() else ""
                    if op.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in l:
                        inflight_fresh += 1
                        loads += rnd
                    elif op.startswith("v_mfma"):
                        inflight_fresh = 0          # whatever is in flight now has matrix work to hide under
                    elif op == "s_waitcnt" and "vmcnt" in l:
                        k = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
                        if inflight_fresh > k and rnd:
                            findings.append((inflight_fresh - k, l.strip()))
                            inflight_fresh = k
            short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", kern)[:90]
            reported = True
            print("%-92s loop %-10s loads %2d  mfma %3d  fresh-waits %s" % (short, name, loads, n_mfma,
                  "none" if not findings else "; ".join("%d load(s) at `%s`" % f for f in findings[:4])))
        if not reported and any("v_mfma" in l for l in body):
            # fully unrolled kernels (split_gemm_deep_kernel): one pass over the body, waits after the first MFMA only (the
            # prologue legitimately waits for its first slice)
            fresh, findings, n_mfma, seen = 0, [], 0, False
            for l in body:
                op = l.strip().split()[0] if l.strip() else ""
                if op.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in l:
                    fresh += 1
                elif op.startswith("v_mfma"):
                    fresh, n_mfma, seen = 0, n_mfma + 1, True
                elif op == "s_waitcnt" and "vmcnt" in l and seen:
                    k = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
                    if fresh > k:
                        findings.append((fresh - k, l.strip()))
                        fresh = k
            short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", kern)[:90]
            print("%-92s straight-line      mfma %3d  fresh-waits %s" % (short, n_mfma,
                  "none" if not findings else "; ".join("%d load(s) at `%s`" % f for f in findings[:4])))


if __name__ == "__main__":
    audit(sys.argv[1], sys.argv[2:])
