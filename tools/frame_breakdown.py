#!/usr/bin/env python
"""Per-frame steady-state kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py --no-graph."""
import collections
import csv
import sys


def main(path, frames=5, out=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ms = [i for i, r in enumerate(rows) if 'msda_fwd' in r['Kernel_Name']]
    starts = ms[::12]                      # 12 MSDeformAttn calls per frame (6 enc + 6 dec)
    seg = rows[starts[-frames - 1]:starts[-1]]
    wall = (int(rows[starts[-1]]['Start_Timestamp']) - int(seg[0]['Start_Timestamp'])) / frames / 1e6
    agg = collections.defaultdict(lambda: [0, 0])
    busy = 0
    for r in seg:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        agg[r['Kernel_Name'][:120]][0] += d
        agg[r['Kernel_Name'][:120]][1] += 1
        busy += d
    lines = ["# GPU busy %.3f ms/frame, %.0f kernels/frame, profiled wall %.2f ms/frame (last %d frames)"
             % (busy / frames / 1e6, len(seg) / frames, wall, frames)]
    for k, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
        lines.append('%8.3f ms/frame %6.1f calls/frame  %s' % (d / frames / 1e6, c / frames, k))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print("\n".join(lines[:45]))


if __name__ == "__main__":
    main(sys.argv[1], out=sys.argv[2] if len(sys.argv) > 2 else None)
