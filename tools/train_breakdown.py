#!/usr/bin/env python
"""Steady-state per-step kernel breakdown of tools/bench_train.py from a rocprofv3 --kernel-trace CSV.

One step = the window between the first encoder-shaped MSDeformAttn backward launch of the last two
steps (second half of step k + first half of step k+1: the same kernels as one whole step, and free of
the MIOpen find-mode kernels of the warm-up iterations)."""
import collections
import csv
import sys


def main(path, out=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    big = [i for i, r in enumerate(rows) if 'msda_bwd' in r['Kernel_Name'] and dur(r) > 1_000_000]
    firsts = [i for k, i in enumerate(big)
              if k == 0 or int(rows[i]['Start_Timestamp']) - int(rows[big[k - 1]]['Start_Timestamp']) > 30_000_000]
    a, b = firsts[-2], firsts[-1]
    seg = rows[a:b]
    wall = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:110]
        agg[name][0] += dur(r)
        agg[name][1] += 1
    busy = sum(v[0] for v in agg.values()) / 1e6
    lines = ["# one training step (bs 2, 800x1333): wall %.1f ms, GPU busy %.1f ms, %d kernels" % (wall, busy, len(seg))]
    for k, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:40]:
        lines.append('%8.2f ms %5d calls  %s' % (d / 1e6, c, k))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
