#!/usr/bin/env python
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fills = [i for i, r in enumerate(rows) if 'erfinv' in r['Kernel_Name']]   # the marker launches between the calls
seg = rows[fills[-2] + 1:fills[-1]]
t0 = int(seg[0]['Start_Timestamp'])
tot = 0
print("# %d launches, %.1f us from first start to last end" % (len(seg), (int(seg[-1]['End_Timestamp']) - t0) / 1e3))
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    name = re.sub(r'void |\(anonymous namespace\)::|at::native::', '', r['Kernel_Name'])[:90]
    print("%8.1f  +%6.1f us  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, d, name))
print("# kernel sum %.1f us" % tot)
