#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel family.

    python tools/pmc_summary.py <counter_collection.csv> [out.json] [--match substr ...]

Sums every counter over the dispatches of each kernel family (name prefix up to the first '<' / '(' and
at most 60 characters), reports dispatch counts and, where the counters are present, derived ratios:
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES      (matrix-core utilisation)
  FETCH_SIZE / WRITE_SIZE are reported in bytes with the gfx950 x2 correction of FETCH_SIZE
  (MI355X_MICROARCH.md, "HBM") applied as `fetch_bytes_corrected`.
"""
import collections
import csv
import json
import sys


def family(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    for sep in ("<", "("):
        if sep in name:
            name = name.split(sep)[0]
    return name.strip()[:60]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = []
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1:]
        args = [a for a in args if a not in match]
    rows = list(csv.DictReader(open(args[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in rows:
        fam = family(r["Kernel_Name"])
        if match and not any(m in r["Kernel_Name"] for m in match):
            continue
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[fam].add(r["Dispatch_Id"])
    out = {}
    for fam, c in agg.items():
        d = dict(c)
        d["dispatches"] = len(disp[fam])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CU_CYCLES"):
            d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CU_CYCLES"]
        if "FETCH_SIZE" in d:   # counter unit: KiB (rocprofv3 derived metric)
            d["fetch_bytes_per_dispatch_corrected"] = 2 * d["FETCH_SIZE"] * 1024 / d["dispatches"]
        if "WRITE_SIZE" in d:
            d["write_bytes_per_dispatch"] = d["WRITE_SIZE"] * 1024 / d["dispatches"]
        out[fam] = d
    text = json.dumps(out, indent=1, sort_keys=True)
    if len(args) > 1:
        open(args[1], "w").write(text + "\n")
    print(text[:6000])


if __name__ == "__main__":
    main()
