#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel family.

    python tools/pmc_summary.py <counter_collection.csv> [out.json] [--match substr ...]

Sums every counter over the dispatches of each kernel family (name prefix up to the first '<' / '(' and
at most 60 characters), reports dispatch counts and, where the counters are present, derived ratios:
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration x 2.4 GHz x 1024 SIMDs)   (matrix-core utilisation)
      SQ_VALU_MFMA_BUSY_CYCLES is summed over every SIMD of the device (checked against first principles on the split GEMM:
      1.069 M v_mfma_f32_32x32x16_bf16 x 32 cycles = 34.2 M = the counter), so the denominator is the kernel's own
      duration (End - Start timestamp of the dispatch) in SIMD-cycles at the 2.4 GHz peak clock.  A SIMD cannot be busy
      longer than the kernel runs: the ratio cannot exceed 1, and a lower actual clock only makes it an under-estimate.
      (Round 2 divided by SQ_BUSY_CU_CYCLES, which is not a per-SIMD quantity: 1.18, meaningless.)
  FETCH_SIZE / WRITE_SIZE are reported in bytes with the gfx950 x2 correction of FETCH_SIZE
  (MI355X_MICROARCH.md, "HBM") applied as `fetch_bytes_corrected`.
"""
import collections
import csv
import json
import sys


PEAK_CLOCK_GHZ = 2.4     # MI355X peak engine clock
N_SIMD = 256 * 4         # 256 CUs x 4 SIMDs


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
    dur_ns = collections.defaultdict(dict)   # family -> dispatch id -> duration
    for r in rows:
        fam = family(r["Kernel_Name"])
        if match and not any(m in r["Kernel_Name"] for m in match):
            continue
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[fam].add(r["Dispatch_Id"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur_ns[fam][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for fam, c in agg.items():
        d = dict(c)
        d["dispatches"] = len(disp[fam])
        total_ns = sum(dur_ns[fam].values())
        if total_ns:
            d["dispatch_ns_total"] = total_ns
            d["avg_dispatch_us"] = total_ns / len(dur_ns[fam]) / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and total_ns:
            simd_cycles = total_ns * PEAK_CLOCK_GHZ * N_SIMD
            d["mfma_util"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles
            assert d["mfma_util"] <= 1.0 + 1e-6, (fam, d["mfma_util"])
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
