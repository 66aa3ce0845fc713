#!/usr/bin/env python
"""One table from a directory of bench.py JSON lines (gpurun_out/r03a, r03b, ...): frames/s, ms per step and the change
against the run named *default* of the same configuration.

    python tools/summarize_bench.py gpurun_out/r03a [gpurun_out/r03b ...]
"""
import glob
import json
import os
import sys


def load(path):
    try:
        for line in open(path):
            line = line.strip()
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    except (OSError, ValueError):
        pass
    return None


def main():
    rows = []
    for d in sys.argv[1:] or ["gpurun_out/r03a"]:
        for f in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
            j = load(f)
            name = os.path.basename(f)[len("bench_"):-len(".json")]
            cfg = name.split("_")[0]
            if j is None:
                err = f[:-5] + ".err"
                tail = open(err).read().strip().splitlines()[-1][:100] if os.path.exists(err) and os.path.getsize(err) else "no JSON line"
                rows.append((cfg, name, None, None, tail))
            else:
                rows.append((cfg, name, j["value"], j["ms_per_step"], j.get("single_sequence_fps")))
    base = {cfg: v for cfg, name, v, _, _ in rows if v is not None and name.endswith("default")}
    print("%-34s %10s %9s %8s  %s" % ("run", "value", "ms/step", "vs dflt", "single-sequence fps / error"))
    for cfg, name, v, ms, extra in rows:
        if v is None:
            print("%-34s %10s %9s %8s  %s" % (name, "-", "-", "-", extra))
        else:
            rel = "%+.1f%%" % (100.0 * (v / base[cfg] - 1.0)) if cfg in base else ""
            print("%-34s %10.2f %9.3f %8s  %s" % (name, v, ms, rel, extra if extra is not None else ""))


if __name__ == "__main__":
    main()
