#!/usr/bin/env python
"""profiles/rNN_mfma_utilisation.json (what bench.py's `mfma_utilisation.pmc` reads) from the per-run summaries tools/pmc_summary.py
wrote under a GPU call's output directory (mfma_lin1.json, mfma_lin2.json, mfma_ffn.json, mfma_frame.json):

    python tools/assemble_mfma_json.py gpurun_out/r04_09 profiles/r04_mfma_utilisation.json --commit "<what tree>" --terms 16 --how "<command>"
"""
import argparse
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--commit", required=True)
    ap.add_argument("--terms", type=int, required=True)
    ap.add_argument("--how", default="")
    a = ap.parse_args()
    out = {"_how": a.how, "_commit": a.commit, "_terms": a.terms}
    for key in ("mfma_lin1", "mfma_lin2", "mfma_ffn", "mfma_frame"):
        f = os.path.join(a.src, key + ".json")
        if os.path.exists(f):
            with open(f) as fh:
                out[key] = json.load(fh)
    with open(a.dst, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", a.dst, sorted(k for k in out if not k.startswith("_")))


if __name__ == "__main__":
    main()
