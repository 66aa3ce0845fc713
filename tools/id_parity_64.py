"""the 64-frame 800x1333 reference-Tracker fixture (tests/golden/full_tracker_cfg2_64.npz, produced by the
reference's own Tracker on CPU: tracker.py:266-550, the two NMS passes at :399 and :493-495) run TO THE END under each
arithmetic set-up of the dense layers, recording the first frame whose set of live track ids differs from the reference's
and the NMS margin the fixture recorded for that frame.

    python tools/id_parity_64.py [--frames 64] [--setups fp32_library,split3,split3_heads_fp32,split6] > profiles/r04_id_parity_64.txt

Set-ups:
  fp32_library       fused.set_split_linear(False): hipBLASLt / MIOpen fp32 for every dense layer
  split3             the three-term bf16 split product everywhere (round-3 default)
  split16            the fp16 split product (three terms; include/tf_fused.h)
  split3_heads_fp32  three-term products, but class_embed / bbox_embed (the decision-critical tail) through the fp32 library
  split6             the six-term (hi / mid / lo) split product everywhere: fp32-accurate products (dropped terms < 2^-24)
Once two runs differ in one id, the track queries fed back differ and everything after is a different sequence: only the
FIRST differing frame means anything.  Pairwise first-difference between set-ups is printed as well (is fp32 on the GPU
any closer to fp32 on the CPU than the split product is?).
"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from tests import util_models as um   # noqa: E402


def run(setup, n_frames, dev):
    from trackformer_amd import config, factory, fused, runtime
    from trackformer_amd.graphed import GraphedDetector
    from trackformer_amd.tracker import Tracker
    model, post, args = um.build("cfg2_full", factory.build_model, config.make_args, device=dev)
    model.to(dev).tracking()
    runtime.configure_inference(verbose=False)
    prev = {}
    if setup == "fp32_library":
        prev["split"] = fused.set_split_linear(False)
    elif setup in ("split3", "split3_heads_fp32"):
        raise ValueError("the three-term bf16 product was removed in round 5 (its results: profiles/r04_id_parity_64.txt)")
    elif setup in ("split6", "split16"):
        if not hasattr(fused, "set_split_terms"):
            return None
        prev["split"] = fused.set_split_linear(True)
        prev["terms"] = fused.set_split_terms(6 if setup == "split6" else 16)
    else:
        raise ValueError(setup)
    try:
        tracker = Tracker(GraphedDetector(model), post, config.tracker_cfg(), False)
        tracker.reset()
        ids = []
        with torch.no_grad():
            for blob in um.full_tracker_sequence(n_frames=n_frames):
                tracker.step(dict(blob, img=blob['img'].to(dev)))
                ids.append(sorted(t.id for t in tracker.tracks))
    finally:
        if "split" in prev:
            fused.set_split_linear(prev["split"])
        if "heads" in prev:
            fused.set_heads_split(prev["heads"])
        if "terms" in prev:
            fused.set_split_terms(prev["terms"])
    return ids


def first_diff(a, b):
    for f, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return f
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--setups", default="fp32_library,split16,split6")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(REPO, "tests", "golden", "full_tracker_cfg2_64.npz"))
    margins = z["nms_iou_margin_per_frame"]
    rows = z["rows"]
    gold = [sorted(int(r[0]) for r in rows[rows[:, 1] == f]) for f in range(a.frames)]
    print("# 64-frame reference-Tracker fixture, %d frames run per set-up; min score margin of the fixture %.3f"
          % (a.frames, float(z["min_score_margin"])))
    print("# NMS IoU margin per frame (reference run): " + " ".join("%.1e" % m for m in margins[:a.frames]))
    must = int(np.argmax(margins < 1e-3)) if (margins < 1e-3).any() else len(margins)
    print("# frames whose NMS margin is >= 1e-3 (must agree under a 1e-3 box tolerance): %d" % must)
    res = {}
    for s in a.setups.split(","):
        ids = run(s, a.frames, dev)
        if ids is None:
            print("%-18s not available in this build" % s)
            continue
        res[s] = ids
        fd = first_diff(ids, gold)
        if fd is None:
            print("%-18s agrees with the reference on all %d frames (%d live tracks at the end)" % (s, a.frames, len(ids[-1])))
        else:
            sym = set(ids[fd]) ^ set(gold[fd])
            print("%-18s first differing frame %d (NMS margin of that frame %.1e, %d live tracks before it, %d ids differ)"
                  % (s, fd, margins[fd], len(gold[fd - 1]) if fd else 0, len(sym)))
    names = list(res)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            fd = first_diff(res[names[i]], res[names[j]])
            print("pair %-18s vs %-18s: %s" % (names[i], names[j], "identical ids on all frames" if fd is None else
                                               "first differing frame %d" % fd))


if __name__ == "__main__":
    main()
