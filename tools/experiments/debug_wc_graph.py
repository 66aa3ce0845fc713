"""Debug aid (round 5): run some GPU model tests first (allocator history), then the 64-frame well-conditioned sequence under
GraphedDetector with per-frame checks; post-mortem of the static graph buffers after a wrong frame.  This is how the
hipMemsetAsync-inside-a-graph ordering bug behind tf_groupnorm_nhwc_f32 was found (profiles/r05_graph_memset_groupnorm.txt:
the finer per-stage hooks that localised it lived in the package for the duration of the hunt and are gone again)."""
import sys
import numpy as np
import pytest
import torch

sys.path.insert(0, ".")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which != "none":
    sel = {"all": "model_forward or tracker_track_ids or variants or (well_conditioned and cfg2_wc and eager)",
           "forward": "model_forward", "tracker": "tracker_track_ids or variants", "wceager": "well_conditioned and cfg2_wc and eager"}[which]
    pytest.main(["tests/test_models_gpu.py", "-m", "gpu", "-q", "-x", "-k", sel])
from tests import test_models_cpu as shared, util_models as um
from trackformer_amd import config, factory, fused, runtime
from trackformer_amd.graphed import GraphedDetector
from trackformer_amd.tracker import Tracker
dev = torch.device("cuda:0")
SYNC = "sync" in sys.argv
if "nogc" in sys.argv:
    import gc
    gc.disable()
NF = 5 if "f5" in sys.argv else 66 if "f66" in sys.argv else 80 if "f80" in sys.argv else 300 if "f300" in sys.argv else 64
if "nosplit" in sys.argv:
    GraphedDetector._splittable = lambda self: False
print("sync", SYNC, "split", "nosplit" not in sys.argv)
print("split terms", fused.split_terms(), "split linear", fused.split_linear_enabled(), "cudnn.benchmark", torch.backends.cudnn.benchmark)
z = np.load("tests/golden/tracker_cfg2_wc.npz")
for rep in range(1):
    model, post, args = um.build("cfg2_deformable_tracking", factory.build_model, config.make_args, device=dev)
    um.shape_well_conditioned(model)
    model.to(dev).tracking()
    det = GraphedDetector(model)
    tracker = Tracker(det, post, config.tracker_cfg(), False)
    tracker.reset()
    wrong = []
    with torch.no_grad():
        for f, blob in enumerate(um.tracker_sequence(n_frames=NF)):
            nq = len(tracker.tracks) + len(tracker.inactive_tracks)
            h = tracker.step_async(blob)
            if SYNC:
                torch.cuda.synchronize()
            tracker.step_finish(h)
            if "trim" in sys.argv:
                for t in tracker.tracks:
                    del t.hs_embed._items[:-1]
            if "extra" in sys.argv and f == 10 and det._enc:
                a_ = list(det._enc.values())[0]
                for _ in range(20):
                    a_["graph"].replay()
            if len(tracker.tracks) != int(z["active_per_frame"][min(f, 63)]):
                host = h["host"][:h["packed_dev"].shape[0]]
                dev_now = h["packed_dev"].cpu()
                a = list(det._enc.values())[0]
                b = list(det._graphs.values())[0]
                nan = lambda t: int(torch.isnan(t).sum())
                print("   static buffers after the wrong frame: memory", nan(a["state"]["enc"]["memory"]), "feats", [nan(x.tensors) for x in a["state"]["features_all"]],
                      "img", nan(a["img"]), "entry hs", nan(b["hs"]), "entry boxes", nan(b["boxes"]), "logits", nan(b["out"][0]["pred_logits"]),
                      "boxes", nan(b["out"][0]["pred_boxes"]), "hs per layer", [nan(x) for x in b["out"][4]], "outputs clone", nan(h["outputs"]["pred_logits"]))
                print("   entry hs absmax %.3g  memory absmax %.3g  tracks' query rows:" % (float(b["hs"].abs().max()), float(a["state"]["enc"]["memory"].abs().max())),
                      b["hs"].abs().amax(1).topk(3).values.tolist())
                wrong.append((f, len(tracker.tracks), float((host - dev_now).abs().max()), float(dev_now[:, 4].max()), float(host[:, 4].max())))
    wrong = [w for w in wrong if w[1] == 0]
    print("WRONG:", [(w[0], w[1]) for w in wrong])
    print("rep", rep, "wrong frames (frame, active, |host - device| of the packed rows, max score device, max score host):", wrong[:6])
