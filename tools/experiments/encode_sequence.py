#!/usr/bin/env python
"""The ordered kernel sequence of the image-only half of a cfg-2 frame (backbone, input projections, encoder) -- run under
rocprofv3 --kernel-trace; tools/experiments/encode_sequence_report.py prints the last of the three calls."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from trackformer_amd import runtime  # noqa: E402

runtime.configure_inference(verbose=False)
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["cfg2"]
model, criterion, post, margs = bench.build_model(cfg, dev)
model.tracking()
img = bench.make_frames(dev, cfg["size"], n=1)[0]['img']
marker = torch.zeros(64, device=dev)
with torch.no_grad():
    for i in range(4):
        st = model.encode_frame(img, None)
        torch.cuda.synchronize()
        marker.erfinv_()   # marker launch between the calls
        torch.cuda.synchronize()
