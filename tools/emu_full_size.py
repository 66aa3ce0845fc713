#!/usr/bin/env python
"""One-off check (minutes on 8 cores; not part of the test suite): the BASELINE-size model (800 x 1333; cfg 2: 300 + 100
queries, cfg 4: hidden 288, 500 + 300) through the package's GPU inference path ON THE SIMT EMULATOR
(tests/util_emu_gpu_path.py), with every opt-in route on, against the full-size goldens of the reference's own classes.

    python tools/emu_full_size.py [cfg2_full|cfg4_full] [--defaults]
"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests import test_full_size_gpu as full  # noqa: E402
from tests import util_models as um  # noqa: E402
from tests.util_emu_gpu_path import gpu_path_on_emulator  # noqa: E402
from trackformer_amd import backbone, config, factory, fused  # noqa: E402


def main():
    case = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg2_full"
    optin = "--defaults" not in sys.argv
    model, post, args = um.build(case, factory.build_model, config.make_args)
    model.tracking()
    img, prev, target = um.model_inputs(case, args.hidden_dim)
    t0 = time.time()
    with gpu_path_on_emulator() as lib:
        if optin:
            backbone.set_conv1x1_split(True)
            backbone.set_conv3x3_split(True)
            fused.set_input_proj_fused(True)
            fused.set_box_refine_fused(True)
            fused.set_ffn_fused(True)        # the row limits of fused.ffn / linear_residual_norm stay: what bench.py would run
            fused.set_linear_ln_fused(True)
            fused.set_stem_pool_fused(True)
            fused.set_pos_add_fused(True)
            fused.set_stem_conv_split(True)
            fused.set_heads_split(True)
            for k, v in ((b"direct9", 1),):
                lib.tf_msda_set_option(k, v)
        with torch.no_grad():
            prev_features = None
            if args.multi_frame_attention:
                _, _, prev_features, _, _ = model(prev, None, None)
            out, _, feats, memory, hs = model(img, target, prev_features)
            res = post['bbox'](out, torch.tensor([list(um.FULL_ORIG)]))[0]
        calls = dict(lib.calls)
    dbox, dlogit = full._compare(case, model, out, res, feats, memory)
    print("%s on the emulator (%s): max |d boxes| %.2e, max |d logits| %.2e, %.0f s" % (
        case, "every opt-in route" if optin else "defaults", dbox, dlogit, time.time() - t0))
    print("calls:", {k: v for k, v in sorted(calls.items())})


if __name__ == "__main__":
    main()
