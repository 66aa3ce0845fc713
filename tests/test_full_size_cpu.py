"""CPU (-m "not gpu"): the BASELINE-size model path (800x1333; cfg 2: 300 + 100 queries, cfg 4: hidden 288,
500 + 300 queries, 8 decoder levels) through the repo's nn.Modules on the host with the C oracle standing
in for the HIP operator, against goldens from the reference's own classes (tests/golden/make_golden_full.py).
Pins the host-side graph at the sizes bench.py runs; the GPU counterpart is tests/test_full_size_gpu.py."""
import pytest
import torch

from oracle import msda_oracle
from tests import test_full_size_gpu as full
from tests import util_models as um
from trackformer_amd import config, factory, msda


@pytest.mark.parametrize("case", um.FULL_DETECTOR_CASES)
def test_full_size_model_matches_reference_on_cpu(case, monkeypatch):
    monkeypatch.setattr(msda, "MSDeformAttnFunction", msda_oracle.make_torch_function())
    model, post, args = um.build(case, factory.build_model, config.make_args)
    model.tracking()
    img, prev, target = um.model_inputs(case, args.hidden_dim)
    with torch.no_grad():
        prev_features = None
        if args.multi_frame_attention:
            _, _, prev_features, _, _ = model(prev, None, None)
        out, _, feats, memory, hs = model(img, target, prev_features)
        res = post['bbox'](out, torch.tensor([list(um.FULL_ORIG)]))[0]
    full._compare(case, model, out, res, feats, memory, box_tol=2e-5, logit_tol=1e-4)


def test_cfg1_plain_detr_480x640_matches_reference_on_cpu():
    """BASELINE cfg 1 is quoted as a CPU forward: the plain DETR (100 object queries, coco classes, ffn 2048) on one 480 x 640
    frame through the repo's modules on the host against the reference's own classes (full_cfg1_full.npz)."""
    model, out, res, feats, memory = full.run_cfg1("cpu")
    full._compare("cfg1_full", model, out, res, feats, memory, box_tol=2e-5, logit_tol=1e-4, orig=um.FULL_IMG_CFG1)


def test_cfg5_mask_head_800x1333_matches_reference_on_cpu():
    """BASELINE cfg 5 on the host (the product's nn.Modules + the library's host operator): detector outputs, mask logits and
    post-processed masks against the reference's classes (full_cfg5_full.npz)."""
    full.test_cfg5_mask_head_800x1333_matches_reference(torch.device("cpu"))


def test_cfg3_training_step_800x1333_batch2_matches_reference_on_cpu():
    """BASELINE cfg 3 on the host: losses, total and the gradient norm of every parameter of one batch-2 step at 800 x 1333
    against the reference (full_cfg3_full.npz); backward through tf_msda_backward_host_f32."""
    full.test_cfg3_training_step_800x1333_batch2_matches_reference(torch.device("cpu"))
