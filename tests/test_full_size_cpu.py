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


@pytest.mark.parametrize("case", list(um.FULL_CASES))
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
