"""GPU (-m gpu): the full per-frame path on the MI355X (ResNet-50 -> deformable encoder/decoder with
the HIP MSDeformAttn -> heads -> Tracker) against goldens from the reference CPU path.

Tolerances are north_star's: boxes / logits within 1e-3 (fp32), track-id assignment bit-exact.
"""
import pytest
import torch

from tests import test_models_cpu as shared
from tests import util_models as um

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from trackformer_amd import _cabi
    _cabi.lib()
    return torch.device("cuda:0")


@pytest.mark.parametrize("case", list(um.MODEL_CASES))
def test_model_forward_matches_reference_cpu_path(dev, case):
    model, out, res, feats = shared.run_case(case, device=dev)
    shared.compare_to_golden(case, model, out, res, feats, box_tol=1e-3, logit_tol=1e-3)


@pytest.mark.parametrize("reid", [False, True], ids=["default", "reid"])
def test_tracker_track_ids_bit_exact(dev, reid):
    tracker, rows, active, inactive = shared.run_tracker(reid, device=dev)
    shared.compare_tracker_to_golden(reid, tracker, rows, active, inactive, box_tol_px=0.64)


def test_tracker_step_does_one_device_to_host_sync_per_frame(dev):
    """The association logic runs on one packed host copy per frame (tracker)."""
    from trackformer_amd import config, factory
    from trackformer_amd.tracker import Tracker
    model, post, args = um.build("cfg2_deformable_tracking", factory.build_model,
                                 config.make_args, device=dev)
    model.to(dev).tracking()
    tracker = Tracker(model, post, config.tracker_cfg(), False)
    tracker.reset()
    frames = um.tracker_sequence()
    with torch.no_grad():
        tracker.step(frames[0])
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")   # any further implicit sync raises
        try:
            orig_cpu = torch.Tensor.cpu
            calls = []

            def counting_cpu(self, *a, **k):
                if self.is_cuda:
                    calls.append(tuple(self.shape))
                    torch.cuda.set_sync_debug_mode("default")
                    try:
                        return orig_cpu(self, *a, **k)
                    finally:
                        torch.cuda.set_sync_debug_mode("error")
                return orig_cpu(self, *a, **k)
            torch.Tensor.cpu = counting_cpu
            tracker.step(frames[1])
        finally:
            torch.Tensor.cpu = orig_cpu
            torch.cuda.set_sync_debug_mode("default")
    assert len(calls) == 1 and calls[0][1] == 6, calls
