"""GPU (-m gpu): tf_linear_split_f32 -- nn.Linear as a bf16 split product (hi.hi + hi.mid + mid.hi, fp32 accumulate) on
the matrix cores -- against a float64 reference, and the model / tracker goldens with the opt-in switched on."""
import pytest
import torch

from tests import test_models_cpu as shared

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from trackformer_amd import _cabi
    _cabi.lib()
    return torch.device("cuda:0")


@pytest.fixture()
def split_on():
    from trackformer_amd import fused
    prev = fused.set_split_linear(True)
    yield fused
    fused.set_split_linear(prev)


SHAPES = [   # M, K, N, bias, relu
    (22223, 256, 256, True, False),     # value_proj / output_proj at the cfg-2 encoder
    (22223, 256, 384, True, False),     # offsets + attention logits in one GEMM
    (5000, 256, 1024, True, True),      # FFN linear1 + ReLU
    (5000, 1024, 256, True, False),     # FFN linear2
    (400, 256, 256, False, False),      # decoder, no bias
    (333, 288, 96, True, True),         # hidden 288 (cfg 4), M and N not multiples of the 128 x 128 block
    (1, 32, 1, True, False),
]


@pytest.mark.parametrize("M,K,N,bias,relu", SHAPES, ids=["%dx%dx%d" % s[:3] for s in SHAPES])
def test_split_linear_matches_float64(dev, split_on, M, K, N, bias, relu):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    y = split_on.linear(x, w, b, relu=relu)
    assert y is not None and y.shape == (M, N) and y.dtype == torch.float32
    ref = x.double() @ w.double().t()
    if bias:
        ref = ref + b.double()
    if relu:
        ref = ref.clamp_min(0)
    # the dropped terms are < 2^-16 of each product: bound the error by that share of sum |x| |w| (+ fp32 accumulation)
    bound = (x.abs().double() @ w.abs().double().t()) * 2.0 ** -15 + 1e-6
    err = (y.double() - ref).abs()
    assert bool((err <= bound).all()), float((err - bound).max())
    # and it is far closer to fp32 than plain bf16 would be
    plain = (x.to(torch.bfloat16).float() @ w.to(torch.bfloat16).float().t()).double()
    if bias:
        plain = plain + b.double()
    if relu:
        plain = plain.clamp_min(0)
    if M * N > 1000:
        assert float(err.max()) < 0.05 * float((plain - ref).abs().max())


def test_split_linear_declines_what_it_cannot_do(dev, split_on):
    x = torch.randn(10, 48, device=dev)
    assert split_on.linear(x, torch.randn(8, 48, device=dev)) is None             # K % 32 != 0
    assert split_on.linear(x.double(), torch.randn(8, 48, device=dev).double()) is None
    from trackformer_amd import _cabi
    one = torch.zeros(64, device=dev)
    rc = _cabi.lib().tf_linear_split_f32(one.data_ptr(), one.data_ptr(), one.data_ptr(), 0, one.data_ptr(), 1, 48, 1, 0, 0)
    assert rc == -2
    rc = _cabi.lib().tf_linear_split_f32(0, one.data_ptr(), one.data_ptr(), 0, one.data_ptr(), 1, 32, 1, 0, 0)
    assert rc == -1
    split_on.set_split_linear(False)
    assert split_on.linear(torch.randn(4, 32, device=dev), torch.randn(4, 32, device=dev)) is None   # switched off


def test_model_and_tracker_goldens_hold_with_split_linears(dev, split_on):
    """The whole path with the encoder / decoder linears as split products: boxes / logits within 1e-3 of the
    reference CPU path, track ids exact (what tools/experiments/bf16_split_linear.py predicts from the CPU)."""
    case = "cfg2_deformable_tracking"
    model, out, res, feats = shared.run_case(case, device=dev)
    shared.compare_to_golden(case, model, out, res, feats, box_tol=1e-3, logit_tol=1e-3)
    tracker, rows, active, inactive = shared.run_tracker(False, device=dev)
    shared.compare_tracker_to_golden(False, tracker, rows, active, inactive, box_tol_px=0.64)
