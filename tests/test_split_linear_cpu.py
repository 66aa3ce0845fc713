"""CPU: the numerical claim behind tf_linear_split_f32 -- every linear of the path computed as the six-term bf16 split product
(three pieces per operand, fp32 accumulation; the three-term form of rounds 2-4 -- hi.hi + hi.mid + mid.hi -- passes the same
bars and is kept here as the weaker of the two emulations although the library no longer has it), emulated
with PyTorch on the CPU, keeps the model and the tracker inside the tolerances of the CPU parity suite (boxes 2e-5, logits 1e-4,
track ids exact), while plain bf16 does not.  (The HIP kernel itself is checked on the GPU:
tests/test_linear_split_gpu.py; the full sweep over all goldens: tools/experiments/bf16_split_linear.py.)"""
import pytest
import torch
import torch.nn.functional as F

from tests import test_models_cpu as shared
from tests.test_models_cpu import host_op  # noqa: F401  (fixture: the product's host operator for CPU tensors)


def _pieces(t):
    hi = t.to(torch.bfloat16).float()
    r = t - hi
    mid = r.to(torch.bfloat16).float()
    lo = (r - mid).to(torch.bfloat16).float()
    return hi, mid, lo


def _patch(monkeypatch, passes):
    def mm(x, w_t):
        xh, xm, xl = _pieces(x)
        wh, wm, wl = _pieces(w_t)
        out = xh @ wh
        if passes >= 3:
            out = out + xh @ wm + xm @ wh
        if passes == 6:
            out = out + xm @ wm + xh @ wl + xl @ wh
        return out

    def linear(x, w, b=None):
        y = mm(x, w.t())
        return y if b is None else y + b

    def addmm_activation(bias, x, w_t, *, beta=1, alpha=1, use_gelu=False):
        y = mm(x, w_t) + bias
        return F.gelu(y) if use_gelu else torch.relu(y)

    monkeypatch.setattr(F, "linear", linear)
    monkeypatch.setattr(torch, "_addmm_activation", addmm_activation)


def test_weight_pieces_reconstruct_the_weight():
    """bf16 (hi, mid): to 16 bits; (hi, mid, lo): exactly; the default, fp16 (hi, lo) of the channel-scaled weight: to 2^-23."""
    from trackformer_amd import fused
    w = torch.randn(64, 96) * 3
    assert fused.split_terms() == 16     # the package's default product
    prev6 = fused.set_split_terms(6)
    hi, mid, lo, none = fused._split_weight(w)
    assert hi.dtype == mid.dtype == lo.dtype == torch.bfloat16 and none is None
    rel = ((hi.float() + mid.float() - w).abs() / w.abs().clamp_min(1e-30)).max()
    assert float(rel) < 2.0 ** -15
    assert torch.equal(hi.double() + mid.double() + lo.double(), w.double())
    assert fused._split_weight(w)[0] is hi      # cached per tensor version
    import pytest
    with pytest.raises(ValueError, match="removed in round 5"):   # the three-term bf16 mode
        fused.set_split_terms(3)
    w.add_(1.0)
    assert fused._split_weight(w)[0] is not hi  # an in-place update invalidates the cache entry
    # fp16 pieces: w t_n = wh + wl to 2^-23 of it, scale = 16 / t_n, the largest |w t_n| of a row in [2^13, 2^14)
    w[3] *= 1e-4
    w[5] *= 300.0
    prev = fused.set_split_terms(16)
    try:
        wh, wl, none, sc = fused._split_weight(w)
        assert wh.dtype == wl.dtype == torch.float16 and none is None and sc.dtype == torch.float32 and sc.shape == (64,)
        t = 16.0 / sc.double()
        assert torch.equal(torch.log2(t), torch.log2(t).round())                      # powers of two
        scaled = w.double() * t[:, None]
        amax = scaled.abs().amax(1)
        assert bool(((amax >= 2.0 ** 13) & (amax < 2.0 ** 14)).all())
        err = (wh.double() + wl.double() - scaled).abs()
        assert bool((err <= scaled.abs() * 2.0 ** -22 + 2.0 ** -25).all())
        assert fused._split_weight(w)[0] is wh
    finally:
        fused.set_split_terms(prev)
    assert fused._split_weight(w)[0].dtype == torch.bfloat16
    assert fused.set_split_terms(prev6) == 6 and fused._split_weight(w)[0].dtype == torch.float16


@pytest.mark.parametrize("passes", [3, 6])
def test_split_product_linears_keep_model_and_tracker_parity(host_op, monkeypatch, passes):
    _patch(monkeypatch, passes=passes)
    case = "cfg2_deformable_tracking"
    model, out, res, feats = shared.run_case(case, device="cpu")
    # the CPU suite's own (tight) tolerances, 50x below the 1e-3 bar
    shared.compare_to_golden(case, model, out, res, feats, box_tol=2e-5, logit_tol=1e-4)
    tracker, rows, active, inactive = shared.run_tracker(False, device="cpu")
    shared.compare_tracker_to_golden(False, tracker, rows, active, inactive, box_tol_px=0.05)


def test_plain_bf16_linears_do_not(host_op, monkeypatch):
    _patch(monkeypatch, passes=1)
    case = "cfg2_deformable_tracking"
    model, out, res, feats = shared.run_case(case, device="cpu")
    with pytest.raises(AssertionError):   # one bf16 pass: boxes off by ~2e-4, ten times the tolerance above
        shared.compare_to_golden(case, model, out, res, feats, box_tol=2e-5, logit_tol=1e-4)


def test_split_pieces_are_cached_on_the_tensor_not_by_address():
    """Two different weights that happen to live at the same address one after the other (the caching allocator hands a
    freed parameter's memory to the next model) must not share split pieces."""
    import torch
    from trackformer_amd import fused
    prev = fused.set_split_terms(6)      # bf16 pieces: the hi piece is simply the weight rounded to bf16
    a = torch.randn(8, 32)
    hi_a = fused._split_weight(a)[0].clone()
    b = torch.empty(0)
    b.set_(a.untyped_storage(), 0, a.shape, a.stride())     # a second tensor object over the same memory
    with torch.no_grad():
        a.mul_(0).add_(torch.randn(8, 32))                  # new values at the old address
    hi_b = fused._split_weight(b)[0]
    fused.set_split_terms(prev)
    assert torch.equal(hi_b, b.to(torch.bfloat16)) and not torch.equal(hi_b, hi_a)


def test_packed_kernel_dispatch_policy():
    """Which call shapes go to tf_linear_packed_f32 (fused._use_packed): many rows, K % 64 == 0, a wide output or a long K,
    256-column blocks mostly full -- the FFN linears of the hidden-256 encoder, nothing else."""
    from trackformer_amd import fused
    prev = fused.set_packed_linear(True)
    try:
        assert fused._use_packed(22223, 256, 1024) and fused._use_packed(22223, 1024, 256)
        assert not fused._use_packed(22223, 256, 384) and not fused._use_packed(66800, 64, 256)
        p = fused.set_split_terms(16)     # fp16 pieces (the default; two stored weight pieces): a wide output or a long K only
        try:
            assert not fused._use_packed(22223, 256, 256) and fused._use_packed(22223, 256, 1024) and fused._use_packed(16700, 512, 128)
        finally:
            fused.set_split_terms(p)
        p6 = fused.set_split_terms(6)
        try:
            assert fused._use_packed(22223, 256, 256) and fused._use_packed(66800, 256, 64)   # six terms: K >= 256
        finally:
            fused.set_split_terms(p6)
        assert not fused._use_packed(400, 256, 1024)       # decoder: few rows
        assert not fused._use_packed(30000, 288, 1024)     # hidden 288: K not a multiple of 64
        assert not fused._use_packed(30000, 1024, 288)     # second 256-column block nearly empty
        fused.set_packed_linear(False)
        assert not fused._use_packed(22223, 256, 1024)
    finally:
        fused.set_packed_linear(prev)
