"""GPU (-m gpu): tf_linear_split_f32 -- nn.Linear as a split product (fp16 pieces, three MFMAs: the default; six bf16 terms; fp32
accumulation) on the matrix cores -- against a float64 reference, and the model / tracker goldens with the routes switched on.
Every test runs for both products."""
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


@pytest.fixture(params=[6, 16], ids=["six_terms", "fp16_pieces"])
def split_on(request):
    from trackformer_amd import fused
    prev = fused.set_split_linear(True)
    prev_terms = fused.set_split_terms(request.param)
    yield fused
    fused.set_split_terms(prev_terms)
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
    # the dropped terms are < 2^-24 (six bf16 terms) / < 2^-22 (fp16 pieces) of each product: bound the error by a share of
    # sum |x| |w| (+ fp32 accumulation over K)
    bound = (x.abs().double() @ w.abs().double().t()) * 2.0 ** -20 + 1e-6
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


PACKED_SHAPES = [   # M, K, N, bias, relu, row block of the weight or None, row tiles per block (0 = per shape)
    (22223, 256, 1024, True, True, None, 0),        # FFN linear1 + ReLU at the cfg-2 encoder
    (22223, 1024, 256, True, False, None, 0),       # FFN linear2
    (22223, 256, 256, True, False, None, 0),        # (shapes the dispatcher keeps on the unpacked kernel run here through
    (22223, 256, 384, True, False, None, 0),        #  the monkeypatched policy) N not a multiple of the 256-column block
    (5000, 256, 1024, True, True, None, 2),
    (5000, 1024, 256, True, False, None, 3),
    (4500, 256, 256, False, False, None, 4),
    (4100, 256, 1024, True, False, (128, 648), 0),  # a row block of a packed projection (in_proj_weight style)
    (4097, 64, 40, True, True, None, 0),            # one row past a whole block, K of one unrolled pair of slices
]


@pytest.mark.parametrize("M,K,N,bias,relu,rows,ti", PACKED_SHAPES, ids=["%dx%dx%d" % s[:3] for s in PACKED_SHAPES])
def test_packed_linear_is_bit_identical_to_the_unpacked_kernel(dev, split_on, monkeypatch, M, K, N, bias, relu, rows, ti):
    """tf_linear_packed_f32 (weight packed once in fragment order, csrc/linear_stream.hip) accumulates in the order of
    tf_linear_split_f32: same bits, for every block shape, at the edges of M and N, for row blocks of the weight."""
    from trackformer_amd import _cabi
    # every shape through the packed kernel, whatever the dispatcher would pick
    monkeypatch.setattr(split_on, "_use_packed", lambda m, k, n: split_on._packed_linear and k % 64 == 0)
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    n_out = N if rows is None else rows[1] - rows[0]
    b = torch.randn(n_out, generator=g).to(dev) if bias else None
    prev_packed = split_on.set_packed_linear(False)
    prev_ti = _cabi.lib().tf_msda_set_option(b"linear_stream_ti", ti)
    try:
        want = split_on.linear(x, w, b, relu=relu, rows=rows)
        split_on.set_packed_linear(True)
        got = split_on.linear(x, w, b, relu=relu, rows=rows)
        assert getattr(w, "_tf_packed", None) is not None     # the packed path ran
        again = split_on.linear(x, w, b, relu=relu, rows=rows)   # from the cached packed weight
    finally:
        split_on.set_packed_linear(prev_packed)
        _cabi.lib().tf_msda_set_option(b"linear_stream_ti", prev_ti)
    assert want is not None and got is not None and got.shape == (M, n_out)
    assert torch.equal(got, want) and torch.equal(again, want)
    w.mul_(2.0)                                                # an in-place update invalidates the cached image
    split_on.set_packed_linear(True)
    try:
        doubled = split_on.linear(x, w, None, rows=rows)
    finally:
        split_on.set_packed_linear(prev_packed)
    split_on.set_packed_linear(False)
    try:
        want2 = split_on.linear(x, w, None, rows=rows)
    finally:
        split_on.set_packed_linear(prev_packed)
    assert torch.equal(doubled, want2)


def test_split_linear_declines_what_it_cannot_do(dev, split_on):
    x = torch.randn(10, 48, device=dev)
    assert split_on.linear(x, torch.randn(8, 48, device=dev)) is None             # K % 32 != 0
    assert split_on.linear(x.double(), torch.randn(8, 48, device=dev).double()) is None
    from trackformer_amd import _cabi
    one = torch.zeros(64, device=dev)
    rc = _cabi.lib().tf_linear_split_f32(one.data_ptr(), one.data_ptr(), one.data_ptr(), 0, 0, 0, one.data_ptr(), 1, 48, 1, 0, 0)
    assert rc == -2
    rc = _cabi.lib().tf_linear_split_f32(0, one.data_ptr(), one.data_ptr(), 0, 0, 0, one.data_ptr(), 1, 32, 1, 0, 0)
    assert rc == -1
    rc = _cabi.lib().tf_linear_split_f32(one.data_ptr(), one.data_ptr(), one.data_ptr(), one.data_ptr() + 2, 0, 0, one.data_ptr(), 1, 32, 1, 0, 0)
    assert rc == -2                                                               # misaligned lo piece
    rc = _cabi.lib().tf_linear_split_f32(one.data_ptr(), one.data_ptr(), one.data_ptr(), one.data_ptr(), one.data_ptr(), 0, one.data_ptr(), 1, 32, 1, 0, 0)
    assert rc == -2                                                               # fp16 pieces (w_scale given) take two pieces, not three
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


def test_packed_linear_dispatch(dev, split_on):
    """Only the shapes the packed kernel measured faster on go to it (trackformer_amd/fused.py: _use_packed)."""
    assert split_on._use_packed(22223, 256, 1024) and split_on._use_packed(22223, 1024, 256)
    six = split_on.split_terms() == 6     # (the fp16 pieces store two weight pieces: the three-term policy)
    assert split_on._use_packed(22223, 256, 256) == six        # six terms: every many-row shape with K >= 256
    assert not split_on._use_packed(22223, 256, 384)           # the second 256-column block would be half empty
    assert not split_on._use_packed(66800, 64, 256)            # a short K: the block kernel
    assert not split_on._use_packed(400, 256, 1024)            # decoder: few rows
    assert not split_on._use_packed(30000, 288, 1024)          # hidden 288: K not a multiple of 64
    assert not split_on._use_packed(30000, 1024, 288)          # second 256-column block nearly empty
    prev = split_on.set_packed_linear(False)
    try:
        assert not split_on._use_packed(22223, 256, 1024)
    finally:
        split_on.set_packed_linear(prev)


# ------------------------------------------------------------------ epilogue / prefetch variants (defaults since round 3)


@pytest.mark.parametrize("M,K,N,bias,relu", SHAPES + [(66800, 64, 256, True, True), (16700, 512, 128, True, True)],
                         ids=["%dx%dx%d" % s[:3] for s in SHAPES] + ["conv_layer1", "conv_layer2"])
def test_buffer_store_epilogue_and_residual(dev, split_on, M, K, N, bias, relu):
    """The epilogue stores through a buffer resource (no per-store branch, no vmcnt(0) between stores; rows >= M / columns >= N
    dropped by the bounds check): nothing is written past Y; the residual epilogue equals the plain kernel + add bit for bit."""
    from trackformer_amd import _cabi
    lib = _cabi.lib()
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    r = torch.randn(M, N, generator=g).to(dev)
    got = split_on.linear(x, w, b, relu=relu)
    got_res = split_on.linear(x, w, b, relu=relu, residual=r)
    plain = split_on.linear(x, w, b, relu=False) + r
    assert torch.equal(got_res, plain.clamp_min(0) if relu else plain)
    guard = torch.full((M + 300, N), 7.0, device=dev)       # the kernel writes into the first M rows of a larger buffer
    hi, mid, lo, wsc = split_on._split_weight(w)
    rc = lib.tf_linear_split_f32(x.data_ptr(), hi.data_ptr(), mid.data_ptr(), 0 if lo is None else lo.data_ptr(),
                                 0 if wsc is None else wsc.data_ptr(),
                                 0 if b is None else b.data_ptr(), guard.data_ptr(), M, K, N, 1 if relu else 0,
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(guard[:M], got)
    assert torch.equal(guard[M:], torch.full((300, N), 7.0, device=dev))


@pytest.mark.parametrize("M,K,N,bias,relu", [(400, 256, 256, True, False), (400, 256, 1024, True, True), (400, 1024, 256, True, False),
                                              (800, 288, 288, True, False), (800, 1152, 288, False, False), (400, 256, 384, True, False),
                                              (400, 96, 256, True, True)],
                         ids=lambda v: str(v))
def test_few_rows_linear_kernels_agree(dev, split_on, monkeypatch, M, K, N, bias, relu):
    """<= 4096 rows: the ring-of-8-slices kernel (K / 32 in {8, 9, 32, 36}) or the 64 x 64 block kernel; same products in the
    same order as the packed kernel (bit-identical where K % 64 == 0), within the split product's bound of float64."""
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    got = split_on.linear(x, w, b, relu=relu)
    ref = x.double() @ w.double().t() + (b.double() if bias else 0)
    ref = ref.clamp_min(0) if relu else ref
    bound = (x.abs().double() @ w.abs().double().t()) * 2.0 ** -20 + 1e-6
    assert bool(((got.double() - ref).abs() <= bound).all())
    if K % 64 == 0:
        monkeypatch.setattr(split_on, "_use_packed", lambda m, k, n: True)
        assert torch.equal(split_on.linear(x, w, b, relu=relu), got)


@pytest.mark.parametrize("rows,n", [(22223, 384), (400, 384), (400, 512), (5000, 256)])
def test_linear_with_add_prologue(rows, n):
    """tf_linear_split_add_f32 (fused.linear_add): (x + pos) @ w^T + b with the add done while the tile is staged; the same
    bits as adding first and calling fused.linear."""
    from trackformer_amd import fused
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + n)
    x = torch.randn(1, rows, 256, generator=g).to(dev)
    pos = torch.randn(1, rows, 256, generator=g).to(dev)
    w = (torch.randn(n, 256, generator=g) / 16).to(dev)
    b = torch.randn(n, generator=g).to(dev)
    prev = fused.set_pos_add_fused(True)
    try:
        got = fused.linear_add(x, pos, w, b)
    finally:
        fused.set_pos_add_fused(prev)
    prev_packed = fused.set_packed_linear(False)   # the reference through the same (unpacked) kernel family
    try:
        ref = fused.linear(x + pos, w, b)
    finally:
        fused.set_packed_linear(prev_packed)
    assert got is not None and ref is not None and torch.equal(got, ref)
