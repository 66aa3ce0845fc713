"""GPU: fused element-wise HIP kernels against the plain PyTorch fp32 formulation they replace."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(1, 64, 50, 84), (2, 256, 13, 21), (1, 2048, 25, 42),
                                   (1, 8, 3, 5), (1, 256, 200, 334)])
@pytest.mark.parametrize("with_res", [False, True])
@pytest.mark.parametrize("relu", [False, True])
def test_bias_act_matches_torch(dev, shape, with_res, relu):
    from trackformer_amd import fused
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn(shape[1], generator=g).to(dev)
    r = torch.randn(shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last) \
        if with_res else None
    ref = x + b.view(1, -1, 1, 1)
    if r is not None:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    y = fused.bias_act_(x, b, r, relu)
    assert y is x
    assert torch.allclose(x, ref, atol=1e-6, rtol=1e-6)
    assert x.is_contiguous(memory_format=torch.channels_last)


def test_bias_act_declines_unsuitable_layouts(dev):
    from trackformer_amd import fused
    x = torch.randn(1, 6, 4, 4, device=dev).contiguous(memory_format=torch.channels_last)
    assert fused.bias_act_(x, torch.zeros(6, device=dev)) is None          # C % 4 != 0
    x = torch.randn(1, 8, 4, 4, device=dev)                                  # NCHW storage
    assert fused.bias_act_(x, torch.zeros(8, device=dev)) is None
    assert fused.bias_act_(torch.randn(1, 8, 4, 4), torch.zeros(8)) is None  # CPU


@pytest.mark.parametrize("rows,C", [(22223, 256), (400, 256), (800, 288), (7, 1024), (3, 4096),
                                    (5, 8)])
@pytest.mark.parametrize("with_res", [False, True])
def test_add_layernorm_matches_torch(dev, rows, C, with_res):
    from trackformer_amd import fused
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(1, rows, C, generator=g) * 3 + 0.5).to(dev)
    r = torch.randn(1, rows, C, generator=g).to(dev) if with_res else None
    ln = torch.nn.LayerNorm(C).to(dev)
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.2)
        ln.bias.normal_(0.0, 0.2)
        ref = ln(x + r if with_res else x)
        out = fused.add_layernorm(x, r, ln)
    assert out is not None and out.shape == x.shape
    assert torch.allclose(out, ref, atol=2e-5, rtol=1e-5)


def test_residual_norm_dispatch(dev):
    from trackformer_amd import fused
    ln = torch.nn.LayerNorm(16).to(dev)
    x, r = torch.randn(2, 3, 16, device=dev), torch.randn(2, 3, 16, device=dev)
    with torch.no_grad():
        a = fused.residual_norm(x, r, ln, inference=True)
        b = fused.residual_norm(x, r, ln, inference=False)
    assert torch.allclose(a, b, atol=1e-5)
    # CPU tensors always take the PyTorch formulation
    lc = torch.nn.LayerNorm(16)
    assert fused.add_layernorm(torch.randn(2, 16), None, lc) is None


@pytest.mark.parametrize("n,length,heads,d,masked", [(1, 400, 8, 32, False), (1, 800, 8, 36, False),
                                                     (2, 77, 8, 32, True), (1, 7, 4, 64, False), (1, 1030, 8, 32, True)])
@pytest.mark.parametrize("mfma", [1, 2, 0], ids=["matrix_cores", "matrix_cores_lds_staged", "vector"])
def test_mha_core_matches_torch_reference(dev, n, length, heads, d, masked, mfma):
    """tf_mha_core_f32 (decoder query self-attention, deformable_transformer.py:364-383) against the plain fp32
    formulation softmax(q k^T / sqrt(d)) v evaluated in float64: the fp32 matrix-core kernel (the default since round 5)
    and the vector kernel it replaced."""
    from trackformer_amd import _cabi, fused
    prev = _cabi.lib().tf_msda_set_option(b"mha_mfma", mfma)
    try:
        _mha_case(dev, n, length, heads, d, masked)
    finally:
        _cabi.lib().tf_msda_set_option(b"mha_mfma", prev)


def _mha_case(dev, n, length, heads, d, masked):
    from trackformer_amd import fused
    g = torch.Generator().manual_seed(length)
    e = heads * d
    qk = torch.randn(n, length, 2 * e, generator=g) * 1.5
    v = torch.randn(n, length, e, generator=g)
    mask = None
    if masked:
        mask = torch.rand(n, length, generator=g) < 0.3
        mask[:, 0] = False
    out = fused.mha_core(qk.to(dev), v.to(dev), heads, None if mask is None else mask.to(dev))
    assert out is not None
    q = qk[..., :e].double().view(n, length, heads, d).transpose(1, 2)
    k = qk[..., e:].double().view(n, length, heads, d).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / d ** 0.5
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v.double().view(n, length, heads, d).transpose(1, 2)).transpose(1, 2).reshape(n, length, e)
    assert torch.allclose(out.cpu().double(), ref, atol=2e-5, rtol=1e-4)


def test_decoder_self_attention_uses_own_kernel_and_matches_module(dev):
    """DeformableTransformerDecoderLayer's inference self-attention (q/k GEMM + tf_mha_core_f32 + out_proj)
    against nn.MultiheadAttention itself on the same weights."""
    from trackformer_amd import fused
    from trackformer_amd.deformable_transformer import DeformableTransformerDecoderLayer
    torch.manual_seed(0)
    layer = DeformableTransformerDecoderLayer(256, 1024, 0.1, "relu", 4, 8, 4).to(dev).eval()
    tgt = torch.randn(1, 400, 256, device=dev)
    pos = torch.randn(1, 400, 256, device=dev)
    calls = []
    orig = fused.mha_core

    def spy(*a, **k):
        r = orig(*a, **k)
        calls.append(r is not None)
        return r
    fused.mha_core = spy
    try:
        with torch.no_grad():
            got, _ = layer._self_attention_inference(tgt + pos, tgt, None)
    finally:
        fused.mha_core = orig
    assert calls == [True]
    with torch.no_grad():
        q = (tgt + pos).transpose(0, 1)
        ref = layer.self_attn(q, q, tgt.transpose(0, 1))[0].transpose(0, 1)
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-4)




@pytest.mark.parametrize("shape,res,relu", [((1, 256, 200, 334), True, True), ((1, 64, 200, 334), False, True), ((2, 512, 25, 42), True, False)])
def test_bias_act_matches_torch_bit_for_bit(shape, res, relu):
    """tf_bias_act_f32 (two grid strides per iteration, loads first): (x + bias) (+ residual) (ReLU) in that order."""
    from trackformer_amd import fused
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(*shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn(shape[1], generator=g).to(dev)
    r = torch.randn(*shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last) if res else None
    base = x + b.view(1, -1, 1, 1)
    if r is not None:
        base = base + r
    if relu:
        base = base.clamp_min(0)
    got = x.clone()
    assert fused.bias_act_(got, b, r, relu) is not None
    assert torch.equal(got, base)


@pytest.mark.parametrize("rows,ti,d", [(22223, 3, 256), (4100, 2, 256), (97, 1, 256), (22223, 2, 288), (150, 1, 288)])
def test_fused_ffn_block(rows, ti, d, monkeypatch):
    """tf_ffn_fused_f32 (fused.ffn): linear1 -> ReLU -> linear2 -> + residual in one launch equals the separate packed
    linears bit for bit (same split, same order of the matrix-core sums); with the LayerNorm in the epilogue it equals
    torch's LayerNorm of that up to rounding."""
    from trackformer_amd import _cabi, fused
    lib = _cabi.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(rows)
    l1, l2, norm = torch.nn.Linear(d, 1024).to(dev), torch.nn.Linear(1024, d).to(dev), torch.nn.LayerNorm(d).to(dev)
    with torch.no_grad():
        norm.weight.add_(0.1 * torch.randn(d, device=dev))
        norm.bias.add_(0.1 * torch.randn(d, device=dev))
    x = torch.randn(1, rows, d, device=dev)
    monkeypatch.setattr(fused, "_FFN_FUSED_MIN_ROWS", 1)
    monkeypatch.setattr(fused, "_PACKED_MIN_ROWS", 0)   # the reference below through tf_linear_packed_f32 at every size
    prev_on, prev_ti = fused.set_ffn_fused(True), lib.tf_msda_set_option(b"ffn_ti", ti)
    try:
        with torch.no_grad():
            h = fused.linear(x, l1.weight, l1.bias, relu=True)
            ref = fused.linear(h, l2.weight, l2.bias) + x
            got = fused.ffn(x, l1, l2, None, residual=x)
            got_ln = fused.ffn(x, l1, l2, norm, residual=x)
            exact = l2(torch.relu(l1(x))) + x   # hipBLASLt fp32
    finally:
        fused.set_ffn_fused(prev_on)
        lib.tf_msda_set_option(b"ffn_ti", prev_ti)
    assert got is not None and got_ln is not None
    assert torch.equal(got, ref)
    assert torch.allclose(got, exact, atol=2e-4, rtol=1e-4)
    assert torch.allclose(got_ln, norm(ref), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("rows,ti,d", [(22223, 0, 256), (22223, 3, 256), (400, 0, 256), (97, 2, 256), (22223, 0, 288), (800, 0, 288)])
def test_linear_residual_layernorm(rows, ti, d, monkeypatch):
    """tf_linear_res_ln_f32 (fused.linear_residual_norm): output projection + residual + LayerNorm in one launch against
    the split-product linear followed by torch's add and LayerNorm."""
    from trackformer_amd import _cabi, fused
    lib = _cabi.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(rows + ti)
    lin, norm = torch.nn.Linear(d, d).to(dev), torch.nn.LayerNorm(d).to(dev)
    with torch.no_grad():
        norm.weight.add_(0.1 * torch.randn(d, device=dev))
        norm.bias.add_(0.1 * torch.randn(d, device=dev))
    x, res = torch.randn(1, rows, d, device=dev), torch.randn(1, rows, d, device=dev)
    monkeypatch.setattr(fused, "_LINLN_MIN_ROWS", 1)
    prev_on, prev_ti = fused.set_linear_ln_fused(True), lib.tf_msda_set_option(b"linln_ti", ti)
    try:
        with torch.no_grad():
            got = fused.linear_residual_norm(x, lin, res, norm)
            ref = norm(res + fused.linear(x, lin.weight, lin.bias))
    finally:
        fused.set_linear_ln_fused(prev_on)
        lib.tf_msda_set_option(b"linln_ti", prev_ti)
    assert got is not None
    assert torch.allclose(got, ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("shape", [(1, 64, 400, 667), (2, 64, 33, 20), (1, 8, 7, 9)])
def test_bias_relu_maxpool_bit_identical(shape):
    """tf_bias_relu_maxpool_f32 (stem route): one pass instead of bias_act_ + MaxPool2d(3, 2, 1); same bits."""
    from trackformer_amd import fused
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randn(*shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn(shape[1], generator=g).to(dev)
    ref = torch.nn.functional.max_pool2d(torch.relu(x + b.view(1, -1, 1, 1)), 3, 2, 1)
    got = fused.bias_relu_maxpool(x, b)
    assert got is not None and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("shape", [(1, 3, 800, 1333), (2, 3, 97, 130), (1, 3, 9, 7)])
def test_stem_convolution_split(shape):
    """tf_stem_conv7x7_f32 (fused.stem_conv): the 7 x 7 / stride 2 stem convolution as a split product against the library
    convolution (1e-3 of the output scale: three-term bf16 products), channels_last output, optional shift + ReLU."""
    from trackformer_amd import fused
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(shape[2])
    x = torch.randn(*shape, generator=g).to(dev)
    w = (torch.randn(64, 3, 7, 7, generator=g) / 12).to(dev)
    b = torch.randn(64, generator=g).to(dev)
    prev = fused.set_stem_conv_split(True)
    try:
        got = fused.stem_conv(x, w)
        got_b = fused.stem_conv(x, w, b, relu=True)
    finally:
        fused.set_stem_conv_split(prev)
    ref = torch.nn.functional.conv2d(x, w, None, stride=2, padding=3)
    assert got is not None and got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 1e-3 * scale
    assert float((got_b - torch.relu(ref + b.view(1, -1, 1, 1))).abs().max()) < 1e-3 * scale


def test_fp16_product_range_contract_is_loud_and_has_a_way_out(dev):
    """VERDICT r04 weak #9 / ADVICE r04 (medium).  An activation beyond 65504 * 16 under the fp16 split product (the default):
    (1) the output row is NaN -- also through a ReLU epilogue, which used to map it to zero; (2) the debug check names the
    operation; (3) audit_activation_range() routes exactly that layer through the six-term product, after which the layer is
    fp32-accurate again while every other layer keeps the fp16 product."""
    from trackformer_amd import fused
    prev_split, prev_terms = fused.set_split_linear(True), fused.set_split_terms(16)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(512, 256, generator=g).to(dev)
    x[17, 3] = 4.0e6
    lin = torch.nn.Linear(256, 256).to(dev)
    other = torch.nn.Linear(256, 256).to(dev)
    ref = F.relu(F.linear(x.double(), lin.weight.double(), lin.bias.double())).float()
    try:
        with torch.no_grad():
            y = fused.linear(x, lin.weight, lin.bias, relu=True)
            assert y is not None and torch.isnan(y[17]).all() and torch.isfinite(y[torch.arange(512, device=dev) != 17]).all()
            prev_check = fused.set_check_finite(True)
            try:
                with pytest.raises(FloatingPointError, match="linear produced a non-finite result"):
                    fused.linear(x, lin.weight, lin.bias, relu=True)
                assert fused.linear(x[:16], lin.weight, lin.bias, relu=True) is not None      # in-range rows pass the check
            finally:
                fused.set_check_finite(prev_check)
            with fused.audit_activation_range() as report:
                fused.linear(x, lin.weight, lin.bias, relu=True)
                fused.linear(x[:16], other.weight, other.bias)
            assert report["routed"] == 1 and report["largest"] == 4.0e6 and fused.six_term_routes() == 1
            assert report["layers"][0][0] == "linear" and report["layers"][0][2] == 4.0e6
            y = fused.linear(x, lin.weight, lin.bias, relu=True)                                # now six terms for this weight
            assert torch.isfinite(y).all()
            assert float((y - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
            assert fused.split_terms() == 16                                                    # the process default is untouched
            z = fused.linear(x[:16], other.weight, other.bias)
            assert float((z - F.linear(x[:16], other.weight, other.bias)).abs().max()) < 1e-4
    finally:
        fused.route_six_terms(lin.weight, False)
        fused.set_split_linear(prev_split)
        fused.set_split_terms(prev_terms)
    assert fused.six_term_routes() == 0


@pytest.mark.parametrize("classes,clip", [(1, True), (20, True), (3, False)])
def test_postprocess_pack_equals_the_module_chain(dev, classes, clip):
    """tf_postprocess_pack_f32 (round 6: what Tracker.step_async enqueues after the detector for a model without a mask head)
    against the chain it replaces ON THE DEVICE -- DeformablePostProcess.forward, clip_boxes_to_image, the stacking -- boxes bit
    for bit, labels equal (ties: the first class), scores to an ulp of the exponential; and the switch."""
    from trackformer_amd import fused
    from trackformer_amd.box_ops import clip_boxes_to_image
    from trackformer_amd.deformable_detr import DeformablePostProcess
    g = torch.Generator().manual_seed(classes)
    q, h, w = 400, 1080, 1920
    logits = torch.randn(1, q, classes, generator=g) * 3
    if classes > 1:
        logits[0, :7, 1] = logits[0, :7, 0]
        logits[0, 7:12] = 40.0
    boxes = torch.rand(1, q, 4, generator=g)
    boxes[0, :20, 2:] *= 3
    logits, boxes = logits.to(dev), boxes.to(dev)
    res = DeformablePostProcess()({'pred_logits': logits, 'pred_boxes': boxes}, torch.tensor([[h, w]], device=dev))[0]
    want_boxes = clip_boxes_to_image(res['boxes'], (h, w)) if clip else res['boxes']
    got = fused.postprocess_pack(logits[0], boxes[0], h, w, clip)
    assert got is not None and got.shape == (q, 6)
    assert torch.equal(got[:, :4], want_boxes)
    assert torch.equal(got[:, 5].long(), res['labels'])
    assert torch.allclose(got[:, 4], res['scores'], rtol=3e-7, atol=0)
    prev = fused.set_postprocess_fused(False)
    try:
        assert fused.postprocess_pack(logits[0], boxes[0], h, w, clip) is None
    finally:
        fused.set_postprocess_fused(prev)


@pytest.mark.parametrize("shape,out_size,qpi", [((6, 32, 25, 42), (50, 84), 3), ((4, 64, 50, 84), (100, 167), 4), ((2, 16, 100, 167), (200, 334), 1)])
def test_upsample_add_equals_interpolate_plus_add(dev, shape, out_size, qpi):
    """tf_upsample_add_nhwc_f32 (round 6, the mask head's FPN merge) against detr_segmentation.MaskHeadSmallConv._merge on the
    device, bit for bit (84 -> 167 columns: the nearest index is not x / 2)."""
    from trackformer_amd import fused
    from trackformer_amd.detr_segmentation import MaskHeadSmallConv
    g = torch.Generator().manual_seed(shape[1])
    low = torch.randn(*shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    fpn = torch.randn(shape[0] // qpi, shape[1], *out_size, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    want = MaskHeadSmallConv._merge(low, fpn, qpi)
    got = fused.upsample_add(low, fpn, qpi)
    assert got is not None and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)


@pytest.mark.parametrize("c,groups,hw,n", [(16, 8, (200, 334), 5), (32, 8, (37, 65), 3), (16, 8, (9, 37), 2)])
def test_groupnorm_relu_conv_to_one_channel(dev, c, groups, hw, n):
    """tf_groupnorm_relu_conv3x3_c1_nhwc_f32 (round 6, the end of the mask head: out_lay(relu(gn5(x)))) against the modules in
    float64 on the device, and against the library's own GroupNorm kernel + fp32 convolution at the same tolerance."""
    from trackformer_amd import fused
    g = torch.Generator().manual_seed(c + groups)
    x = (torch.randn(n, c, *hw, generator=g) * 2 + 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    gn = torch.nn.GroupNorm(groups, c).to(dev)
    conv = torch.nn.Conv2d(c, 1, 3, padding=1).to(dev)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(c, generator=g) * 0.2)
        want = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.group_norm(x.double(), groups, gn.weight.double(), gn.bias.double(), gn.eps)),
                                          conv.weight.double(), conv.bias.double(), padding=1)
        lib32 = conv(torch.relu(gn(x)))
        got = fused.groupnorm_relu_conv3x3_c1(x, gn, conv)
    assert got is not None and got.shape == (n, 1, *hw)
    scale = float(want.abs().max())
    err, err_lib = float((got.double() - want).abs().max()) / scale, float((lib32.double() - want).abs().max()) / scale
    print("gn + relu + conv %d -> 1 at %s: max err / max |y| %.2e (library fp32 modules: %.2e)" % (c, hw, err, err_lib))
    assert err < 2e-6 and err < 4 * err_lib + 1e-7


@pytest.mark.parametrize("lowres,img,out,n", [((200, 334), (800, 1333), (1080, 1800), 100), ((25, 42), (100, 160), (67, 107), 7)])
def test_mask_label_map_equals_the_module_chain(dev, lowres, img, out, n):
    """tf_mask_label_map_f32 (round 6) against PostProcessSegm + stack + max + threshold on the device: the same ownership map up
    to pixels where the two best tracks differ by fp32 round-off; a track without a mask never owns a pixel, ties go to the
    first track."""
    from trackformer_amd import fused
    from trackformer_amd.detr_segmentation import PostProcessSegm
    g = torch.Generator().manual_seed(n)
    logits = (torch.randn(1, n, *lowres, generator=g) * 3).to(dev)
    logits[0, n - 2] = logits[0, 1]
    order = torch.randperm(n, generator=g).tolist()
    order[0] = -1
    seg = PostProcessSegm()([{}], {'pred_masks': logits}, torch.tensor([list(out)]), torch.tensor([list(img)]), return_probs=True)[0]['masks'].squeeze(1)
    probs = torch.stack([seg[r] if r >= 0 else torch.full(out, -1.0, device=dev) for r in order])
    best, owner = probs.max(dim=0)
    want = torch.where(best > 0.5, owner, torch.full_like(owner, -1)).to(torch.int16)
    got = fused.mask_label_map(logits[0].contiguous(), order, img, img, out)
    assert got is not None and got.shape == want.shape and got.dtype == torch.int16
    assert float((got != want).float().mean()) < 1e-4
    assert not bool((got == 0).any())                                                      # track 0 has no mask
    first, second = sorted((order.index(1), order.index(n - 2))) if 1 in order and (n - 2) in order else (None, None)
    if first is not None:
        assert not bool((got == second).any())                                             # equal rows: the first track wins


@pytest.mark.parametrize("n,qpi,low,out,cin,cout,norm", [(6, 3, (50, 84), (100, 167), 64, 32, True), (4, 4, (100, 167), (200, 334), 32, 16, True),
                                                          (4, 2, (25, 42), (50, 84), 128, 64, True), (2, 1, (13, 21), (25, 42), 288, 128, False)])
def test_conv3x3_merged_equals_the_pass_by_pass_chain(dev, n, qpi, low, out, cin, cout, norm):
    """tf_conv3x3_merge_packed_f32 (round 6: the mask head's FPN levels with the merge and the previous layer's GroupNorm + ReLU in the
    convolution's fetch) against the chain it replaces on the device -- fused.groupnorm_nhwc(relu), fused.upsample_add,
    fused.conv3x3 -- at the head's shapes: the same products of the same values in (possibly) another fp32 order, and against torch in float64 within the split product's bound."""
    from trackformer_amd import fused
    g = torch.Generator().manual_seed(cin + cout)
    lo = (torch.randn(n, cin, *low, generator=g) * 2 + 0.3).to(dev).contiguous(memory_format=torch.channels_last)
    fpn = torch.randn(n // qpi, cin, *out, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    taps = (torch.randn(cout, 3, 3, cin, generator=g) / (9 * cin) ** 0.5).to(dev).reshape(cout, 9 * cin).contiguous()
    b = torch.randn(cout, generator=g).to(dev)
    gn = torch.nn.GroupNorm(8, cin).to(dev)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(cin, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(cin, generator=g) * 0.2)
        ws = fused.groupnorm_stats(lo, gn) if norm else None
        got = fused.conv3x3_merged(lo, fpn, qpi, taps, b, gn=gn if norm else None, ws=ws)
        assert got is not None and got.shape == (n, cout, *out)
        act = lo
        if norm:
            z = fused.groupnorm_nhwc(lo.permute(0, 2, 3, 1).reshape(n * low[0] * low[1], cin), n, gn, relu=True)
            act = z.view(n, *low, cin).permute(0, 3, 1, 2)
        merged = fused.upsample_add(act, fpn, qpi)
        chain = fused.conv3x3(merged, taps, b, False, 1)
        ref = lo.double()
        if norm:
            ref = torch.relu(torch.nn.functional.group_norm(ref, 8, gn.weight.double(), gn.bias.double(), gn.eps))
        ref = torch.nn.functional.interpolate(ref, size=out, mode="nearest")
        ref = (ref.view(n // qpi, qpi, *ref.shape[1:]) + fpn.double()[:, None]).flatten(0, 1)
        ref = torch.nn.functional.conv2d(ref, taps.view(cout, 3, 3, cin).permute(0, 3, 1, 2).double(), b.double(), padding=1)
    scale = float(ref.abs().max())
    assert float((got - chain).abs().max()) <= 4e-6 * scale   # (the chain's convolution may cut its K loop into pieces: another fp32 order)
    assert float((got.double() - ref).abs().max()) < 3e-6 * scale
