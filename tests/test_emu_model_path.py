"""CPU: the GPU INFERENCE PATH of the model -- backbone with folded FrozenBN + bias_act, input projections, the encoder
with the fused MSDeformAttn entry and the LDS-window kernel, split-product linears, fused residual + LayerNorm, the
decoder with its own attention kernel and box refinement -- executed on CPU tensors through the SIMT emulator's build of
the kernel sources (tests/util_emu_gpu_path.py), against the goldens of the reference's own classes.  Once with the
round-3 routes (the defaults since their hardware validation), once with every one of them switched off (the round-2 path)."""
import numpy as np
import pytest
import torch

from tests import emu_lib, test_models_cpu as shared
from tests.util_emu_gpu_path import gpu_path_on_emulator

pytestmark = pytest.mark.skipif(not emu_lib.available(), reason="needs a host clang++ (ROCm's llvm) to build the emulated library")


def _run(case, optin, fn=None, terms=None):
    from trackformer_amd import backbone, fused
    with gpu_path_on_emulator() as lib:
        setters = [backbone.set_conv1x1_split, backbone.set_conv3x3_split, fused.set_input_proj_fused, fused.set_box_refine_fused,
                   fused.set_ffn_fused, fused.set_linear_ln_fused, fused.set_stem_pool_fused, fused.set_pos_add_fused,
                   fused.set_stem_conv_split, fused.set_heads_split]
        prev = [(s, s(bool(optin))) for s in setters]
        # the stream form of the convolutions for EVERY shape (at the small test frame the default rule -- many output pixels
        # under >= 128 channels -- would select it nowhere)
        prev.append((fused.set_conv_stream, fused.set_conv_stream("all" if optin else False)))
        if terms is not None:
            prev.append((fused.set_split_terms, fused.set_split_terms(terms)))
        prev += [(lambda v: setattr(fused, "_LINLN_MIN_ROWS", v), fused._LINLN_MIN_ROWS),
                 (lambda v: setattr(fused, "_FFN_FUSED_MIN_ROWS", v), fused._FFN_FUSED_MIN_ROWS)]
        fused._LINLN_MIN_ROWS = 1
        fused._FFN_FUSED_MIN_ROWS = 1   # the test models have few tokens: the decoder's feed-forward blocks take it too
        on = 1 if optin else 0
        opts = {b"direct9": on}
        prev_opts = {k: lib.tf_msda_set_option(k, v) for k, v in opts.items()}
        try:
            return (fn() if fn is not None else shared.run_case(case)) + (dict(lib.calls),)
        finally:
            for setter, value in prev:
                setter(value)
            for k, v in prev_opts.items():
                lib.tf_msda_set_option(k, v)


@pytest.mark.parametrize("optin,terms", [(False, None), (True, None), (True, 6)], ids=["round2_routes", "defaults", "defaults_six_terms"])
def test_gpu_inference_path_on_the_emulator_matches_reference(optin, terms):
    """`defaults`: every route on, the package's default split product (the fp16 product, include/tf_fused.h); `defaults_six_terms`:
    the same routes with the six-term bf16 product."""
    case = "cfg2_deformable_tracking"
    model, out, res, feats, calls = _run(case, optin, terms=terms)
    # the GPU path really ran: 6 encoder + 6 decoder layers through the fused MSDeformAttn entry, the split-product linears,
    # the fused LayerNorm and bias_act passes, the own attention kernel
    assert calls.get("tf_msda_forward_fused_f32") == 12 and calls.get("tf_mha_core_f32") == 6
    # 30 LayerNorms: 12 + 18.  Opt-in: the 12 feed-forward blocks are one launch each (12 norms, 24 linears inside), and so
    # are the 18 output projections with their residual add and norm (6 encoder, 2 x 6 decoder): no separate LayerNorm left
    assert calls.get("tf_add_layernorm_f32", 0) == (0 if optin else 30) and calls.get("tf_linear_split_f32", 0) >= (18 if optin else 60)
    assert calls.get("tf_ffn_fused_f32") == (12 if optin else None) and calls.get("tf_linear_res_ln_f32") == (18 if optin else None)
    assert calls.get("tf_linear_split_add_f32") == (18 if optin else None)   # positional add inside the 6 + 2 x 6 attention projections
    routes = ("tf_conv3x3_split_f32", "tf_linear_split_res_f32", "tf_groupnorm_nhwc_f32", "tf_box_refine_f32")
    if optin:   # ResNet-50: 16 bottlenecks (their 3 x 3 and closing 1 x 1 convolutions), 3 + 1 projection levels, 6 decoder layers
        # 3 x 3 convolutions: the 16 bottlenecks' + the extra pyramid level's, the small ones with their K loop split
        # (round 4: through the stream GEMM, tf_conv_packed_f32 -- as are the strided projections of layer2..4 (3) and, at this
        # small test frame, the few-pixel 1 x 1 convolutions with a long K, their K loop split)
        assert calls.get("tf_conv_packed_f32", 0) >= 17 + 3 + 1, calls
        assert calls.get("tf_conv3x3_split_f32") is None and calls.get("tf_conv1x1_strided_split_f32") is None, calls
        assert [calls.get(r) for r in routes[1:]] == [16, 4, 6], calls
        assert calls.get("tf_bias_act_f32", 0) <= 1 and calls.get("tf_bias_relu_maxpool_f32") == 1   # the stem: shift + ReLU + pooling in one pass
        assert calls.get("tf_stem_conv7x7_f32") == 1   # ... after the 7 x 7 convolution as a split product: all 53 ResNet convolutions on own kernels
    else:
        assert all(calls.get(r) is None for r in routes + ("tf_conv3x3_splitk_f32", "tf_conv_packed_f32")), calls
        assert calls.get("tf_bias_act_f32", 0) >= 50
    # north_star's bar is 1e-3; the fp32-class split products (fp16 pieces, six bf16 terms) hold the CPU suite's own tolerances
    shared.compare_to_golden(case, model, out, res, feats, box_tol=2e-5 if optin else 2e-4, logit_tol=1e-4 if optin else 1e-3)
    z = np.load(shared.os.path.join(shared.GOLDEN, "model_%s.npz" % case))
    print("max |d boxes| %.2e  max |d logits| %.2e" % (
        float(np.abs(out['pred_boxes'].numpy() - z['pred_boxes']).max()),
        float(np.abs(out['pred_logits'].numpy() - z['pred_logits']).max())))


def test_multi_frame_model_hidden_288_on_the_emulator():
    """cfg 4's model family (hidden 288: head dimension 36, two frames x 4 levels in the decoder, GroupNorm with 9 channels
    per group) through the GPU path with every opt-in route: msda_fwd_f32_pquad<.., 36>, msda_fwd_f32_direct9, the
    K = 288 / 1152 deep-prefetch linears, ffn_fused_kernel<288, ..> / linear_res_ln_kernel<288, ..>."""
    case = "cfg4_multi_frame_tracking"
    model, out, res, feats, calls = _run(case, True)
    shared.compare_to_golden(case, model, out, res, feats, box_tol=2e-5, logit_tol=1e-4)
    assert calls.get("tf_msda_forward_fused_f32", 0) >= 12 and calls.get("tf_groupnorm_nhwc_f32", 0) >= 3
    # hidden 288: the one-launch feed-forward blocks and projection + norm launches in their three-wave geometry
    assert calls.get("tf_ffn_fused_f32", 0) >= 12 and calls.get("tf_linear_res_ln_f32", 0) >= 18 and calls.get("tf_add_layernorm_f32", 0) == 0


def test_tracker_sequence_ids_on_the_emulator_with_every_opt_in_route():
    """Six frames of Tracker.step through the emulated GPU path, all opt-in routes on, the default split product: track ids /
    frames / source queries equal the reference golden bit for bit (the decisions hang on scores next to thresholds)."""
    tracker, rows, active, inactive, calls = _run("cfg2_deformable_tracking", True, fn=lambda: shared.run_tracker(False))
    shared.compare_tracker_to_golden(False, tracker, rows, active, inactive, box_tol_px=0.05)
    assert calls.get("tf_conv_packed_f32", 0) >= (17 + 3) * 6 and calls.get("tf_box_refine_f32") == 36


def test_mask_head_through_the_split_product_convolutions():
    """MaskHeadSmallConv's GPU inference route (detr_segmentation.py: lay2 .. lay5 through fused.conv3x3 on channels_last
    activations -- lay2's 264 input channels padded to 288 --, their GroupNorms through tf_groupnorm_nhwc_f32) against the same
    module's PyTorch path (reference: models/detr_segmentation.py:105-157), hidden 256 + 8 heads, three queries."""
    from trackformer_amd import detr_segmentation as ds
    torch.manual_seed(0)
    head = ds.MaskHeadSmallConv(264, [1024, 512, 256], 256).eval()
    for m in head.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.normal_(m.bias, 0, 0.1)
        if isinstance(m, torch.nn.GroupNorm):
            torch.nn.init.normal_(m.weight, 1, 0.2)
            torch.nn.init.normal_(m.bias, 0, 0.2)
    B, Q, h, w = 1, 3, 5, 7
    x, bm = torch.randn(B, 256, h, w), torch.rand(B, Q, 8, h, w)
    fpns = [torch.randn(B, 1024, 2 * h, 2 * w) * 0.3, torch.randn(B, 512, 4 * h, 4 * w) * 0.3, torch.randn(B, 256, 8 * h, 8 * w) * 0.3]
    with torch.no_grad():
        ref = head(x, bm, fpns)
        with gpu_path_on_emulator() as lib:
            got = head(x, bm, fpns)
            calls = dict(lib.calls)
            lib.calls.clear()
            prev_tail = ds.set_mask_head_fused_tail(False)      # the pass-by-pass route of rounds 4-5
            try:
                passes = head(x, bm, fpns)
                calls_passes = dict(lib.calls)
            finally:
                ds.set_mask_head_fused_tail(prev_tail)
            prev = ds.set_mask_head_split(False)
            try:
                off = head(x, bm, fpns)
            finally:
                ds.set_mask_head_split(prev)
    # round 6 (the default): lay2 as a convolution, lay3 .. lay5 with the FPN merge and the previous GroupNorm + ReLU in their fetch
    # (statistics passes only), gn5 + ReLU + out_lay in one pass, the front's GroupNorm + ReLU through the library's own kernel
    # (+ lay1's image part, once per image)
    assert calls.get("tf_conv_packed_f32", 0) + calls.get("tf_conv3x3_split_f32", 0) == 2 and calls.get("tf_conv3x3_merge_packed_f32") == 3, calls
    assert calls.get("tf_groupnorm_stats_nhwc_f32") == 3 and calls.get("tf_groupnorm_relu_conv3x3_c1_nhwc_f32") == 1, calls
    assert calls.get("tf_groupnorm_relu_nhwc_f32") == 1 and calls.get("tf_upsample_add_nhwc_f32") == 1, calls
    n_conv = sum(calls_passes.get(k, 0) for k in ("tf_conv3x3_split_f32", "tf_conv3x3_splitk_f32", "tf_conv_packed_f32"))
    assert n_conv == 4 and calls_passes.get("tf_groupnorm_relu_nhwc_f32") == 4, calls_passes   # GroupNorm + ReLU in one pass each
    for out in (got, passes):
        assert out.shape == ref.shape and float((out - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(off, ref)     # switched off: the library path


def test_training_step_through_the_emulated_kernels():
    """One training step (padded two-image batch, previous-frame pass, track-query augmentation, SetCriterion, backward)
    with MSDeformAttnFunction running the HIP kernels under the emulator -- forward msda_fwd_f32_pquad / _direct, backward
    msda_bwd_f32_sorted2 / _buf -- against the reference's losses and gradient norms."""
    with gpu_path_on_emulator() as lib:
        loss_dict, total, grads = shared.run_train_step()
        calls = dict(lib.calls)
    shared.compare_train_to_golden(loss_dict, total, grads, rtol=2e-4)
    assert calls.get("tf_msda_backward_f32", 0) >= 5 and calls.get("tf_msda_forward_f32", 0) >= 5


@pytest.mark.skipif(shared.os.environ.get("TF_EMU_FULL") != "1",
                    reason="two minutes on 8 cores: TF_EMU_FULL=1, or tools/emu_full_size.py (profiles/r02_emulator_full_size_parity.txt)")
def test_full_size_cfg2_model_on_the_emulator_with_every_opt_in_route():
    """BASELINE cfg 2 (800 x 1333, 300 object + 100 track queries) through the emulated GPU path with every opt-in route
    on, against the full-size golden of the reference's own classes -- tests/test_full_size_gpu.py's comparison without
    a GPU (about two minutes on 8 cores; cfg 4 takes seven: tools/emu_full_size.py, profiles/r02_emulator_full_size_parity.txt)."""
    from tests import test_full_size_gpu as full, util_models as um
    from trackformer_amd import config, factory

    def fn():
        model, post, args = um.build("cfg2_full", factory.build_model, config.make_args)
        model.tracking()
        img, prev, target = um.model_inputs("cfg2_full", args.hidden_dim)
        with torch.no_grad():
            out, _, feats, memory, hs = model(img, target, None)
            res = post['bbox'](out, torch.tensor([list(um.FULL_ORIG)]))[0]
        return model, out, res, feats, memory
    model, out, res, feats, memory, calls = _run("cfg2_full", True, fn=fn)
    dbox, dlogit = full._compare("cfg2_full", model, out, res, feats, memory)
    print("cfg2_full, every opt-in route, emulator: max |d boxes| %.2e, max |d logits| %.2e" % (dbox, dlogit))
    # the 12 packed FFN linears of the encoder are inside the 6 one-launch blocks (+ 6 of the decoder: _run lowers the row limit)
    assert calls.get("tf_conv_packed_f32", 0) >= 17 + 3 and calls.get("tf_ffn_fused_f32") == 12
