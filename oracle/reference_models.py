"""oracle/reference_models.py -- TEST INFRASTRUCTURE ONLY; works ONLY in the build container.

Imports the reference's own Python model classes from /root/reference/src (read-only, unmodified) so
that golden fixtures can be generated from them on CPU (tests/golden/make_golden_models.py).  The GPU
box has no /root/reference: nothing under tests/ run with `-m gpu`, smoke() or bench.py imports this.

Stubs installed in sys.modules (recipe of SURVEY.md Appendix D):
  * MultiScaleDeformableAttention -- empty module; `MSDeformAttnFunction.apply` is routed to the
    reference's pure-PyTorch `ms_deform_attn_core_pytorch` (func.py:34-54), i.e. the reference CPU path.
  * torchvision(.models/.ops...) -- torchvision is not installed and not vendored by the reference.
    Its ResNet-50, IntermediateLayerGetter, nms, box_iou, box_area, clip_boxes_to_image are supplied
    from trackformer_amd (written from the published definitions).  Parity of exactly these pieces is
    therefore UNPINNED by any reference artefact; everything else in the goldens is reference code.
  * visdom -- imported at module top by util/misc.py:22, never used on this path.
"""
import os
import sys
import types

REF_SRC = "/root/reference/src"


def available():
    return os.path.isdir(os.path.join(REF_SRC, "trackformer"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = None


def load():
    """Returns a namespace of reference modules (models.*, util.*)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference sources not present (expected in the build container only)")
    import torch
    import torch.nn.functional as F

    from trackformer_amd import backbone as tf_backbone
    from trackformer_amd import box_ops as tf_box_ops

    _mod("MultiScaleDeformableAttention")
    tv = _mod("torchvision", __version__="0.25.0")
    tv.models = _mod("torchvision.models",
                     resnet50=lambda replace_stride_with_dilation=None, pretrained=False,
                     norm_layer=None: tf_backbone.resnet(
                         "resnet50", replace_stride_with_dilation, norm_layer),
                     resnet101=lambda replace_stride_with_dilation=None, pretrained=False,
                     norm_layer=None: tf_backbone.resnet(
                         "resnet101", replace_stride_with_dilation, norm_layer))
    _mod("torchvision.models._utils", IntermediateLayerGetter=tf_backbone.IntermediateLayerGetter)
    tv.ops = _mod("torchvision.ops")
    tv.ops.misc = _mod("torchvision.ops.misc", interpolate=F.interpolate)
    _mod("torchvision.ops.feature_pyramid_network", FeaturePyramidNetwork=object,
         LastLevelMaxPool=object)
    tv.ops.boxes = _mod("torchvision.ops.boxes", box_area=tf_box_ops.box_area, nms=tf_box_ops.nms,
                        box_iou=tf_box_ops.box_iou,
                        clip_boxes_to_image=tf_box_ops.clip_boxes_to_image)
    _mod("visdom", Visdom=type("Visdom", (), {}))

    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    import trackformer.models.ops.modules.ms_deform_attn as ref_msda_module
    from trackformer.models.ops.functions import ms_deform_attn_core_pytorch

    class _CpuFunction:
        @staticmethod
        def apply(value, shapes, loc, attn, im2col_step):
            return ms_deform_attn_core_pytorch(value, shapes, loc, attn)

    ref_msda_module.MSDeformAttnFunction = _CpuFunction

    import trackformer.models as ref_models
    import trackformer.models.backbone as ref_backbone
    import trackformer.models.deformable_detr as ref_deformable_detr
    import trackformer.models.deformable_transformer as ref_deformable_transformer
    import trackformer.models.detr as ref_detr
    import trackformer.models.detr_tracking as ref_detr_tracking
    import trackformer.models.matcher as ref_matcher
    import trackformer.models.position_encoding as ref_position_encoding
    import trackformer.models.tracker as ref_tracker
    import trackformer.models.transformer as ref_transformer
    import trackformer.util.box_ops as ref_box_ops
    import trackformer.util.misc as ref_misc

    _loaded = types.SimpleNamespace(
        models=ref_models, backbone=ref_backbone, deformable_detr=ref_deformable_detr,
        deformable_transformer=ref_deformable_transformer, detr=ref_detr,
        detr_tracking=ref_detr_tracking, matcher=ref_matcher,
        position_encoding=ref_position_encoding, tracker=ref_tracker,
        transformer=ref_transformer, box_ops=ref_box_ops, misc=ref_misc,
        msda_module=ref_msda_module, core_pytorch=ms_deform_attn_core_pytorch)
    return _loaded


def accept_prev_features():
    """The reference's plain `DETR.forward(samples, targets)` cannot be called by its own
    `DETRTrackingBase.forward` / `Tracker.step`, which pass `prev_features` as a third positional
    argument (detr_tracking.py:275, tracker.py:307 vs detr.py:62).  For the goldens of the
    dense-attention tracking model the harness widens the signature (the extra argument is ignored,
    exactly as the deformable detector ignores it without multi-frame attention); nothing else of
    the reference is touched."""
    ref = load()
    if getattr(ref.detr.DETR.forward, "_accepts_prev_features", False):
        return
    original = ref.detr.DETR.forward

    def forward(self, samples, targets=None, prev_features=None):
        return original(self, samples, targets)

    forward._accepts_prev_features = True
    ref.detr.DETR.forward = forward
