"""build_model(args) -> (model, criterion, postprocessors); mirrors models/__init__.py:16-130."""
import torch

from .backbone import build_backbone
from .deformable_detr import DeformableDETR, DeformablePostProcess
from .deformable_transformer import build_deforamble_transformer
from .detr import DETR, PostProcess
from .detr_tracking import DeformableDETRTracking, DETRTracking

_NUM_CLASSES = {
    'coco': 91,
    'coco_panoptic': 250,
    'coco_person': 20, 'mot': 20, 'mot_crowdhuman': 20, 'crowdhuman': 20, 'mot_coco_person': 20,
}


def build_model(args):
    if args.dataset not in _NUM_CLASSES:
        raise NotImplementedError
    num_classes = _NUM_CLASSES[args.dataset]

    from .criterion import SetCriterion
    from .matcher import build_matcher

    device = torch.device(args.device)
    backbone = build_backbone(args)
    matcher = build_matcher(args)

    detr_kwargs = {
        'backbone': backbone,
        'num_classes': num_classes - 1 if args.focal_loss else num_classes,
        'num_queries': args.num_queries,
        'aux_loss': args.aux_loss,
        'overflow_boxes': args.overflow_boxes}
    tracking_kwargs = {
        'track_query_false_positive_prob': args.track_query_false_positive_prob,
        'track_query_false_negative_prob': args.track_query_false_negative_prob,
        'matcher': matcher,
        'backprop_prev_frame': args.track_backprop_prev_frame}
    mask_kwargs = {'freeze_detr': args.freeze_detr}

    if args.masks:
        from . import detr_segmentation as seg

    if args.deformable:
        detr_kwargs.update(
            transformer=build_deforamble_transformer(args),
            num_feature_levels=args.num_feature_levels,
            with_box_refine=args.with_box_refine,
            two_stage=args.two_stage,
            multi_frame_attention=args.multi_frame_attention,
            multi_frame_encoding=args.multi_frame_encoding,
            merge_frame_features=args.merge_frame_features)
        if args.tracking:
            model = seg.DeformableDETRSegmTracking(mask_kwargs, tracking_kwargs, detr_kwargs) \
                if args.masks else DeformableDETRTracking(tracking_kwargs, detr_kwargs)
        else:
            model = seg.DeformableDETRSegm(mask_kwargs, detr_kwargs) if args.masks \
                else DeformableDETR(**detr_kwargs)
    else:
        from .transformer import build_transformer
        detr_kwargs['transformer'] = build_transformer(args)
        if args.tracking:
            model = seg.DETRSegmTracking(mask_kwargs, tracking_kwargs, detr_kwargs) \
                if args.masks else DETRTracking(tracking_kwargs, detr_kwargs)
        else:
            model = seg.DETRSegm(mask_kwargs, detr_kwargs) if args.masks else DETR(**detr_kwargs)

    weight_dict = {'loss_ce': args.cls_loss_coef, 'loss_bbox': args.bbox_loss_coef,
                   'loss_giou': args.giou_loss_coef}
    if args.masks:
        weight_dict["loss_mask"] = args.mask_loss_coef
        weight_dict["loss_dice"] = args.dice_loss_coef
    if args.aux_loss:
        aux_weight_dict = {}
        for i in range(args.dec_layers - 1):
            aux_weight_dict.update({k + f'_{i}': v for k, v in weight_dict.items()})
        if args.two_stage:
            aux_weight_dict.update({k + '_enc': v for k, v in weight_dict.items()})
        weight_dict.update(aux_weight_dict)

    losses = ['labels', 'boxes', 'cardinality'] + (['masks'] if args.masks else [])
    criterion = SetCriterion(
        num_classes, matcher=matcher, weight_dict=weight_dict, eos_coef=args.eos_coef,
        losses=losses, focal_loss=args.focal_loss, focal_alpha=args.focal_alpha,
        focal_gamma=args.focal_gamma, tracking=args.tracking,
        track_query_false_positive_eos_weight=args.track_query_false_positive_eos_weight)
    criterion.to(device)

    postprocessors = {'bbox': DeformablePostProcess() if args.focal_loss else PostProcess()}
    if args.masks:
        postprocessors['segm'] = seg.PostProcessSegm()
        if args.dataset == "coco_panoptic":
            is_thing_map = {i: i <= 90 for i in range(201)}
            postprocessors["panoptic"] = seg.PostProcessPanoptic(is_thing_map, threshold=0.85)
    return model, criterion, postprocessors
