"""Model / tracker hyper-parameters in the reference's config vocabulary.

The reference drives everything from sacred + YAML overlays (cfgs/train.yaml plus the named configs
`deformable`, `tracking`, `multi_frame`, ...; src/train.py:23-35) converted to a nested Namespace
(util/misc.py:574-580).  The same keys and effective values are restated here as plain dicts so that
`build_model(args)` receives an identical `args` object without sacred (absent in this build).
A reference `config.yaml` stored next to a checkpoint can also be loaded with `args_from_yaml`.
"""
import copy
from argparse import Namespace

# cfgs/train.yaml (model-relevant subset + the loss coefficients build_model reads)
BASE = dict(
    lr=2e-4, lr_backbone_names=['backbone.0'], lr_backbone=2e-5,
    lr_linear_proj_names=['reference_points', 'sampling_offsets'], lr_linear_proj_mult=0.1,
    lr_track=1e-4, batch_size=2, weight_decay=1e-4, epochs=50, lr_drop=40, clip_max_norm=0.1,
    deformable=False, with_box_refine=False, two_stage=False,
    freeze_detr=False, load_mask_head_from_model=None,
    backbone='resnet50', dilation=False, position_embedding='sine', num_feature_levels=1,
    enc_layers=6, dec_layers=6, dim_feedforward=2048, hidden_dim=256, dropout=0.1, nheads=8,
    num_queries=100, pre_norm=False, dec_n_points=4, enc_n_points=4,
    tracking=False, tracking_eval=True, track_prev_frame_range=0, track_prev_frame_rnd_augs=0.01,
    track_prev_prev_frame=False, track_backprop_prev_frame=False,
    track_query_false_positive_prob=0.1, track_query_false_negative_prob=0.4,
    track_attention=False, multi_frame_attention=False, multi_frame_encoding=True,
    multi_frame_attention_separate_encoder=True, merge_frame_features=False, overflow_boxes=False,
    masks=False, aux_loss=True,
    set_cost_class=1.0, set_cost_bbox=5.0, set_cost_giou=2.0,
    mask_loss_coef=1.0, dice_loss_coef=1.0, cls_loss_coef=1.0, bbox_loss_coef=5.0,
    giou_loss_coef=2.0, eos_coef=0.1, focal_loss=False, focal_alpha=0.25, focal_gamma=2,
    track_query_false_positive_eos_weight=True,
    dataset='coco', device='cuda', seed=42,
)

# named overlays, applied in the order given on the reference command line
OVERLAYS = {
    # cfgs/train_deformable.yaml
    'deformable': dict(deformable=True, num_feature_levels=4, num_queries=300, dim_feedforward=1024,
                       focal_loss=True, focal_alpha=0.25, focal_gamma=2, cls_loss_coef=2.0,
                       set_cost_class=2.0, overflow_boxes=True, with_box_refine=True),
    # cfgs/train_tracking.yaml
    'tracking': dict(tracking=True, tracking_eval=True, track_prev_frame_range=5,
                     track_query_false_positive_eos_weight=True),
    # cfgs/train_multi_frame.yaml
    'multi_frame': dict(num_queries=500, hidden_dim=288, multi_frame_attention=True,
                        multi_frame_encoding=True, multi_frame_attention_separate_encoder=True),
    # dataset overlays only matter for num_classes here (models/__init__.py:16-27)
    'mot17': dict(dataset='mot'),
    'mot17_crowdhuman': dict(dataset='mot_crowdhuman'),
    'crowdhuman': dict(dataset='crowdhuman'),
    'coco_person_masks': dict(dataset='coco_person', masks=True),
    'mots20': dict(dataset='mot', masks=True),
}

# cfgs/track.yaml:26-47
TRACKER_CFG = dict(
    public_detections=False, detection_obj_score_thresh=0.4, track_obj_score_thresh=0.4,
    detection_nms_thresh=0.9, track_nms_thresh=0.9, steps_termination=1, prev_frame_dist=1,
    inactive_patience=-1, reid_sim_threshold=0.0, reid_sim_only=False, reid_score_thresh=0.4,
    reid_greedy_matching=False,
)
# cfgs/track_reid.yaml
TRACKER_CFG_REID = dict(TRACKER_CFG, inactive_patience=5)


def make_args(*overlays, **overrides) -> Namespace:
    """make_args('deformable', 'tracking', 'mot17', num_queries=300) -> Namespace for build_model."""
    cfg = copy.deepcopy(BASE)
    for name in overlays:
        if name not in OVERLAYS:
            raise KeyError("unknown config overlay %r (have %s)" % (name, sorted(OVERLAYS)))
        cfg.update(OVERLAYS[name])
    unknown = set(overrides) - set(cfg)
    if unknown:
        raise KeyError("unknown config keys %s" % sorted(unknown))
    cfg.update(overrides)
    return Namespace(**cfg)


def args_from_yaml(path, **overrides) -> Namespace:
    """Load a reference-style config.yaml (as saved next to a checkpoint, track.py:63-68)."""
    import yaml
    with open(path) as f:
        loaded = yaml.safe_load(f)
    cfg = copy.deepcopy(BASE)
    cfg.update({k: v for k, v in loaded.items() if not isinstance(v, dict)})
    cfg.update(overrides)
    return Namespace(**cfg)


def tracker_cfg(reid=False, **overrides) -> dict:
    cfg = dict(TRACKER_CFG_REID if reid else TRACKER_CFG)
    unknown = set(overrides) - set(cfg)
    if unknown:
        raise KeyError("unknown tracker config keys %s" % sorted(unknown))
    cfg.update(overrides)
    return cfg
