"""Shared builders for the model-level parity tests (used by the golden generator and the tests)."""
import torch

from tests.util_weights import perturb_state_dict

# name -> (config overlays, overrides, image size (H, W), #track queries, #frames)
MODEL_CASES = {
    "cfg2_deformable_tracking": (("deformable", "tracking", "mot17"), {}, (192, 256), 7),
    "cfg4_multi_frame_tracking": (("deformable", "tracking", "multi_frame", "mot17"),
                                  dict(num_queries=60), (160, 224), 5),
    "cfg1_plain_detr": ((), dict(dataset="coco"), (160, 192), 0),
    # detector variants the overlays do not exercise (deformable_detr.py:96-122, deformable_transformer.py:
    # 174-200,257-283): two-stage proposals, shared heads without box refinement, a single feature level
    "deformable_two_stage": (("deformable", "mot17"), dict(two_stage=True, num_queries=30), (128, 160), 0),
    "deformable_no_refine": (("deformable", "mot17"), dict(with_box_refine=False, num_queries=30),
                             (128, 160), 0),
    "deformable_one_level": (("deformable", "mot17"), dict(num_feature_levels=1, num_queries=30),
                             (128, 160), 0),
    # dense-attention tracking model (DETRTracking: transformer.py track-query prepend, detr.py focal off)
    "plain_detr_tracking": (("tracking", "mot17"), dict(num_queries=20), (128, 160), 5),
    # BASELINE cfg 5 model path: mask head on the tracking detector (MOTS20) ...
    "cfg5_segm_tracking": (("deformable", "tracking", "mots20"), dict(num_queries=24),
                           (128, 160), 4),
    # (the `multi_frame` overlay cannot be combined with masks in the reference either: hidden 288
    #  gives the head a GroupNorm(8, 36) -- detr_segmentation.py:125 -- which torch rejects)
    # ... and the plain-DETR flavour of the head (single feature level, dense attention)
    "cfg1_plain_detr_masks": (("coco_person_masks",), dict(num_queries=12), (96, 128), 0),
}
MASK_CASES = ("cfg5_segm_tracking", "cfg1_plain_detr_masks")
MASK_SIZES = {"cfg5_segm_tracking": ((128, 160), (200, 250)),      # (padded input size, original size)
              "cfg1_plain_detr_masks": ((96, 128), (150, 200))}

# BASELINE-size cases (800x1333; cfg 2: 300 object + 100 track queries, cfg 4: hidden 288, 500 + 300,
# 8 decoder levels).  Goldens: tests/golden/full_*.npz from tests/golden/make_golden_full.py; the
# tests run on the GPU only (the reference CPU path needs ~1 min per frame on 8 cores).
FULL_CASES = {
    # BASELINE cfg 1: plain DETR, one 480x640 frame, 100 object queries, coco classes (detr.py:62-128)
    "cfg1_full": ((), dict(dataset="coco"), (480, 640), 0),
    "cfg2_full": (("deformable", "tracking", "mot17"), {}, (800, 1333), 100),
    # BASELINE cfg 3: one training step at bs 2 (engine.py:119-158): not a forward case -- see FULL_TRAIN / train_batch(full=True)
    "cfg3_full": (("deformable", "tracking", "mot17"), dict(dropout=0.0), (800, 1333), 0),
    "cfg4_full": (("deformable", "tracking", "multi_frame", "mot17"), {}, (800, 1333), 300),
    # BASELINE cfg 5: mask head on the tracking detector (the buildable MOTS20 model, hidden 256: DESIGN.md section 2)
    "cfg5_full": (("deformable", "tracking", "mots20"), {}, (800, 1333), 100),
}
# the detector-forward cases every route / set-up test of tests/test_full_size_gpu.py is parametrised over
FULL_DETECTOR_CASES = ("cfg2_full", "cfg4_full")
FULL_TRAIN_SIZES = ((800, 1333), (768, 1280))   # cfg 3: two images of one batch (the second one padded: masks, valid ratios < 1)
FULL_TRAIN_BOXES = 30                           # 30 boxes per image
# cfg 5 fixture: pred_masks of every FULL_MASK_QUERY_STRIDE-th query, post-processed probabilities of the first 3
FULL_MASK_QUERY_STRIDE = 50
FULL_TRACKER_FRAMES = 3
FULL_IMG = (800, 1333)
FULL_ORIG = (1080, 1800)      # aspect of 800x1333 (datasets/transforms.py:115-146 resize)
FULL_IMG_CFG1 = (480, 640)    # cfg 1 is quoted on a 480x640 frame (no resize)
# rows / channels of the large tensors kept in the fixtures (full tensors would be tens of MB)
FULL_MEMORY_ROW_STRIDE = 89
FULL_FEAT_CH_STRIDE = 32

TRACKER_FRAMES = 6
TRACKER_IMG = (192, 256)
TRACKER_ORIG = (480, 640)


def build(case, build_model_fn, make_args_fn, device="cpu", seed=42, weight_seed=1):
    overlays, overrides, _, _ = (MODEL_CASES.get(case) or FULL_CASES[case])
    args = make_args_fn(*overlays, device=str(device), **overrides)
    torch.manual_seed(seed)
    model, criterion, post = build_model_fn(args)
    perturb_state_dict(model, weight_seed)
    return model, post, args


def model_inputs(case, hidden_dim, seed=5):
    _, _, (h, w), n_track = (MODEL_CASES.get(case) or FULL_CASES[case])
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(1, 3, h, w, generator=g)
    prev = img + 0.05 * torch.randn(1, 3, h, w, generator=g)
    target = None
    if n_track:
        target = [{'track_query_hs_embeds': torch.randn(n_track, hidden_dim, generator=g),
                   'track_query_boxes': torch.rand(n_track, 4, generator=g) * 0.5 + 0.2,
                   'image_id': torch.tensor([1])}]
    return img, prev, target


def tracker_sequence(seed=11, img=None, orig=None, n_frames=None):
    """Synthetic sequence in the blob format of datasets/tracking/mot17_sequence.py:65-83."""
    g = torch.Generator().manual_seed(seed)
    h, w = img or TRACKER_IMG
    base = torch.randn(1, 3, h, w, generator=g)
    frames = []
    for _ in range(n_frames or TRACKER_FRAMES):
        base = base + 0.15 * torch.randn(1, 3, h, w, generator=g)
        frames.append({'img': base.clone(), 'orig_size': torch.tensor([list(orig or TRACKER_ORIG)]),
                       'size': torch.tensor([[h, w]]), 'dets': torch.zeros(1, 0, 4)})
    return frames


def full_tracker_sequence(seed=31, n_frames=None):
    return tracker_sequence(seed, FULL_IMG, FULL_ORIG, n_frames or FULL_TRACKER_FRAMES)


# extra Tracker configurations pinned against the reference (tracker.py:124-165 public detections,
# :181-199 greedy re-identification, :356-360 termination counter, NMS thresholds)
TRACKER_VARIANTS = {
    "publicdet_iou": dict(cfg=dict(public_detections='min_iou_0_5'), reid=False, dets=True),
    "publicdet_center": dict(cfg=dict(public_detections='center_distance'), reid=False, dets=True),
    "nms_termination": dict(cfg=dict(detection_nms_thresh=0.5, track_nms_thresh=0.6, steps_termination=2,
                                     detection_obj_score_thresh=0.5, track_obj_score_thresh=0.45),
                            reid=False, dets=False),
    "reid_greedy": dict(cfg=dict(reid_greedy_matching=True, reid_sim_threshold=2.0,
                                 reid_score_thresh=0.45), reid=True, dets=False),
}


def public_detections(model, post, frames, every=3, seed=21):
    """Synthetic public detections: every `every`-th confident detection of the detector on each
    frame (no track queries), jittered by up to 2 px -- generated once with the reference model and
    stored in the fixture."""
    g = torch.Generator().manual_seed(seed)
    dets = []
    with torch.no_grad():
        for blob in frames:
            out, *_ = model(blob['img'], None, None)
            res = post['bbox'](out, blob['orig_size'])[0]
            boxes = res['boxes'][res['scores'] > 0.4][::every]
            dets.append((boxes + (torch.rand(boxes.shape, generator=g) * 4 - 2)).clamp(min=0)[None])
    return dets


def to_device(target, device):
    if target is None:
        return None
    return [{k: v.to(device) for k, v in t.items()} for t in target]


# ---------------------------------------------------------------------------------- training (cfg 3)
TRAIN_OVERRIDES = dict(dropout=0.0, num_queries=40, enc_layers=2, dec_layers=3)


def build_train(build_model_fn, make_args_fn, device="cpu", seed=42, weight_seed=2, masks=False, full=False):
    """full=True: the BASELINE cfg-3 model (300 queries, 6 + 6 layers); dropout stays 0 so that the step is deterministic."""
    args = make_args_fn("deformable", "tracking", "mots20" if masks else "mot17", device=str(device),
                        **(dict(dropout=0.0) if full else TRAIN_OVERRIDES))
    torch.manual_seed(seed)
    model, criterion, post = build_model_fn(args)
    perturb_state_dict(model, weight_seed)
    return model, criterion, args


def train_batch(seed=9, device="cpu", masks=False, full=False):
    """Two differently sized images (padding masks, valid ratios < 1) with previous-frame targets;
    masks=True adds box-shaped instance masks (the `masks` key loss_masks reads, detr.py:330-358).
    full=True: BASELINE cfg 3 -- 800x1333 + 768x1280, 30 boxes per image."""
    g = torch.Generator().manual_seed(seed)
    samples, targets = [], []
    for i, (h, w) in enumerate(FULL_TRAIN_SIZES if full else [(128, 160), (112, 144)]):
        img = torch.randn(3, h, w, generator=g)
        n = FULL_TRAIN_BOXES if full else 5 + i
        cxcy = torch.rand(n, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(n, 2, generator=g) * 0.2 + 0.05
        boxes = torch.cat([cxcy, wh], 1)
        prev_boxes = (boxes + 0.01 * torch.randn(n, 4, generator=g)).clamp(0.02, 0.98)
        ids = torch.arange(n) + 10 * i
        prev_keep = torch.arange(n) != 1          # one object is new in the current frame
        t = {'boxes': boxes.to(device), 'labels': torch.zeros(n, dtype=torch.long, device=device),
             'track_ids': ids.to(device), 'image_id': torch.tensor([i], device=device),
             'prev_image': (img + 0.05 * torch.randn(3, h, w, generator=g)).to(device),
             'prev_target': {'boxes': prev_boxes[prev_keep].to(device),
                             'labels': torch.zeros(int(prev_keep.sum()), dtype=torch.long,
                                                   device=device),
                             'track_ids': ids[prev_keep].to(device),
                             'image_id': torch.tensor([i], device=device)}}
        if masks:
            def box_masks(bx):
                m = torch.zeros(len(bx), h, w, dtype=torch.bool)
                for k, (cx, cy, bw, bh) in enumerate(bx.tolist()):
                    m[k, max(0, int((cy - bh / 2) * h)):int((cy + bh / 2) * h) + 1,
                      max(0, int((cx - bw / 2) * w)):int((cx + bw / 2) * w) + 1] = True
                return m
            t['masks'] = box_masks(boxes).to(device)
            t['prev_target']['masks'] = box_masks(prev_boxes[prev_keep]).to(device)
        samples.append(img.to(device))
        targets.append(t)
    return samples, targets


def train_step(model, criterion, samples, targets, rng_seed=7):
    """One forward + loss + backward exactly as engine.py:126-148 does it (without the optimiser)."""
    model.train()
    criterion.train()
    model.zero_grad()
    torch.manual_seed(rng_seed)      # host RNG of the track-query augmentation
    # an earlier test of the process may have switched MIOpen's exhaustive search on (runtime.configure_inference): a training
    # step would then time every solver for the forward / backward-data / backward-weight pass of ~50 convolution shapes --
    # minutes of search for one step.  The immediate-mode choice computes the same convolutions.
    prev_benchmark = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False
    try:
        outputs, targets, *_ = model(samples, targets)
        loss_dict = criterion(outputs, targets)
        weight_dict = criterion.weight_dict
        losses = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)
        losses.backward()
    finally:
        torch.backends.cudnn.benchmark = prev_benchmark
    grads = {n: float(p.grad.detach().double().norm()) for n, p in model.named_parameters()
             if p.grad is not None}
    return {k: float(v) for k, v in loss_dict.items()}, float(losses), grads


TRAIN_MASK_GRAD_KEYS = [
    "bbox_attention.q_linear.weight", "bbox_attention.k_linear.weight", "mask_head.lay1.weight",
    "mask_head.gn3.weight", "mask_head.adapter2.weight", "mask_head.out_lay.weight",
    "transformer.encoder.layers.0.self_attn.sampling_offsets.weight", "input_proj.1.0.weight",
]

TRAIN_GRAD_KEYS = [
    "transformer.encoder.layers.0.self_attn.sampling_offsets.weight",
    "transformer.encoder.layers.0.self_attn.value_proj.weight",
    "transformer.encoder.layers.1.self_attn.attention_weights.bias",
    "transformer.decoder.layers.0.cross_attn.sampling_offsets.bias",
    "transformer.decoder.layers.2.cross_attn.output_proj.weight",
    "transformer.level_embed", "input_proj.0.0.weight", "input_proj.3.0.weight",
    "backbone.0.body.layer2.0.conv1.weight", "backbone.0.body.layer4.2.conv3.weight",
    "class_embed.2.weight", "transformer.decoder.bbox_embed.0.layers.2.weight", "query_embed.weight",
]
