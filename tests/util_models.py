"""Shared builders for the model-level parity tests (used by the golden generator and the tests)."""
import torch

from tests.util_weights import perturb_state_dict

# name -> (config overlays, overrides, image size (H, W), #track queries, #frames)
MODEL_CASES = {
    "cfg2_deformable_tracking": (("deformable", "tracking", "mot17"), {}, (192, 256), 7),
    "cfg4_multi_frame_tracking": (("deformable", "tracking", "multi_frame", "mot17"),
                                  dict(num_queries=60), (160, 224), 5),
    "cfg1_plain_detr": ((), dict(dataset="coco"), (160, 192), 0),
}

TRACKER_FRAMES = 6
TRACKER_IMG = (192, 256)
TRACKER_ORIG = (480, 640)


def build(case, build_model_fn, make_args_fn, device="cpu", seed=42, weight_seed=1):
    overlays, overrides, _, _ = MODEL_CASES[case]
    args = make_args_fn(*overlays, device=str(device), **overrides)
    torch.manual_seed(seed)
    model, criterion, post = build_model_fn(args)
    perturb_state_dict(model, weight_seed)
    return model, post, args


def model_inputs(case, hidden_dim, seed=5):
    _, _, (h, w), n_track = MODEL_CASES[case]
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(1, 3, h, w, generator=g)
    prev = img + 0.05 * torch.randn(1, 3, h, w, generator=g)
    target = None
    if n_track:
        target = [{'track_query_hs_embeds': torch.randn(n_track, hidden_dim, generator=g),
                   'track_query_boxes': torch.rand(n_track, 4, generator=g) * 0.5 + 0.2,
                   'image_id': torch.tensor([1])}]
    return img, prev, target


def tracker_sequence(seed=11):
    """Synthetic sequence in the blob format of datasets/tracking/mot17_sequence.py:65-83."""
    g = torch.Generator().manual_seed(seed)
    h, w = TRACKER_IMG
    base = torch.randn(1, 3, h, w, generator=g)
    frames = []
    for _ in range(TRACKER_FRAMES):
        base = base + 0.15 * torch.randn(1, 3, h, w, generator=g)
        frames.append({'img': base.clone(), 'orig_size': torch.tensor([list(TRACKER_ORIG)]),
                       'size': torch.tensor([[h, w]]), 'dets': torch.zeros(1, 0, 4)})
    return frames


def to_device(target, device):
    if target is None:
        return None
    return [{k: v.to(device) for k, v in t.items()} for t in target]
