"""Tiny end-to-end check used by __graft_entry__.smoke(): two Tracker.step calls on cuda:0."""
import torch


def run(device):
    from . import config, factory
    from .tracker import Tracker
    args = config.make_args('deformable', 'tracking', 'mot17', device=str(device))
    torch.manual_seed(42)
    model, _, post = factory.build_model(args)
    with torch.no_grad():  # make a few queries fire so that the track-query path is exercised
        for head in model.class_embed:
            head.bias.fill_(-0.5)
            head.bias[0] = 1.5
    model.to(device).tracking()
    tracker = Tracker(model, post, config.tracker_cfg(), False)
    tracker.reset()
    g = torch.Generator().manual_seed(0)
    img = torch.randn(1, 3, 256, 320, generator=g)
    with torch.no_grad():
        for _ in range(2):
            img = img + 0.1 * torch.randn(1, 3, 256, 320, generator=g)
            tracker.step({'img': img, 'orig_size': torch.tensor([[512, 640]]),
                          'size': torch.tensor([[256, 320]]), 'dets': torch.zeros(1, 0, 4)})
    torch.cuda.synchronize()
    results = tracker.get_results()
    assert tracker.frame_index == 2
    print("smoke: Tracker.step x2 on %s: %d tracks after 2 frames (ids %s...)" % (
        torch.cuda.get_device_name(0), len(tracker.tracks), sorted(results)[:5]))
