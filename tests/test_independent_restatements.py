"""CPU: the pieces of the path whose source is NOT in /root/reference -- torchvision 0.6's `nms`, `box_iou`, `generalized_box_iou`,
`clip_boxes_to_image` and the ResNet-50 trunk with FrozenBatchNorm2d (tracker.py:11, :326, :399, :495; backbone.py:45-55, :98-100;
util/box_ops.py) -- checked against INDEPENDENT restatements written here from the published definitions, with nothing from
trackformer_amd in the expected values: a float64 / float32 double loop for the box functions on random crowded inputs (ties,
duplicates, zero-area boxes, +inf scores), and the Bottleneck / stem formulas spelled out with torch.nn.functional on the
module's own parameters.  (The reference goldens cannot pin these pieces: the reference classes are run with the repo's
restatements injected under the torchvision names, oracle/reference_models.py:52-66.)"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from trackformer_amd import backbone, box_ops


def _iou32(a, b):
    """IoU of two xyxy boxes in float32 arithmetic, operation by operation as torchvision.ops.boxes.box_iou: inter / (area_a +
    area_b - inter), areas without + 1, the intersection clamped at zero."""
    f = np.float32
    area_a = f(f(a[2] - a[0]) * f(a[3] - a[1]))
    area_b = f(f(b[2] - b[0]) * f(b[3] - b[1]))
    w = max(f(0), f(min(a[2], b[2]) - max(a[0], b[0])))
    h = max(f(0), f(min(a[3], b[3]) - max(a[1], b[1])))
    inter = f(w * h)
    with np.errstate(invalid="ignore", divide="ignore"):
        return f(inter / f(f(area_a + area_b) - inter))


def _nms_by_the_book(boxes, scores, thr):
    """torchvision.ops.nms as published: visit the boxes in descending score order (stable), keep a box unless a kept box has
    IoU > thr with it (strictly); returns the kept indices in visiting order."""
    order = sorted(range(len(scores)), key=lambda i: (-scores[i] if not np.isnan(scores[i]) else np.inf, i))
    kept = []
    for i in order:
        if all(not (_iou32(boxes[j], boxes[i]) > thr) for j in kept):
            kept.append(i)
    return kept


def _random_boxes(rng, n, crowd):
    c = rng.random((n, 2)).astype(np.float32) * (20 if crowd else 200)
    wh = rng.random((n, 2)).astype(np.float32) * 30
    b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    if n >= 8:
        b[3] = b[1]                       # an exact duplicate
        b[5, 2:] = b[5, :2]               # a zero-area box
        b[6, 2] = b[6, 0]                 # zero width only
    return b


@pytest.mark.parametrize("n,thr,crowd", [(1, 0.5, True), (8, 0.5, True), (60, 0.3, True), (200, 0.9, True), (200, 0.5, False),
                                         (333, 0.1, True)])
def test_nms_equals_the_published_algorithm(n, thr, crowd):
    rng = np.random.default_rng(n)
    for trial in range(3):
        boxes = _random_boxes(rng, n, crowd)
        scores = rng.random(n).astype(np.float32)
        if n >= 8:
            scores[2] = scores[4]                     # a tie: input order decides
            scores[7] = np.inf                        # the tracker's "existing tracks first" trick (tracker.py:394-397)
            if trial == 2:
                scores[0] = np.inf
        want = _nms_by_the_book(boxes, scores, thr)
        got_host = box_ops.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).tolist()
        assert got_host == want
        mask = box_ops.nms_keep_mask(torch.from_numpy(boxes), torch.from_numpy(scores), thr)
        assert sorted(torch.nonzero(mask).flatten().tolist()) == sorted(want)


def test_box_iou_and_generalized_iou_equal_the_double_loop():
    rng = np.random.default_rng(5)
    a, b = _random_boxes(rng, 40, True), _random_boxes(rng, 31, True)
    iou = box_ops.box_iou(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    giou = box_ops.generalized_box_iou(torch.from_numpy(a[:, :]), torch.from_numpy(b)).numpy()
    for i in range(len(a)):
        for j in range(len(b)):
            want = _iou32(a[i], b[j])
            assert (np.isnan(want) and np.isnan(iou[i, j])) or abs(float(iou[i, j]) - float(want)) <= 1e-6, (i, j)
            # GIoU = IoU - (hull - union) / hull (Rezatofighi et al.; util/box_ops.py:41-60), in float64 from the definitions
            x = a[i].astype(np.float64)
            y = b[j].astype(np.float64)
            inter = max(0.0, min(x[2], y[2]) - max(x[0], y[0])) * max(0.0, min(x[3], y[3]) - max(x[1], y[1]))
            union = (x[2] - x[0]) * (x[3] - x[1]) + (y[2] - y[0]) * (y[3] - y[1]) - inter
            hull = (max(x[2], y[2]) - min(x[0], y[0])) * (max(x[3], y[3]) - min(x[1], y[1]))
            if union > 0 and hull > 0:
                assert abs(float(giou[i, j]) - (inter / union - (hull - union) / hull)) <= 2e-5, (i, j)


def test_clip_boxes_to_image_equals_the_definition():
    rng = np.random.default_rng(9)
    b = (rng.random((50, 4)).astype(np.float32) * 900 - 150)
    h, w = 480, 640
    got = box_ops.clip_boxes_to_image(torch.from_numpy(b), (h, w)).numpy()
    want = b.copy()
    want[:, 0::2] = np.clip(want[:, 0::2], 0, w)      # torchvision: x to [0, width], y to [0, height]
    want[:, 1::2] = np.clip(want[:, 1::2], 0, h)
    assert np.array_equal(got, want)


def _frozen_bn(x, bn):
    """FrozenBatchNorm2d as backbone.py:45-55 defines it: x * (weight * rsqrt(running_var + 1e-5)) + (bias - running_mean * scale)."""
    scale = bn.weight * (bn.running_var + 1e-5).rsqrt()
    shift = bn.bias - bn.running_mean * scale
    return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


def _randomise(net, seed):
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, backbone.FrozenBatchNorm2d):
            m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


@pytest.mark.parametrize("stride,down", [(1, False), (1, True), (2, True)])
def test_bottleneck_equals_the_published_formula(stride, down):
    """torchvision Bottleneck (v1.5): out = relu(bn1(conv1(x))); out = relu(bn2(conv2(out)))  [3 x 3, the block's stride];
    out = bn3(conv3(out)); out = relu(out + (downsample(x) if downsample else x)) -- spelled out with torch.nn.functional on the
    block's own parameters, against the module's forward (which folds the norms into the convolutions in inference)."""
    torch.manual_seed(0)
    cin, planes = (256, 64) if not down else (128, 64)
    ds = None
    if down:
        ds = torch.nn.Sequential(torch.nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False), backbone.FrozenBatchNorm2d(planes * 4))
    blk = backbone.Bottleneck(cin, planes, stride=stride, downsample=ds).eval()
    with torch.no_grad():
        _randomise(blk, 1)
        x = torch.randn(2, cin, 14, 18)
        out = F.relu(_frozen_bn(F.conv2d(x, blk.conv1.weight), blk.bn1))
        out = F.relu(_frozen_bn(F.conv2d(out, blk.conv2.weight, stride=stride, padding=1), blk.bn2))
        out = _frozen_bn(F.conv2d(out, blk.conv3.weight), blk.bn3)
        idt = x if ds is None else _frozen_bn(F.conv2d(x, ds[0].weight, stride=stride), ds[1])
        want = F.relu(out + idt)
        got = blk(x)
    assert got.shape == want.shape and float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_resnet50_trunk_equals_the_published_composition():
    """conv1 (7 x 7 / 2, padding 3) -> FrozenBN -> ReLU -> MaxPool(3, 2, 1) -> layer1..4, layer by layer with torch.nn.functional
    for the stem and the blocks' own forward for the layers, against IntermediateLayerGetter's outputs '0'..'3'."""
    torch.manual_seed(0)
    net = backbone.resnet("resnet50", [False, False, False], backbone.FrozenBatchNorm2d).eval()
    with torch.no_grad():
        _randomise(net, 2)
        x = torch.randn(1, 3, 64, 96)
        y = F.max_pool2d(F.relu(_frozen_bn(F.conv2d(x, net.conv1.weight, stride=2, padding=3), net.bn1)), 3, 2, 1)
        want = {}
        for i, name in enumerate(("layer1", "layer2", "layer3", "layer4")):
            y = getattr(net, name)(y)
            want[str(i)] = y
        getter = backbone.IntermediateLayerGetter(net, {"layer1": "0", "layer2": "1", "layer3": "2", "layer4": "3"})
        got = getter(x)
    for k in want:
        assert float((got[k] - want[k]).abs().max()) <= 2e-5 * float(want[k].abs().max()), k
