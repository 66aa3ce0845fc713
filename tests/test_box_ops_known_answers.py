"""CPU: hand-derived known-answer cases for the torchvision functions this build restates (torchvision is not
installed and not vendored by the reference, so no reference artefact pins them):
`torchvision.ops.nms`, `box_iou`, `box_area`, `clip_boxes_to_image` (tracker.py:11, :326, :399, :495) and the
ResNet-50 / IntermediateLayerGetter key layout (backbone.py:70-100).  Every expected value below is worked out
by hand from the published definitions: area = (x2 - x1)(y2 - y1) (no +1), IoU = inter / (a1 + a2 - inter),
greedy NMS in descending score order suppressing IoU > threshold (strictly), kept indices by descending score."""
import math

import torch

from trackformer_amd import box_ops


def test_area_and_iou_by_hand():
    a = torch.tensor([[0., 0., 10., 10.], [5., 5., 15., 15.], [20., 20., 30., 25.], [2., 2., 4., 4.]])
    assert box_ops.box_area(a).tolist() == [100., 100., 50., 4.]
    iou = box_ops.box_iou(a, a)
    # boxes 0/1 overlap in [5,10]^2: inter 25, union 175
    assert math.isclose(float(iou[0, 1]), 25. / 175., rel_tol=1e-6)
    assert float(iou[0, 2]) == 0.0                      # disjoint
    assert math.isclose(float(iou[0, 3]), 4. / 100., rel_tol=1e-6)    # contained: inter = the small box
    assert torch.allclose(iou, iou.t()) and torch.allclose(iou.diagonal(), torch.ones(4))
    # boxes that only touch along an edge have zero intersection
    touch = torch.tensor([[0., 0., 10., 10.], [10., 0., 20., 10.]])
    assert float(box_ops.box_iou(touch, touch)[0, 1]) == 0.0


def test_nms_threshold_is_strict_and_order_is_by_score():
    # IoU(0,1) = 1/3 exactly representable? inter 50, union 150 -> 0.3333..; use boxes with IoU exactly 0.5:
    # [0,0,10,10] and [0,0,10,5]: inter 50, union 100
    boxes = torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 5.], [100., 100., 110., 110.]])
    scores = torch.tensor([0.6, 0.9, 0.3])
    # threshold 0.5: IoU == 0.5 is NOT > 0.5 -> nothing suppressed; order by descending score
    assert box_ops.nms(boxes, scores, 0.5).tolist() == [1, 0, 2]
    # threshold just below: the lower scoring box of the pair (index 0) goes
    assert box_ops.nms(boxes, scores, 0.49).tolist() == [1, 2]
    assert box_ops.nms_keep_mask(boxes, scores, 0.49).tolist() == [False, True, True]


def test_nms_is_greedy_not_transitive():
    # chain A-B-C: A overlaps B, B overlaps C, A does not overlap C.  Greedy: A kept, B suppressed by A,
    # C kept (its only suppressor B is already dead).
    boxes = torch.tensor([[0., 0., 10., 10.], [4., 0., 14., 10.], [8., 0., 18., 10.]])
    scores = torch.tensor([0.9, 0.8, 0.7])
    # IoU(A,B) = 60/140 = 0.428..., IoU(B,C) = 0.428..., IoU(A,C) = 20/180 = 0.111
    assert box_ops.nms(boxes, scores, 0.4).tolist() == [0, 2]
    # with B the best box, both neighbours go
    assert box_ops.nms(boxes, torch.tensor([0.8, 0.9, 0.7]), 0.4).tolist() == [1]


def test_nms_with_infinite_scores_keeps_input_order_among_ties():
    """tracker.py:493-495: existing tracks get score +inf before the joint NMS with the new detections; among
    equal (+inf) scores the earlier box wins (stable order), and every +inf box outranks every finite one."""
    inf = float("inf")
    boxes = torch.tensor([[0., 0., 10., 10.],      # track 0
                          [1., 0., 11., 10.],      # track 1: IoU with track 0 = 90/110 = 0.818
                          [0., 0., 10., 10.],      # detection duplicating track 0
                          [50., 50., 60., 60.]])   # free-standing detection
    scores = torch.tensor([inf, inf, 0.99, 0.5])
    assert box_ops.nms(boxes, scores, 0.9).tolist() == [0, 1, 3]   # detection 2 (IoU 1.0 with track 0) goes
    assert box_ops.nms(boxes, scores, 0.8).tolist() == [0, 3]      # ... and track 1 loses the tie to track 0
    # swapping the two tracks swaps the winner: ties are resolved by input order, not by box content
    swapped = boxes[[1, 0, 2, 3]]
    assert box_ops.nms(swapped, scores, 0.8).tolist() == [0, 3]
    assert box_ops.nms(boxes[:0], scores[:0], 0.5).tolist() == []


def test_nms_gpu_formulation_equals_the_sweep():
    """The fixed-point form used on device tensors (box_ops.nms_keep_mask) against the sequential sweep on random
    crowded boxes; run on CPU through the same code path by calling the matrix form directly."""
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        k = 60
        xy = torch.rand(k, 2, generator=g) * 50
        wh = torch.rand(k, 2, generator=g) * 30 + 1
        boxes = torch.cat([xy, xy + wh], 1)
        scores = torch.rand(k, generator=g)
        scores[torch.randperm(k, generator=g)[:5]] = float("inf")
        thr = 0.3
        ref = box_ops.nms_keep_mask(boxes, scores, thr)
        order = torch.sort(scores, descending=True, stable=True)[1]
        b = boxes[order]
        sup = torch.triu(box_ops.box_iou(b, b) > thr, diagonal=1)
        keep = torch.ones(k, dtype=torch.bool)
        for _ in range(k):
            new_keep = ~((sup & keep[:, None]).any(dim=0))
            if torch.equal(new_keep, keep):
                break
            keep = new_keep
        mask = torch.zeros(k, dtype=torch.bool)
        mask[order] = keep
        assert torch.equal(mask, ref)


def test_clip_boxes_to_image():
    boxes = torch.tensor([[-5., -2., 700., 500.], [10., 20., 30., 40.]])
    out = box_ops.clip_boxes_to_image(boxes, (480, 640))     # size = (h, w): x to [0, 640], y to [0, 480]
    assert out.tolist() == [[0., 0., 640., 480.], [10., 20., 30., 40.]]


def test_cxcywh_xyxy_round_trip():
    b = torch.tensor([[0.5, 0.4, 0.2, 0.1]])
    xyxy = box_ops.box_cxcywh_to_xyxy(b)
    assert torch.allclose(xyxy, torch.tensor([[0.4, 0.35, 0.6, 0.45]]))
    assert torch.allclose(box_ops.box_xyxy_to_cxcywh(xyxy), b)


def test_resnet50_layout_matches_the_published_definition():
    """torchvision resnet50 (v1.5) as backbone.py:98-100 uses it: [3,4,6,3] bottlenecks, stride 2 on the 3x3
    conv2 of the first block of layer2-4, downsample = (conv1x1, bn) on the first block of every layer,
    25 557 032 parameters with the classifier (23 508 032 without fc), state-dict names of SURVEY Appendix B;
    IntermediateLayerGetter returns layer1..4 under '0'..'3' with strides 4/8/16/32."""
    from trackformer_amd import backbone
    net = backbone.resnet("resnet50", [False, False, False], backbone.FrozenBatchNorm2d)
    sd = net.state_dict()
    for name, shape in (("conv1.weight", (64, 3, 7, 7)), ("layer1.0.conv1.weight", (64, 64, 1, 1)),
                        ("layer1.0.downsample.0.weight", (256, 64, 1, 1)), ("layer2.0.conv2.weight", (128, 128, 3, 3)),
                        ("layer3.5.conv3.weight", (1024, 256, 1, 1)), ("layer4.2.bn3.running_var", (2048,)),
                        ("layer4.0.downsample.1.weight", (2048,))):
        assert tuple(sd[name].shape) == shape, name
    assert [len(getattr(net, "layer%d" % i)) for i in (1, 2, 3, 4)] == [3, 4, 6, 3]
    assert net.layer2[0].conv2.stride == (2, 2) and net.layer2[0].conv1.stride == (1, 1)   # v1.5: stride on the 3x3
    assert net.layer1[0].conv2.stride == (1, 1) and net.layer2[1].conv2.stride == (1, 1)
    conv_params = sum(v.numel() for k, v in sd.items() if "conv" in k or "downsample.0" in k)
    assert conv_params == 23454912          # torchvision resnet50: 25 557 032 - fc 2 049 000 - 53 120 bn affine
    getter = backbone.IntermediateLayerGetter(net, {"layer1": "0", "layer2": "1", "layer3": "2", "layer4": "3"})
    with torch.no_grad():
        out = getter(torch.zeros(1, 3, 64, 96))
    assert list(out) == ["0", "1", "2", "3"]
    assert [tuple(v.shape[1:]) for v in out.values()] == [(256, 16, 24), (512, 8, 12), (1024, 4, 6), (2048, 2, 3)]


def test_host_nms_in_the_library_equals_the_matrix_sweep():
    """box_ops.nms on host boxes runs tf_nms_host_f32 (csrc/msda_host.cpp); nms_keep_mask is the K x K IoU matrix + sweep it
    replaced in the tracker's association leg: same kept set, same order, incl. +inf scores, ties, degenerate boxes."""
    g = torch.Generator().manual_seed(0)
    for n, thr in [(1, 0.5), (7, 0.3), (250, 0.9), (250, 0.5), (400, 0.1)]:
        c = torch.rand(n, 2, generator=g) * 100
        wh = torch.rand(n, 2, generator=g) * 30
        boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
        boxes[::17, 2:] = boxes[::17, :2]                      # zero-area boxes: IoU 0 / 0 = NaN never suppresses
        boxes[5 % n] = boxes[0]                                # an exact duplicate
        scores = torch.rand(n, generator=g)
        scores[::11] = float("inf")                            # tracker.py:493: existing tracks win
        scores[3 % n] = scores[2 % n]                          # a tie keeps the input order
        order = torch.sort(scores, descending=True, stable=True)[1]
        expect = order[box_ops.nms_keep_mask(boxes, scores, thr)[order]]
        assert box_ops.nms(boxes, scores, thr).tolist() == expect.tolist()
