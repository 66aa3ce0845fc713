"""Box utilities of the forward/tracking path.

Restates util/box_ops.py:9-61 of the reference and the torchvision.ops.boxes functions the
reference imports (box_area, box_iou, nms, clip_boxes_to_image; tracker.py:11, box_ops.py:6).
torchvision is not part of this build, so these are written from the published definitions:
area = (x2-x1)(y2-y1) without +1; greedy NMS suppresses a box whose IoU with an already kept,
higher scoring box is > threshold and returns kept indices by descending score.
"""
import torch
from torch import Tensor


def box_cxcywh_to_xyxy(x: Tensor) -> Tensor:
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x: Tensor) -> Tensor:
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def box_area(boxes: Tensor) -> Tensor:
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def _inter_union(boxes1: Tensor, boxes2: Tensor):
    area1 = box_area(boxes1)
    area2 = box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    return inter, union


def box_iou_union(boxes1: Tensor, boxes2: Tensor):
    """[N,4] x [M,4] -> (iou[N,M], union[N,M])   (util/box_ops.py:24-37)."""
    inter, union = _inter_union(boxes1, boxes2)
    return inter / union, union


def box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """torchvision.ops.boxes.box_iou semantics: the IoU matrix only."""
    inter, union = _inter_union(boxes1, boxes2)
    return inter / union


def generalized_box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """GIoU matrix [N,M] for xyxy boxes (util/box_ops.py:40-61)."""
    assert (boxes1[:, 2:] >= boxes1[:, :2]).all()
    assert (boxes2[:, 2:] >= boxes2[:, :2]).all()
    iou, union = box_iou_union(boxes1, boxes2)
    lt = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    hull = wh[:, :, 0] * wh[:, :, 1]
    return iou - (hull - union) / hull


def generalized_box_iou_pairs(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """GIoU of boxes1[i] with boxes2[i] ([N, 4] xyxy each -> [N]): the diagonal of generalized_box_iou, operation by operation, without
    the N x N matrix the reference builds to take its diagonal (models/detr.py:320-323)."""
    area1, area2 = box_area(boxes1), box_area(boxes2)
    wh = (torch.min(boxes1[:, 2:], boxes2[:, 2:]) - torch.max(boxes1[:, :2], boxes2[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area1 + area2 - inter
    iou = inter / union
    wh = (torch.max(boxes1[:, 2:], boxes2[:, 2:]) - torch.min(boxes1[:, :2], boxes2[:, :2])).clamp(min=0)
    hull = wh[:, 0] * wh[:, 1]
    return iou - (hull - union) / hull


def clip_boxes_to_image(boxes: Tensor, size) -> Tensor:
    """Clamp xyxy boxes into an image of size (h, w)."""
    h, w = size
    x = boxes[..., 0::2].clamp(min=0, max=w)
    y = boxes[..., 1::2].clamp(min=0, max=h)
    return torch.stack((x[..., 0], y[..., 0], x[..., 1], y[..., 1]), dim=-1)


def nms_keep_mask(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """Greedy NMS as a boolean keep mask [K] in the ORIGINAL box order.

    Boxes are visited in stable descending-score order (ties, e.g. several +inf scores as produced by
    tracker.py:493, keep their input order).  On CPU this is the textbook sweep; on a GPU tensor the
    sweep runs as a fixed-point iteration on the K x K "i suppresses j" matrix
    (keep_j = not any_{i<j, keep_i} IoU_ij > thr), which reaches the greedy solution in at most K
    steps and usually 2-3, without a per-box host round trip.
    """
    k = boxes.shape[0]
    if k == 0:
        return torch.zeros(0, dtype=torch.bool, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order]
    sup = torch.triu(box_iou(b, b) > iou_threshold, diagonal=1)  # sup[i, j]: i (better) overlaps j
    if boxes.device.type == "cpu":
        sup_np = sup.numpy()
        dead = sup_np[0] & False
        keep_np = dead.copy()
        for i in range(k):
            if not dead[i]:
                keep_np[i] = True
                dead |= sup_np[i]
        keep = torch.from_numpy(keep_np)
    else:
        keep = torch.ones(k, dtype=torch.bool, device=boxes.device)
        for _ in range(k):
            new_keep = ~((sup & keep[:, None]).any(dim=0))
            if torch.equal(new_keep, keep):
                break
            keep = new_keep
    mask = torch.zeros(k, dtype=torch.bool, device=boxes.device)
    mask[order] = keep
    return mask


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """torchvision.ops.nms semantics: indices of kept boxes, sorted by decreasing score."""
    if boxes.shape[0] == 0:
        return torch.zeros(0, dtype=torch.int64, device=boxes.device)
    if boxes.device.type == "cpu" and boxes.dtype == torch.float32 and scores.dtype == torch.float32:
        # host boxes (the tracker's association leg): the library's C sweep -- same arithmetic, same visiting order
        import ctypes
        from . import _cabi
        b, s = boxes.contiguous(), scores.contiguous()
        keep = torch.empty(b.shape[0], dtype=torch.int64)
        n_keep = ctypes.c_int(0)
        rc = _cabi.lib().tf_nms_host_f32(b.data_ptr(), s.data_ptr(), b.shape[0], float(iou_threshold), keep.data_ptr(),
                                         ctypes.addressof(n_keep))
        _cabi.check(rc, "tf_nms_host_f32")
        return keep[:n_keep.value]
    order = torch.sort(scores, descending=True, stable=True)[1]
    keep = nms_keep_mask(boxes, scores, iou_threshold)
    return order[keep[order]]
