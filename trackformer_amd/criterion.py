"""Set-prediction loss of DETR / TrackFormer (training only; BASELINE cfg 3).

Same surface as the reference's SetCriterion (models/detr.py:139-443) and its helper losses
(util/misc.py:448-463 accuracy, :522-571 dice / sigmoid focal loss).
"""
import copy
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from . import box_ops
from .nested import nested_tensor_from_tensor_list


def _world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


@torch.no_grad()
def accuracy(output, target, topk=(1,)):
    """precision@k in percent."""
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    pred = output.topk(max(topk), 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]


def dice_loss(inputs, targets, num_boxes):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * (inputs * targets).sum(1)
    denominator = inputs.sum(-1) + targets.sum(-1)
    return (1 - (numerator + 1) / (denominator + 1)).sum() / num_boxes


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2,
                       query_mask=None, reduction=True):
    prob = inputs.sigmoid()
    ce_loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if not reduction:
        return loss
    if query_mask is not None:
        loss = torch.stack([l[m].mean(0) for l, m in zip(loss, query_mask)])
        return loss.sum() / num_boxes
    return loss.mean(1).sum() / num_boxes


# DEFAULT since round 6 (TF_CRITERION_LAYERS_AT_ONCE=0 / set_layers_at_once(False): the reference's loop over the decoder layers)
_LAYERS_AT_ONCE = os.environ.get("TF_CRITERION_LAYERS_AT_ONCE", "1") != "0"


def set_layers_at_once(on):
    global _LAYERS_AT_ONCE
    prev, _LAYERS_AT_ONCE = _LAYERS_AT_ONCE, bool(on)
    return prev


def counts_match(indices, aux_indices):
    n = sum(len(src) for src, _ in indices)
    return all(sum(len(src) for src, _ in ind) == n for ind in aux_indices)


class SetCriterion(nn.Module):
    """1) Hungarian assignment of predictions to targets, 2) class / box (/ mask) losses on the pairs,
    repeated for every auxiliary decoder output."""

    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses, focal_loss,
                 focal_alpha, focal_gamma, tracking, track_query_false_positive_eos_weight):
        super().__init__()
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.eos_coef = eos_coef
        self.losses = losses
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer('empty_weight', empty_weight)
        self.focal_loss = focal_loss
        self.focal_alpha = focal_alpha
        self.focal_gamma = focal_gamma
        self.tracking = tracking
        self.track_query_false_positive_eos_weight = track_query_false_positive_eos_weight

    # ------------------------------------------------------------------ helpers
    def _get_src_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return batch_idx, torch.cat([src for (src, _) in indices])

    def _get_tgt_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(tgt, i) for i, (_, tgt) in enumerate(indices)])
        return batch_idx, torch.cat([tgt for (_, tgt) in indices])

    def _target_classes(self, src_logits, targets, indices):
        idx = self._get_src_permutation_idx(indices)
        matched = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
        classes = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64,
                             device=src_logits.device)
        classes[idx] = matched
        return idx, matched, classes

    # ------------------------------------------------------------------ losses
    def loss_labels(self, outputs, targets, indices, _, log=True):
        src_logits = outputs['pred_logits']
        idx, matched, target_classes = self._target_classes(src_logits, targets, indices)
        loss_ce = F.cross_entropy(src_logits.transpose(1, 2), target_classes,
                                  weight=self.empty_weight, reduction='none')
        if self.tracking and self.track_query_false_positive_eos_weight:
            for i, target in enumerate(targets):
                if 'track_query_boxes' in target:
                    fp = target['track_queries_fal_pos_mask']
                    loss_ce[i, fp] *= 1 / self.eos_coef       # undo the no-object down-weighting
                    target_classes = target_classes.clone()
                    target_classes[i, fp] = 0                  # ... also in the normaliser
        losses = {'loss_ce': loss_ce.sum() / self.empty_weight[target_classes].sum()}
        if log:
            losses['class_error'] = 100 - accuracy(src_logits[idx], matched)[0]
        return losses

    def loss_labels_focal(self, outputs, targets, indices, num_boxes, log=True):
        src_logits = outputs['pred_logits']
        idx, matched, target_classes = self._target_classes(src_logits, targets, indices)
        onehot = torch.zeros(src_logits.shape[0], src_logits.shape[1], src_logits.shape[2] + 1,
                             dtype=src_logits.dtype, device=src_logits.device)
        onehot.scatter_(2, target_classes.unsqueeze(-1), 1)
        loss_ce = sigmoid_focal_loss(src_logits, onehot[:, :, :-1], num_boxes,
                                     alpha=self.focal_alpha, gamma=self.focal_gamma)
        losses = {'loss_ce': loss_ce * src_logits.shape[1]}
        if log:
            losses['class_error'] = 100 - accuracy(src_logits[idx], matched)[0]
        return losses

    @torch.no_grad()
    def loss_cardinality(self, outputs, targets, indices, num_boxes):
        pred_logits = outputs['pred_logits']
        tgt_lengths = torch.as_tensor([len(v["labels"]) for v in targets],
                                      device=pred_logits.device)
        card_pred = (pred_logits.argmax(-1) != pred_logits.shape[-1] - 1).sum(1)
        return {'cardinality_error': F.l1_loss(card_pred.float(), tgt_lengths.float())}

    def loss_boxes(self, outputs, targets, indices, num_boxes):
        idx = self._get_src_permutation_idx(indices)
        src_boxes = outputs['pred_boxes'][idx]
        target_boxes = torch.cat([t['boxes'][i] for t, (_, i) in zip(targets, indices)], dim=0)
        loss_bbox = F.l1_loss(src_boxes, target_boxes, reduction='none')
        loss_giou = 1 - torch.diag(box_ops.generalized_box_iou(
            box_ops.box_cxcywh_to_xyxy(src_boxes), box_ops.box_cxcywh_to_xyxy(target_boxes)))
        return {'loss_bbox': loss_bbox.sum() / num_boxes, 'loss_giou': loss_giou.sum() / num_boxes}

    def loss_masks(self, outputs, targets, indices, num_boxes):
        src_idx = self._get_src_permutation_idx(indices)
        tgt_idx = self._get_tgt_permutation_idx(indices)
        src_masks = outputs["pred_masks"]
        target_masks, _ = nested_tensor_from_tensor_list([t["masks"] for t in targets]).decompose()
        target_masks = target_masks.to(src_masks)
        src_masks = F.interpolate(src_masks[src_idx][:, None], size=target_masks.shape[-2:],
                                  mode="bilinear", align_corners=False)[:, 0].flatten(1)
        target_masks = target_masks[tgt_idx].flatten(1)
        return {"loss_mask": sigmoid_focal_loss(src_masks, target_masks, num_boxes),
                "loss_dice": dice_loss(src_masks, target_masks, num_boxes)}

    def _layers_at_once(self, layer_outputs, targets, all_indices, num_boxes):
        """The class (focal), cardinality and box losses of the final + auxiliary decoder layers computed TOGETHER (round 6): the
        layers' predictions are stacked [L, B, Q, ...] and every loss is one chain of kernels with a leading layer dimension instead
        of L chains (the reference loops over the layers, models/detr.py:266-289: ~70 launches forward and ~100 backward per layer on
        [B, Q] tensors; 7.6 ms of a cfg-3 step sat in the criterion).  The arithmetic per element is that of loss_labels_focal /
        loss_cardinality / loss_boxes; the GIoU of the matched pairs is computed for the pairs (box_ops.generalized_box_iou_pairs: the
        diagonal the reference takes of an N x N matrix).  -> {key: 0-d tensor} with the reference's keys (`loss_ce`, `loss_ce_0`, ...),
        or None when the layers cannot be stacked (the caller loops)."""
        L = len(layer_outputs)
        logits = [o['pred_logits'] for o in layer_outputs]
        boxes = [o['pred_boxes'] for o in layer_outputs]
        if any(t.shape != logits[0].shape for t in logits) or any(t.shape != boxes[0].shape for t in boxes):
            return None
        counts = [sum(len(src) for src, _ in ind) for ind in all_indices]
        if len(set(counts)) != 1:
            return None
        logits, boxes = torch.stack(logits), torch.stack(boxes)          # [L, B, Q, C], [L, B, Q, 4]
        dev = logits.device
        B, Q, C = logits.shape[1:]
        # (layer, image, query) of every matched prediction and its target's (image, index), layer by layer in the order the
        # per-layer losses concatenate them
        lay = torch.cat([torch.full((counts[0],), l, dtype=torch.int64) for l in range(L)])
        bat = torch.cat([torch.full_like(src, i) for ind in all_indices for i, (src, _) in enumerate(ind)])
        qry = torch.cat([src for ind in all_indices for (src, _) in ind])
        tgt_of = [torch.cat([t_idx + off for (_, t_idx), off in zip(ind, self._offsets(targets))]) for ind in all_indices]
        tgt = torch.cat(tgt_of)
        lay, bat, qry, tgt = (t.to(dev, non_blocking=True) for t in (lay, bat, qry, tgt))
        all_labels = torch.cat([t["labels"] for t in targets])
        all_boxes = torch.cat([t["boxes"] for t in targets])
        losses = {}
        suffix = [''] + ['_%d' % i for i in range(L - 1)]
        if 'labels' in self.losses:
            classes = torch.full((L, B, Q), self.num_classes, dtype=torch.int64, device=dev)
            matched = all_labels[tgt]
            classes[lay, bat, qry] = matched
            onehot = torch.zeros(L, B, Q, C + 1, dtype=logits.dtype, device=dev)
            onehot.scatter_(3, classes.unsqueeze(-1), 1)
            per_elem = sigmoid_focal_loss(logits, onehot[..., :-1], num_boxes, alpha=self.focal_alpha, gamma=self.focal_gamma,
                                          reduction=False)
            loss_ce = per_elem.mean(2).sum((1, 2)) / num_boxes * Q                       # [L]
            for l in range(L):
                losses['loss_ce' + suffix[l]] = loss_ce[l]
            n0 = counts[0]
            losses['class_error'] = 100 - accuracy(logits[0][bat[:n0], qry[:n0]], matched[:n0])[0]
        if 'cardinality' in self.losses:
            with torch.no_grad():
                tgt_lengths = torch.as_tensor([len(v["labels"]) for v in targets], device=dev).float()
                card_pred = (logits.argmax(-1) != C - 1).sum(2).float()                  # [L, B]
                card = (card_pred - tgt_lengths[None]).abs().mean(1)
            for l in range(L):
                losses['cardinality_error' + suffix[l]] = card[l]
        if 'boxes' in self.losses:
            src_boxes = boxes[lay, bat, qry]                                             # [L * n, 4]
            target_boxes = all_boxes[tgt]
            loss_bbox = F.l1_loss(src_boxes, target_boxes, reduction='none').view(L, -1).sum(1) / num_boxes
            a, b = box_ops.box_cxcywh_to_xyxy(src_boxes), box_ops.box_cxcywh_to_xyxy(target_boxes)
            loss_giou = (1 - box_ops.generalized_box_iou_pairs(a, b)).view(L, -1).sum(1) / num_boxes
            for l in range(L):
                losses['loss_bbox' + suffix[l]] = loss_bbox[l]
                losses['loss_giou' + suffix[l]] = loss_giou[l]
        return losses

    @staticmethod
    def _offsets(targets):
        off, out = 0, []
        for t in targets:
            out.append(off)
            off += len(t["labels"])
        return out

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        loss_map = {'labels': self.loss_labels_focal if self.focal_loss else self.loss_labels,
                    'cardinality': self.loss_cardinality, 'boxes': self.loss_boxes,
                    'masks': self.loss_masks}
        assert loss in loss_map, f'do you really want to compute {loss} loss?'
        return loss_map[loss](outputs, targets, indices, num_boxes, **kwargs)

    def _extra_losses(self, outputs, targets, num_boxes, suffix, indices=None):
        if indices is None:
            indices = self.matcher(outputs, targets)
        out = {}
        for loss in self.losses:
            if loss == 'masks':  # too costly on intermediate outputs
                continue
            kwargs = {'log': False} if loss == 'labels' else {}
            l_dict = self.get_loss(loss, outputs, targets, indices, num_boxes, **kwargs)
            out.update({k + suffix: v for k, v in l_dict.items()})
        return out

    def forward(self, outputs, targets):
        outputs_without_aux = {k: v for k, v in outputs.items() if k != 'aux_outputs'}
        aux_list = list(outputs.get('aux_outputs', []))
        # the final layer and the auxiliary layers are matched against the same targets: one device pass + one host copy for all of
        # them (matcher.match_many; the reference calls its matcher once per layer, detr.py:266-289 -- same assignments)
        if aux_list and hasattr(self.matcher, "match_many"):
            all_indices = self.matcher.match_many([outputs_without_aux] + aux_list, targets)
            indices, aux_indices = all_indices[0], all_indices[1:]
        else:
            indices, aux_indices = self.matcher(outputs_without_aux, targets), [None] * len(aux_list)

        # number of target boxes averaged over all ranks (the one scalar all-reduce of the loss)
        num_boxes = torch.as_tensor([sum(len(t["labels"]) for t in targets)], dtype=torch.float,
                                    device=next(iter(outputs.values())).device)
        if _world_size() > 1:
            dist.all_reduce(num_boxes)
        num_boxes = torch.clamp(num_boxes / _world_size(), min=1).item()

        losses = None
        if (_LAYERS_AT_ONCE and self.focal_loss and aux_list and all(ix is not None for ix in aux_indices) and counts_match(indices, aux_indices)
                and all(loss in ('labels', 'cardinality', 'boxes', 'masks') for loss in self.losses)):
            losses = self._layers_at_once([outputs_without_aux] + aux_list, targets, [indices] + list(aux_indices), num_boxes)
            if losses is not None and 'masks' in self.losses:   # (the final layer only, as in the loop)
                losses.update(self.get_loss('masks', outputs, targets, indices, num_boxes))
        if losses is None:
            losses = {}
            for loss in self.losses:
                losses.update(self.get_loss(loss, outputs, targets, indices, num_boxes))
            for i, aux_outputs in enumerate(aux_list):
                losses.update(self._extra_losses(aux_outputs, targets, num_boxes, f'_{i}', aux_indices[i]))
        if 'enc_outputs' in outputs:
            bin_targets = copy.deepcopy(targets)
            for bt in bin_targets:
                bt['labels'] = torch.zeros_like(bt['labels'])
            losses.update(self._extra_losses(outputs['enc_outputs'], bin_targets, num_boxes, '_enc'))
        return losses
