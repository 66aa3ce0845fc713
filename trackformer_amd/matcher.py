"""Hungarian matching between predictions and ground truth with TrackFormer's track-query constraints.

Same module surface as the reference's models/matcher.py (HungarianMatcher :12-131, build_matcher
:134-141).  The cost matrix is built on the device exactly as there (focal / softmax class cost +
L1 + GIoU); the track-query constraints, which the reference writes with a Python double loop over
every query (:104-125), are applied as three vectorised index assignments on the host copy.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

from .box_ops import box_cxcywh_to_xyxy, generalized_box_iou


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1,
                 focal_loss: bool = False, focal_alpha: float = 0.25, focal_gamma: float = 2.0):
        super().__init__()
        self.cost_class = cost_class
        self.cost_bbox = cost_bbox
        self.cost_giou = cost_giou
        self.focal_loss = focal_loss
        self.focal_alpha = focal_alpha
        self.focal_gamma = focal_gamma
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def forward(self, outputs, targets):
        """outputs: pred_logits [B,Q,C], pred_boxes [B,Q,4]; targets: list of dicts with labels,
        boxes (+ track_query_match_ids / track_queries_mask / track_queries_fal_pos_mask).
        -> list of (query indices, target indices) int64 tensors, one pair per sample."""
        return self.match_many([outputs], targets)[0]

    @torch.no_grad()
    def match_many(self, outputs_list, targets):
        """forward() for several prediction sets against the SAME targets (the criterion's final + auxiliary decoder layers:
        models/detr.py:266-289 of the reference calls the matcher once per layer) -> one list of index pairs per set.  The cost
        matrices of all sets are built by ONE chain of device kernels and reach the host in ONE copy (per call of forward(): ~25
        launches and three synchronising copies per target); the arithmetic per element is forward()'s, so are the assignments."""
        n_sets = len(outputs_list)
        batch_size, num_queries = outputs_list[0]["pred_logits"].shape[:2]
        logits = torch.stack([o["pred_logits"] for o in outputs_list]).flatten(0, 2)       # [sets * B * Q, C]
        out_prob = logits.sigmoid() if self.focal_loss else logits.softmax(-1)
        out_bbox = torch.stack([o["pred_boxes"] for o in outputs_list]).flatten(0, 2)
        tgt_ids = torch.cat([v["labels"] for v in targets])
        tgt_bbox = torch.cat([v["boxes"] for v in targets])

        if self.focal_loss:
            neg = (1 - self.focal_alpha) * (out_prob ** self.focal_gamma) \
                * (-(1 - out_prob + 1e-8).log())
            pos = self.focal_alpha * ((1 - out_prob) ** self.focal_gamma) \
                * (-(out_prob + 1e-8).log())
            cost_class = pos[:, tgt_ids] - neg[:, tgt_ids]
        else:
            cost_class = -out_prob[:, tgt_ids]
        cost_bbox = torch.cdist(out_bbox, tgt_bbox, p=1)
        cost_giou = -generalized_box_iou(box_cxcywh_to_xyxy(out_bbox),
                                         box_cxcywh_to_xyxy(tgt_bbox))
        cost_matrix = self.cost_bbox * cost_bbox + self.cost_class * cost_class \
            + self.cost_giou * cost_giou
        cost_all = cost_matrix.view(n_sets, batch_size, num_queries, -1).cpu()

        sizes = [len(v["boxes"]) for v in targets]
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        constraints = []   # per sample: (false-positive mask, pinned rows, pinned columns) on the host, once for all sets
        for i, target in enumerate(targets):
            if 'track_query_match_ids' not in target:
                constraints.append(None)
                continue
            fal_pos = target['track_queries_fal_pos_mask'].cpu()[:num_queries]
            is_track = target['track_queries_mask'].cpu()[:num_queries] & ~fal_pos
            rows = is_track.nonzero()[:, 0]
            cols = target['track_query_match_ids'].cpu()[:len(rows)] + int(offsets[i])
            constraints.append((fal_pos, rows, cols))
        result = []
        for s in range(n_sets):
            cost_matrix = cost_all[s]
            for i, con in enumerate(constraints):
                if con is None:
                    continue
                fal_pos, rows, cols = con
                # false positive track queries must stay unmatched; true ones are pinned to their target
                cost_matrix[i, fal_pos] = np.inf
                cost_matrix[i, rows] = np.inf
                cost_matrix[i][:, cols] = np.inf
                cost_matrix[i, rows, cols] = -1
            indices = [linear_sum_assignment(c[i])
                       for i, c in enumerate(cost_matrix.split(sizes, -1))]
            result.append([(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64))
                           for i, j in indices])
        return result


def build_matcher(args):
    return HungarianMatcher(cost_class=args.set_cost_class, cost_bbox=args.set_cost_bbox,
                            cost_giou=args.set_cost_giou, focal_loss=args.focal_loss,
                            focal_alpha=args.focal_alpha, focal_gamma=args.focal_gamma)
