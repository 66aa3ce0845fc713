"""Track-query mix-in: turns a detector into TrackFormer's tracking-by-attention model.

Same surface as the reference's models/detr_tracking.py: DETRTrackingBase (.train/.tracking/.forward,
add_track_queries_to_targets), DETRTracking, DeformableDETRTracking (:281-290) with the
(tracking_kwargs, detr_kwargs) constructor convention of build_model.
"""
from contextlib import nullcontext

import torch
import torch.nn as nn

from .deformable_detr import DeformableDETR
from .detr import DETR
from .nested import NestedTensor
from .track_queries import add_track_queries_to_targets


class DETRTrackingBase(nn.Module):
    def __init__(self, track_query_false_positive_prob: float = 0.0,
                 track_query_false_negative_prob: float = 0.0, matcher=None,
                 backprop_prev_frame=False):
        # NB: like the reference this does not call nn.Module.__init__ -- the detector base class,
        # initialised first by the concrete subclasses, already did.
        self._matcher = matcher
        self._track_query_false_positive_prob = track_query_false_positive_prob
        self._track_query_false_negative_prob = track_query_false_negative_prob
        self._backprop_prev_frame = backprop_prev_frame
        self._tracking = False

    def train(self, mode: bool = True):
        """Leaving tracking mode whenever the train/eval state is set (detr_tracking.py:29-32)."""
        self._tracking = False
        return super().train(mode)

    def tracking(self):
        """Inference mode used by Tracker.step: eval + targets carry track queries only."""
        self.eval()
        self._tracking = True

    def add_track_queries_to_targets(self, targets, prev_indices, prev_out, add_false_pos=True):
        return add_track_queries_to_targets(
            targets, prev_indices, prev_out, add_false_pos=add_false_pos,
            false_positive_prob=self._track_query_false_positive_prob,
            false_negative_prob=self._track_query_false_negative_prob,
            num_queries=self.num_queries)

    def forward(self, samples: NestedTensor, targets: list = None, prev_features=None, encoded=None):
        if targets is not None and not self._tracking:
            prev_targets = [target['prev_target'] for target in targets]
            if self.training:
                # previous frame(s) first -- without gradients unless configured otherwise -- to
                # obtain the output embeddings that become this frame's track queries
                ctx = nullcontext if self._backprop_prev_frame else torch.no_grad
                with ctx():
                    if 'prev_prev_image' in targets[0]:
                        for target, prev_target in zip(targets, prev_targets):
                            prev_target['prev_target'] = target['prev_prev_target']
                        prev_prev_targets = [target['prev_prev_target'] for target in targets]
                        prev_prev_out, _, prev_prev_features, _, _ = super().forward(
                            [t['prev_prev_image'] for t in targets])
                        no_aux = {k: v for k, v in prev_prev_out.items() if 'aux_outputs' not in k}
                        prev_prev_indices = self._matcher(no_aux, prev_prev_targets)
                        self.add_track_queries_to_targets(
                            prev_targets, prev_prev_indices, prev_prev_out, add_false_pos=False)
                        prev_out, _, prev_features, _, _ = super().forward(
                            [t['prev_image'] for t in targets], prev_targets, prev_prev_features)
                    else:
                        prev_out, _, prev_features, _, _ = super().forward(
                            [t['prev_image'] for t in targets])
                    no_aux = {k: v for k, v in prev_out.items() if 'aux_outputs' not in k}
                    prev_indices = self._matcher(no_aux, prev_targets)
                    self.add_track_queries_to_targets(targets, prev_indices, prev_out)
            else:
                # plain detection evaluation: empty track-query fields (detr_tracking.py:262-273)
                for target in targets:
                    device = target['boxes'].device
                    target['track_query_hs_embeds'] = torch.zeros(0, self.hidden_dim,
                                                                  device=device)
                    target['track_queries_mask'] = torch.zeros(self.num_queries, dtype=torch.bool,
                                                               device=device)
                    target['track_queries_fal_pos_mask'] = torch.zeros(
                        self.num_queries, dtype=torch.bool, device=device)
                    target['track_query_boxes'] = torch.zeros(0, 4, device=device)
                    target['track_query_match_ids'] = torch.zeros(0, dtype=torch.long,
                                                                  device=device)
        if encoded is not None:   # (Tracker.step_prepare: the image-only half of this frame has already run)
            return super().forward(samples, targets, prev_features, encoded=encoded)
        return super().forward(samples, targets, prev_features)


class DETRTracking(DETRTrackingBase, DETR):
    def __init__(self, tracking_kwargs, detr_kwargs):
        DETR.__init__(self, **detr_kwargs)
        DETRTrackingBase.__init__(self, **tracking_kwargs)


class DeformableDETRTracking(DETRTrackingBase, DeformableDETR):
    def __init__(self, tracking_kwargs, detr_kwargs):
        DeformableDETR.__init__(self, **detr_kwargs)
        DETRTrackingBase.__init__(self, **tracking_kwargs)
