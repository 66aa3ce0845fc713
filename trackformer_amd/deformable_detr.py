"""Deformable DETR detector (multi-level, optional multi-frame attention) and its post-processing.

Same surface as the reference's models/deformable_detr.py: DeformableDETR (:29-283) -- constructor
arguments, attributes, state_dict layout (`input_proj.L.{0,1}`, `class_embed.N`, `bbox_embed.N.layers.K`,
`query_embed`, `transformer...`), forward(samples, targets, prev_features) ->
(out, targets, features_all, memory, hs) -- and DeformablePostProcess (:286-334).
"""
import copy
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops, fused
from .detr import DETR, PostProcess
from .nested import (NestedTensor, all_valid_mask, inverse_sigmoid, is_all_valid,
                     nested_tensor_from_tensor_list)


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class DeformableDETR(DETR):
    # GraphedDetector may append filler track queries (masked as self-attention keys, deformable_transformer.py)
    track_query_filler_ok = True

    def __init__(self, backbone, transformer, num_classes, num_queries, num_feature_levels,
                 aux_loss=True, with_box_refine=False, two_stage=False, overflow_boxes=False,
                 multi_frame_attention=False, multi_frame_encoding=False,
                 merge_frame_features=False):
        super().__init__(backbone, transformer, num_classes, num_queries, aux_loss)
        self.merge_frame_features = merge_frame_features
        self.multi_frame_attention = multi_frame_attention
        self.multi_frame_encoding = multi_frame_encoding
        self.overflow_boxes = overflow_boxes
        self.num_feature_levels = num_feature_levels
        if not two_stage:
            self.query_embed = nn.Embedding(num_queries, self.hidden_dim * 2)  # (query_pos | tgt)

        def proj(in_ch, **conv_kw):
            return nn.Sequential(nn.Conv2d(in_ch, self.hidden_dim, **conv_kw),
                                 nn.GroupNorm(32, self.hidden_dim))

        num_channels = backbone.num_channels[-3:]
        if num_feature_levels > 1:
            num_backbone_outs = len(backbone.strides) - 1  # layer2..layer4
            projs = [proj(num_channels[i], kernel_size=1) for i in range(num_backbone_outs)]
            in_ch = num_channels[num_backbone_outs - 1]
            for _ in range(num_feature_levels - num_backbone_outs):  # extra stride-2 levels
                projs.append(proj(in_ch, kernel_size=3, stride=2, padding=1))
                in_ch = self.hidden_dim
            self.input_proj = nn.ModuleList(projs)
        else:
            self.input_proj = nn.ModuleList([proj(num_channels[0], kernel_size=1)])
        self.with_box_refine = with_box_refine
        self.two_stage = two_stage

        prior_prob = 0.01
        self.class_embed.bias.data.fill_(-math.log((1 - prior_prob) / prior_prob))
        nn.init.constant_(self.bbox_embed.layers[-1].weight.data, 0)
        nn.init.constant_(self.bbox_embed.layers[-1].bias.data, 0)
        for p in self.input_proj:
            nn.init.xavier_uniform_(p[0].weight, gain=1)
            nn.init.constant_(p[0].bias, 0)

        # one head per decoder layer (+1 for the two-stage proposal head)
        num_pred = transformer.decoder.num_layers + (1 if two_stage else 0)
        if with_box_refine:
            self.class_embed = _get_clones(self.class_embed, num_pred)
            self.bbox_embed = _get_clones(self.bbox_embed, num_pred)
            nn.init.constant_(self.bbox_embed[0].layers[-1].bias.data[2:], -2.0)
            self.transformer.decoder.bbox_embed = self.bbox_embed  # iterative refinement
        else:
            nn.init.constant_(self.bbox_embed.layers[-1].bias.data[2:], -2.0)
            self.class_embed = nn.ModuleList([self.class_embed for _ in range(num_pred)])
            self.bbox_embed = nn.ModuleList([self.bbox_embed for _ in range(num_pred)])
            self.transformer.decoder.bbox_embed = None
        if two_stage:
            self.transformer.decoder.class_embed = self.class_embed
            for box_embed in self.bbox_embed:
                nn.init.constant_(box_embed.layers[-1].bias.data[2:], 0.0)

        if self.merge_frame_features:
            merge = nn.Conv2d(self.hidden_dim * 2, self.hidden_dim, kernel_size=1)
            self.merge_features = _get_clones(merge, num_feature_levels)

    # ------------------------------------------------------------------------------------------
    def _input_proj(self, level, x):
        proj = self.input_proj[level]
        if not self.training and not torch.is_grad_enabled() and x.is_cuda:
            y = fused.input_proj_1x1(x, proj[0], proj[1])   # opt-in route; None = not applicable / switched off
            if y is not None:
                return y
        return proj(x)

    def _project(self, level, x, prev_x=None):
        y = self._input_proj(level, x)
        if self.merge_frame_features:
            y = self.merge_features[level](torch.cat([y, self._input_proj(level, prev_x)], dim=1))
        return y

    def forward(self, samples: NestedTensor, targets: list = None, prev_features=None, encoded=None):
        """samples: NestedTensor / list of images / [B,3,H,W] tensor.  targets (tracking): list of
        dicts with `track_query_hs_embeds` [T,C] and `track_query_boxes` [T,4].  prev_features: the
        `features_all` returned for the previous frame (multi-frame attention).
        encoded: what encode_frame(samples, prev_features) returned for THIS frame (optional; not in the reference): the
        image-only half -- backbone, input projections, encoder -- is then not run again.

        Returns (out, targets, features_all, memory, hs) as deformable_detr.py:275; `out` holds
        pred_logits [B,Q,C], pred_boxes [B,Q,4] (cxcywh, [0,1]), hs_embed [B,Q,hidden], aux_outputs."""
        if encoded is None:
            encoded = self.encode_frame(samples, prev_features)
        features_all, src_shapes, enc = encoded["features_all"], encoded["src_shapes"], encoded["enc"]

        query_embeds = None if self.two_stage else self.query_embed.weight
        hs, memory, init_reference, inter_references, enc_outputs_class, enc_outputs_coord_unact = \
            self.transformer.decode(enc, query_embeds, targets)
        return self._heads(hs, memory, init_reference, inter_references, enc_outputs_class, enc_outputs_coord_unact,
                           targets, features_all, src_shapes)

    def encode_frame(self, samples, prev_features=None):
        """Everything of forward() that depends on the image (and, with multi-frame attention, the previous frame's features)
        only: backbone, input projections, position encodings, the encoder (deformable_detr.py:124-223 +
        deformable_transformer.py:133-173 of the reference).  The track queries of `targets` enter afterwards (the decoder),
        so a tracker can run this half for frame t + 1 while it still associates frame t (Tracker.step_prepare)."""
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_tensor_list(samples)
        features_all, pos = self.backbone(samples)
        features = features_all[-3:]
        prev_features = features if prev_features is None else prev_features[-3:]

        frames = [prev_features, features] if self.multi_frame_attention else [features]
        src_list, mask_list, pos_list = [], [], []
        for frame, frame_feat in enumerate(frames):
            per_frame_pos = self.multi_frame_attention and self.multi_frame_encoding
            pos_list.extend([p[:, frame] for p in pos[-3:]] if per_frame_pos else pos[-3:])
            for lvl, feat in enumerate(frame_feat):
                src, mask = feat.decompose()
                assert mask is not None
                prev_src = prev_features[lvl].tensors if self.merge_frame_features else None
                src_list.append(self._project(lvl, src, prev_src))
                mask_list.append(mask)

            for lvl in range(len(frame_feat), self.num_feature_levels):  # extra coarse levels
                if lvl == len(frame_feat):
                    src = self._project(lvl, frame_feat[-1].tensors,
                                        prev_features[-1].tensors if self.merge_frame_features
                                        else None)
                else:
                    src = self.input_proj[lvl](src_list[-1])
                m = frame_feat[0].mask
                if is_all_valid(m):
                    mask = all_valid_mask((m.shape[0],) + tuple(src.shape[-2:]), src.device)
                else:
                    mask = F.interpolate(m[None].float(), size=src.shape[-2:]).to(torch.bool)[0]
                pos_l = self.backbone[1](NestedTensor(src, mask)).to(src.dtype)
                src_list.append(src)
                mask_list.append(mask)
                pos_list.append(pos_l[:, frame] if per_frame_pos else pos_l)

        return {"features_all": features_all, "src_shapes": [tuple(s.shape[-2:]) for s in src_list],
                "enc": self.transformer.encode(src_list, mask_list, pos_list)}

    def _heads(self, hs, memory, init_reference, inter_references, enc_outputs_class, enc_outputs_coord_unact, targets,
               features_all, src_shapes):
        reuse_refinement = (self.with_box_refine and not self.two_stage and not self.training
                            and not torch.is_grad_enabled()
                            and inter_references.shape[0] == hs.shape[0])
        if reuse_refinement:
            # With iterative box refinement the decoder has already evaluated
            # sigmoid(bbox_embed[l](hs[l]) + inverse_sigmoid(reference_l)) -- the very expression of
            # the box head (deformable_transformer.py:412-422 vs deformable_detr.py:238-246, same
            # modules, same inputs): the refined reference points ARE the per-layer box predictions.
            outputs_coord = inter_references
            outputs_class = torch.stack([fused.head_linear(self.class_embed[lvl], hs[lvl])
                                         for lvl in range(hs.shape[0])])
        else:
            outputs_classes, outputs_coords = [], []
            for lvl in range(hs.shape[0]):
                reference = inverse_sigmoid(init_reference if lvl == 0
                                            else inter_references[lvl - 1])
                outputs_classes.append(self.class_embed[lvl](hs[lvl]))
                tmp = self.bbox_embed[lvl](hs[lvl])
                if reference.shape[-1] == 4:
                    tmp += reference
                else:
                    assert reference.shape[-1] == 2
                    tmp[..., :2] += reference
                outputs_coords.append(tmp.sigmoid())
            outputs_class = torch.stack(outputs_classes)
            outputs_coord = torch.stack(outputs_coords)

        out = {'pred_logits': outputs_class[-1], 'pred_boxes': outputs_coord[-1],
               'hs_embed': hs[-1]}
        if self.aux_loss:
            out['aux_outputs'] = self._set_aux_loss(outputs_class, outputs_coord)
        if self.two_stage:
            out['enc_outputs'] = {'pred_logits': enc_outputs_class,
                                  'pred_boxes': enc_outputs_coord_unact.sigmoid()}

        # encoder memory re-sliced into per-level [B, C, H, W] views
        batch_size, _, channels = memory.shape
        memory_slices, offset = [], 0
        for h, w in src_shapes:
            memory_slices.append(memory[:, offset:offset + h * w].permute(0, 2, 1).view(
                batch_size, channels, h, w))
            offset += h * w
        return out, targets, features_all, memory_slices, hs


class DeformablePostProcess(PostProcess):
    """Sigmoid scores: per query the best class and its score; boxes scaled to the image size."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes, results_mask=None):
        out_logits, out_bbox = outputs['pred_logits'], outputs['pred_boxes']
        assert len(out_logits) == len(target_sizes)
        assert target_sizes.shape[1] == 2
        scores, labels = out_logits.sigmoid().max(-1)
        boxes = self.process_boxes(out_bbox, target_sizes)
        results = [{'scores': s, 'scores_no_object': 1 - s, 'labels': l, 'boxes': b}
                   for s, l, b in zip(scores, labels, boxes)]
        if results_mask is not None:
            for i, mask in enumerate(results_mask):
                results[i] = {k: v[mask] for k, v in results[i].items()}
        return results
