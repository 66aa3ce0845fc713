"""DETR base detector, box head MLP and softmax post-processing.

Same surface as the reference's models/detr.py: DETR (:17-136), PostProcess (:446-490), MLP (:493-507).
(SetCriterion, the training loss, lives in criterion.py.)
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import box_ops, fused
from .nested import NestedTensor, nested_tensor_from_tensor_list


class MLP(nn.Module):
    """num_layers Linear layers with ReLU in between (the box regression head)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(i, o) for i, o in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = fused.head_linear(layer, x, relu=i < self.num_layers - 1)   # (the opt-in split product or layer(x) + ReLU)
        return x


class DETR(nn.Module):
    """Plain DETR: backbone (last level) -> 1x1 projection -> dense transformer -> class / box heads."""

    def __init__(self, backbone, transformer, num_classes, num_queries, aux_loss=False,
                 overflow_boxes=False):
        super().__init__()
        self.num_queries = num_queries
        self.transformer = transformer
        self.overflow_boxes = overflow_boxes
        self.class_embed = nn.Linear(self.hidden_dim, num_classes + 1)
        self.bbox_embed = MLP(self.hidden_dim, self.hidden_dim, 4, 3)
        self.query_embed = nn.Embedding(num_queries, self.hidden_dim)
        self.input_proj = nn.Conv2d(backbone.num_channels[-1], self.hidden_dim, kernel_size=1)
        self.backbone = backbone
        self.aux_loss = aux_loss

    @property
    def hidden_dim(self):
        return self.transformer.d_model

    @property
    def fpn_channels(self):
        return self.backbone.num_channels[:3][::-1]

    def forward(self, samples: NestedTensor, targets: list = None, prev_features=None):
        """-> (out, targets, features, memory, hs); out has pred_logits [B,Q,C+1], pred_boxes
        [B,Q,4] (cxcywh in [0,1]), hs_embed and (with aux_loss) aux_outputs.

        `prev_features` is accepted and ignored.  The reference's DETR.forward takes only
        (samples, targets) (detr.py:62) although Tracker.step and DETRTrackingBase.forward always pass
        three positionals (tracker.py:306, detr_tracking.py:275), which makes vanilla-DETR tracking
        raise TypeError there; this is the minimal signature extension documented in DESIGN.md."""
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_tensor_list(samples)
        features, pos = self.backbone(samples)
        src, mask = features[-1].decompose()
        assert mask is not None
        src = self.input_proj(src)
        batch_size = src.shape[0]

        query_embed = self.query_embed.weight.unsqueeze(1).repeat(1, batch_size, 1)
        tgt = None
        if targets is not None and 'track_query_hs_embeds' in targets[0]:
            track_hs = torch.stack([t['track_query_hs_embeds'] for t in targets])  # [B, T, C]
            num_track = track_hs.shape[1]
            query_embed = torch.cat([query_embed.new_zeros(num_track, batch_size, self.hidden_dim),
                                     query_embed], dim=0)
            tgt = torch.zeros_like(query_embed)
            tgt[:num_track] = track_hs.transpose(0, 1)
            for i, target in enumerate(targets):
                target['track_query_hs_embeds'] = tgt[:, i]

        hs, hs_without_norm, memory = self.transformer(src, mask, query_embed, pos[-1], tgt)
        outputs_class = self.class_embed(hs)
        outputs_coord = self.bbox_embed(hs).sigmoid()
        out = {'pred_logits': outputs_class[-1], 'pred_boxes': outputs_coord[-1],
               'hs_embed': hs_without_norm[-1]}
        if self.aux_loss:
            out['aux_outputs'] = self._set_aux_loss(outputs_class, outputs_coord)
        return out, targets, features, memory, hs

    @torch.jit.unused
    def _set_aux_loss(self, outputs_class, outputs_coord):
        return [{'pred_logits': a, 'pred_boxes': b}
                for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]


class PostProcess(nn.Module):
    """Softmax scores (excluding the trailing no-object class) and boxes scaled to the image size."""

    def process_boxes(self, boxes, target_sizes):
        boxes = box_ops.box_cxcywh_to_xyxy(boxes)
        img_h, img_w = target_sizes.unbind(1)
        scale_fct = torch.stack([img_w, img_h, img_w, img_h], dim=1)
        return boxes * scale_fct[:, None, :]

    @torch.no_grad()
    def forward(self, outputs, target_sizes, results_mask=None):
        out_logits, out_bbox = outputs['pred_logits'], outputs['pred_boxes']
        assert len(out_logits) == len(target_sizes)
        assert target_sizes.shape[1] == 2
        prob = F.softmax(out_logits, -1)
        scores, labels = prob[..., :-1].max(-1)
        boxes = self.process_boxes(out_bbox, target_sizes)
        results = [{'scores': s, 'labels': l, 'boxes': b, 'scores_no_object': n}
                   for s, l, b, n in zip(scores, labels, boxes, prob[..., -1])]
        if results_mask is not None:
            for i, mask in enumerate(results_mask):
                results[i] = {k: v[mask] for k, v in results[i].items()}
        return results
