"""Instance-mask head on top of the (Deformable) DETR detectors (BASELINE cfg 5: MOTS20).

Same surface as the reference's models/detr_segmentation.py:
    DETRSegmBase (:29-71), DETRSegm / DeformableDETRSegm / DETRSegmTracking / DeformableDETRSegmTracking
    (:75-99) with the (mask_kwargs, [tracking_kwargs,] detr_kwargs) constructor convention of
    build_model, MaskHeadSmallConv (:106-181), MHAttentionMap (:184-222), PostProcessSegm (:225-263),
    PostProcessPanoptic (:266-388) -- same parameter names, so reference checkpoints load unchanged
    (`bbox_attention.{q,k}_linear`, `mask_head.{lay1..5,gn1..5,out_lay,adapter1..3}`).

Differences, all result-preserving:
  * forward() takes the `prev_features` argument the tracker passes (tracker.py:307 calls the
    detector with three arguments; the reference's DETRSegmBase.forward only accepts two and cannot
    be driven by its own Tracker) and hands it through to the detector.
  * MaskHeadSmallConv never materialises the per-query copies of the image features
    (`_expand`, :102-103): the first convolution is linear, so its image part is evaluated once per
    image and only the 8 attention-map channels are convolved per query (33x fewer flops in that
    layer for hidden 256), and the FPN adapters are broadcast over the queries instead of repeated.
  * PostProcessSegm keeps the masks on the device of the model outputs (the reference moves all
    Q masks to the host before resizing them).
"""
import contextlib
import io
import os
import threading
from collections import defaultdict
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import box_ops
from .deformable_detr import DeformableDETR
from .detr import DETR
from .detr_tracking import DETRTrackingBase
from .nested import NestedTensor


class MaskContext:
    """What the mask head needs besides the queries (all of it per image, none of it per query): the projected image
    features, their padding mask, the FPN inputs and the encoder memory.  Returned under out["mask_context"] in place of
    out["pred_masks"] when the detector runs with `lazy_masks` (DETRSegmBase.mask_rows evaluates the head later, for
    the queries somebody actually wants)."""
    __slots__ = ("src", "mask", "fpns", "memory")

    def __init__(self, src, mask, fpns, memory):
        self.src, self.mask, self.fpns, self.memory = src, mask, fpns, memory


_LAZY_SCOPE = threading.local()


@contextlib.contextmanager
def lazy_mask_scope(on=True):
    """Within the block (this thread only) a DETRSegm* forward under no_grad returns out["mask_context"] instead of
    out["pred_masks"]; DETRSegmBase.mask_rows evaluates the head for chosen queries afterwards."""
    prev = getattr(_LAZY_SCOPE, "on", False)
    _LAZY_SCOPE.on = bool(on)
    try:
        yield
    finally:
        _LAZY_SCOPE.on = prev


class DETRSegmBase(nn.Module):
    """Mix-in: adds `bbox_attention` + `mask_head` to a detector and `pred_masks` [B,Q,H/4,W/4]
    (deformable: stride of backbone layer1) to its outputs."""

    def __init__(self, freeze_detr=False):
        # like the reference this does not call nn.Module.__init__: the detector base class,
        # initialised first by the concrete subclasses, already did
        if freeze_detr:
            for param in self.parameters():
                param.requires_grad_(False)
        nheads = self.transformer.nhead
        self.bbox_attention = MHAttentionMap(self.hidden_dim, self.hidden_dim, nheads, dropout=0.0)
        self.mask_head = MaskHeadSmallConv(self.hidden_dim + nheads, self.fpn_channels,
                                           self.hidden_dim)

    def forward(self, samples: NestedTensor, targets: list = None, prev_features=None, encoded=None):
        # encoded: what encode_frame() returned for this frame (the image-only half ran ahead: GraphedDetector.prepare /
        # Tracker.step_prepare); everything the mask head reads -- features, encoder memory -- is part of it
        if encoded is not None:
            out, targets, features, memory, hs = super().forward(samples, targets, prev_features, encoded=encoded)
        else:
            out, targets, features, memory, hs = super().forward(samples, targets, prev_features)

        if isinstance(memory, list):   # deformable: per-level encoder memory
            src, mask = features[-2].decompose()
            src = self.input_proj[-3](src)
            mask = F.interpolate(mask[None].float(), size=src.shape[-2:]).to(torch.bool)[0]
            fpns = [features[-2].tensors, features[-3].tensors, features[-4].tensors]
            memory = memory[-3]
        else:
            src, mask = features[-1].decompose()
            src = self.input_proj(src)
            fpns = [features[2].tensors, features[1].tensors, features[0].tensors]

        if self.lazy_masks_active() and not self.training and not torch.is_grad_enabled():
            # OPT-IN (Tracker(lazy_masks=True) / TF_LAZY_MASKS=1): a query's mask depends on that query alone (attention map
            # and GroupNorm are per (image, query) sample), and the tracker only ever reads the masks of its surviving
            # tracks -- the head (3.3 TFLOP for 400 queries at 800 x 1333, 41 of cfg 5's 47 ms) is evaluated for those rows
            # only, by whoever holds the context.
            out["mask_context"] = MaskContext(src, mask, fpns, memory)
            return out, targets, features, memory, hs
        bbox_mask = self.bbox_attention(hs[-1], memory, mask=mask)     # [B, Q, heads, h, w]
        seg_masks = self.mask_head(src, bbox_mask, fpns)               # [B*Q, 1, H, W]
        out["pred_masks"] = seg_masks.view(src.shape[0], hs.shape[2], seg_masks.shape[-2],
                                           seg_masks.shape[-1])
        return out, targets, features, memory, hs

    lazy_masks = False     # permanent opt-in of a model instance (every no_grad caller then gets `mask_context`)

    def lazy_masks_active(self) -> bool:
        """True when this call leaves the mask head to the caller: the instance switch, or the calling thread is inside
        `lazy_mask_scope()` (what Tracker uses: the switch is scoped to its own detector call, a PostProcessSegm consumer
        or a second tracker sharing the module is not affected)."""
        return bool(self.lazy_masks or getattr(_LAZY_SCOPE, "on", False))

    def mask_rows(self, ctx: "MaskContext", hs_rows: Tensor) -> Tensor:
        """pred_masks of the queries whose last-layer decoder outputs are hs_rows [B, n, C] -> [B, n, H, W]: the same
        arithmetic as forward() on those rows (bbox_attention and mask_head treat every query independently)."""
        bbox_mask = self.bbox_attention(hs_rows, ctx.memory, mask=ctx.mask)
        seg = self.mask_head(ctx.src, bbox_mask, ctx.fpns)
        return seg.view(ctx.src.shape[0], hs_rows.shape[1], seg.shape[-2], seg.shape[-1])


class DETRSegm(DETRSegmBase, DETR):
    def __init__(self, mask_kwargs, detr_kwargs):
        DETR.__init__(self, **detr_kwargs)
        DETRSegmBase.__init__(self, **mask_kwargs)


class DeformableDETRSegm(DETRSegmBase, DeformableDETR):
    def __init__(self, mask_kwargs, detr_kwargs):
        DeformableDETR.__init__(self, **detr_kwargs)
        DETRSegmBase.__init__(self, **mask_kwargs)


class DETRSegmTracking(DETRSegmBase, DETRTrackingBase, DETR):
    def __init__(self, mask_kwargs, tracking_kwargs, detr_kwargs):
        DETR.__init__(self, **detr_kwargs)
        DETRTrackingBase.__init__(self, **tracking_kwargs)
        DETRSegmBase.__init__(self, **mask_kwargs)


class DeformableDETRSegmTracking(DETRSegmBase, DETRTrackingBase, DeformableDETR):
    def __init__(self, mask_kwargs, tracking_kwargs, detr_kwargs):
        DeformableDETR.__init__(self, **detr_kwargs)
        DETRTrackingBase.__init__(self, **tracking_kwargs)
        DETRSegmBase.__init__(self, **mask_kwargs)


_mask_head_split = os.environ.get("TF_MASK_HEAD_SPLIT", "1") != "0"
_SPLIT_QUERY_CHUNK = 128


def set_mask_head_split(on):
    """The mask head's 3 x 3 convolutions through the split-product kernels in the GPU inference path (process-wide); returns the
    previous setting."""
    global _mask_head_split
    prev, _mask_head_split = _mask_head_split, bool(on)
    return prev


# DEFAULT since round 6 (TF_MASK_HEAD_FUSED_TAIL=0 / set_mask_head_fused_tail(False)): the FPN merges and the end of the mask head
# through their own one-pass kernels (fused.upsample_add, fused.groupnorm_relu_conv3x3_c1) on the split-product route.
_mask_head_fused_tail = os.environ.get("TF_MASK_HEAD_FUSED_TAIL", "1") != "0"


def set_mask_head_fused_tail(on):
    global _mask_head_fused_tail
    prev, _mask_head_fused_tail = _mask_head_fused_tail, bool(on)
    return prev


class MaskHeadSmallConv(nn.Module):
    """Small FPN-style convolutional head with GroupNorm: [image features | attention maps] at stride
    16 (deformable; 32 for plain DETR) -> one mask logit map per query at the stride of fpns[2]."""

    def __init__(self, dim, fpn_dims, context_dim):
        super().__init__()
        inter = [dim, context_dim // 2, context_dim // 4, context_dim // 8, context_dim // 16,
                 context_dim // 64]
        self.lay1 = nn.Conv2d(dim, dim, 3, padding=1)
        self.gn1 = nn.GroupNorm(8, dim)
        self.lay2 = nn.Conv2d(dim, inter[1], 3, padding=1)
        self.gn2 = nn.GroupNorm(8, inter[1])
        self.lay3 = nn.Conv2d(inter[1], inter[2], 3, padding=1)
        self.gn3 = nn.GroupNorm(8, inter[2])
        self.lay4 = nn.Conv2d(inter[2], inter[3], 3, padding=1)
        self.gn4 = nn.GroupNorm(8, inter[3])
        self.lay5 = nn.Conv2d(inter[3], inter[4], 3, padding=1)
        self.gn5 = nn.GroupNorm(8, inter[4])
        self.out_lay = nn.Conv2d(inter[4], 1, 3, padding=1)
        self.dim = dim
        self.adapter1 = nn.Conv2d(fpn_dims[0], inter[1], 1)
        self.adapter2 = nn.Conv2d(fpn_dims[1], inter[2], 1)
        self.adapter3 = nn.Conv2d(fpn_dims[2], inter[3], 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def _merge(x, fpn, num_queries):
        """fpn [B,C,H,W] (or already [B*Q,C,H,W]) + nearest-upsampled x [B*Q,C,h,w]; the adapter
        output is broadcast over the queries (the reference repeats it, :154-156)."""
        x = F.interpolate(x, size=fpn.shape[-2:], mode="nearest")
        if fpn.size(0) == x.size(0):
            return fpn + x
        return (x.view(fpn.size(0), num_queries, *x.shape[1:]) + fpn[:, None]).flatten(0, 1)

    # ---- the GPU inference route (round 4; TF_MASK_HEAD_SPLIT=0 / set_mask_head_split(False) switches it off): lay2 .. lay5 --
    # 97 % of the head's flops, ~17 of the 20.6 ms of a cfg-5 step in the library convolutions -- through the split-product
    # convolution kernels (fused.conv3x3: the fp16 product by default) on channels_last activations, their GroupNorms through
    # tf_groupnorm_nhwc_f32.  lay2's 264 input channels are padded to 288 with zeros (the kernels take Cin % 32 == 0); lay1
    # (8 attention channels per query after the decomposition below) and out_lay (16 -> 1) stay in the library.  A layer the
    # kernels decline (tensor beyond their 3 GiB offsets: more than ~310 queries at 200 x 334) takes the library path.
    @staticmethod
    def _taps(conv, cin_pad):
        """[Cout, 9 * cin_pad] tap-major weight of a 3 x 3 convolution (zero columns for the padded input channels), cached on the
        module with the weight's version."""
        hit = getattr(conv, "_tf_taps", None)
        if hit is None or hit[0] != (conv.weight._version, cin_pad) or hit[1].device != conv.weight.device:
            w = conv.weight.detach()
            cout, cin = w.shape[:2]
            taps = w.new_zeros(cout, 3, 3, cin_pad)
            taps[..., :cin] = w.permute(0, 2, 3, 1)
            hit = ((conv.weight._version, cin_pad), taps.reshape(cout, 9 * cin_pad).contiguous())
            if w.is_cuda and torch.cuda.is_current_stream_capturing():
                return hit[1]   # built inside a graph's memory pool: part of that graph, never a cached buffer
            if w.is_cuda:
                from . import fused
                fused._publish_barrier(w.device)   # built on this stream, read by every sequence's stream from now on
            conv._tf_taps = hit
        return hit[1]

    def _merged_levels(self, xp, feats, num_queries):
        """lay2 .. out_lay with NO intermediate activation written in normalised or up-sampled form (round 6): every convolution
        leaves its RAW output and one statistics pass over it; the next convolution applies GroupNorm + ReLU, the nearest
        up-sampling and the adapter's broadcast add in its own fetch (fused.conv3x3_merged), the last one feeds gn5 + ReLU +
        out_lay (fused.groupnorm_relu_conv3x3_c1).  Per level that is a read of the low-resolution tensor instead of a
        normalise pass (read + write), a merge pass (read + 4x write) and a read of the merged tensor.  None: a kernel declined
        (the caller runs the pass-by-pass route)."""
        from . import fused
        y = fused.conv3x3(xp, self._taps(self.lay2, xp.shape[1]), self.lay2.bias, False, 1)
        if y is None:
            return None
        gn_prev = self.gn2
        for conv, gn, feat in ((self.lay3, self.gn3, feats[0]), (self.lay4, self.gn4, feats[1]), (self.lay5, self.gn5, feats[2])):
            ws = fused.groupnorm_stats(y, gn_prev)
            if ws is None:
                return None
            y = fused.conv3x3_merged(y, feat, num_queries, self._taps(conv, conv.in_channels), conv.bias, gn=gn_prev, ws=ws)
            if y is None:
                return None
            gn_prev = gn
        return fused.groupnorm_relu_conv3x3_c1(y, self.gn5, self.out_lay)

    def _taps_part(self, c0, c1):
        """[Cout, 9 * (c1 - c0)] tap-major weight of lay1 restricted to its input channels c0 .. c1 (the image part of the decomposition in
        forward()), cached on the module with the weight's version."""
        conv = self.lay1
        hit = getattr(conv, "_tf_taps_part", None)
        key = (conv.weight._version, c0, c1)
        if hit is None or hit[0] != key or hit[1].device != conv.weight.device:
            w = conv.weight.detach()[:, c0:c1]
            hit = (key, w.permute(0, 2, 3, 1).reshape(w.shape[0], 9 * (c1 - c0)).contiguous())
            if w.is_cuda and torch.cuda.is_current_stream_capturing():
                return hit[1]   # built inside a graph's memory pool: part of that graph, never a cached buffer
            if w.is_cuda:
                from . import fused
                fused._publish_barrier(w.device)
            conv._tf_taps_part = hit
        return hit[1]

    def _conv_gn_relu(self, x, conv, gn):
        """relu(gn(conv(x))) for a channels_last x [N, Cin(_pad), H, W]; -> channels_last [N, Cout, H, W]."""
        from . import fused
        cin_pad = x.shape[1]
        y = fused.conv3x3(x, self._taps(conv, cin_pad), conv.bias, False, 1)
        if y is None:     # declined (size / alignment): the library convolution on the unpadded channels
            y = F.conv2d(x[:, :conv.in_channels], conv.weight, conv.bias, padding=1).contiguous(memory_format=torch.channels_last)
        n, c, h, w = y.shape
        z = fused.groupnorm_nhwc(y.permute(0, 2, 3, 1).reshape(n * h * w, c), n, gn, relu=True)   # GroupNorm + ReLU in one pass
        if z is None:
            return F.relu(gn(y))
        return z.view(n, h, w, c).permute(0, 3, 1, 2)

    def _split_route(self, x):
        from . import fused
        return (_mask_head_split and fused.split_linear_enabled() and x.is_cuda and x.dtype == torch.float32
                and not torch.is_grad_enabled() and self.lay2.in_channels <= 288 and self.lay3.in_channels % 32 == 0
                and self.lay4.in_channels % 32 == 0 and self.lay5.in_channels % 32 == 0)

    def forward(self, x: Tensor, bbox_mask: Tensor, fpns: List[Tensor]):
        """x [B,C,h,w] projected image features, bbox_mask [B,Q,heads,h,w], fpns 3 x [B,C_k,H_k,W_k]
        (coarse to fine) -> [B*Q, 1, H_2, W_2]."""
        num_queries = bbox_mask.shape[1]
        c_img = x.shape[1]
        split = self._split_route(x)
        # What does not depend on the query is computed ONCE per image: lay1 over cat([x repeated per query, attention maps])
        # == lay1_img(x) + lay1_att(maps) -- the image part (and the bias) here, the attention part per query below -- and
        # the three adapter convolutions of the backbone features
        y_img = None
        if split and _mask_head_fused_tail and c_img % 32 == 0:
            # lay1's image part through the split-product convolution as well (the library picked a grouped-convolution kernel that
            # ran this 1 050-pixel, 256 -> 264 layer at ~5 TFLOP/s: 0.27 ms per frame)
            from . import fused
            y_img = fused.conv3x3(x.contiguous(memory_format=torch.channels_last), self._taps_part(0, c_img), self.lay1.bias, False, 1)
        if y_img is None:
            y_img = F.conv2d(x, self.lay1.weight[:, :c_img], self.lay1.bias, padding=1)             # [B, dim, h, w]
        feats = [self.adapter1(fpns[0]), self.adapter2(fpns[1]), self.adapter3(fpns[2])]
        if split:
            feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
            if x.shape[0] * num_queries > _SPLIT_QUERY_CHUNK:
                # (image, query) pairs are independent: chunks of one image's queries keep every activation of the route below
                # the kernels' 3 GiB offsets (the finest level is 8.6 MB per query and channel group at 800 x 1333) and bound
                # the head's memory, for any batch size
                return torch.cat([self._per_query(y_img[b:b + 1], bbox_mask[b:b + 1, q0:q0 + _SPLIT_QUERY_CHUNK],
                                                  [f[b:b + 1] for f in feats], c_img, True)
                                  for b in range(x.shape[0]) for q0 in range(0, num_queries, _SPLIT_QUERY_CHUNK)], 0)
        return self._per_query(y_img, bbox_mask, feats, c_img, split)

    def _per_query(self, y_img, bbox_mask, feats, c_img, split):
        """The per-query part of the head: y_img [B, dim, h, w] (lay1's image part), bbox_mask [B, Q', heads, h, w],
        feats 3 x [B, C_k, H_k, W_k] (the adapters' outputs) -> [B * Q', 1, H_2, W_2]."""
        batch, num_queries = bbox_mask.shape[:2]
        xp = None
        if split and _mask_head_fused_tail:
            # round 6: the front of the per-query part channels-innermost from the start -- lay1's attention part (library
            # convolution, NHWC in and out), + its image part broadcast over the queries (fused.upsample_add at equal sizes: one
            # coalesced pass), GroupNorm + ReLU in the library's own two passes -- instead of NCHW element-wise passes, ATen's
            # GroupNorm, a ReLU pass and a strided permute-copy into the padded buffer
            from . import fused
            y_att = F.conv2d(bbox_mask.flatten(0, 1).contiguous(memory_format=torch.channels_last), self.lay1.weight[:, c_img:], None, padding=1)
            y_att = y_att.contiguous(memory_format=torch.channels_last)
            z = fused.upsample_add(y_att, y_img.contiguous(memory_format=torch.channels_last), num_queries)
            if z is not None:
                n, c, h, wd = z.shape
                z2 = fused.groupnorm_nhwc(z.permute(0, 2, 3, 1).reshape(n * h * wd, c), n, self.gn1, relu=True)
                if z2 is not None:
                    cin_pad = -(-c // 32) * 32
                    xp = z2.new_zeros(n, h, wd, cin_pad)                             # channels innermost, zero tail
                    xp[..., :c] = z2.view(n, h, wd, c)
        if xp is None:
            y_att = F.conv2d(bbox_mask.flatten(0, 1), self.lay1.weight[:, c_img:], None, padding=1)     # [B*Q', dim, h, w]
            x = (y_att.view(batch, num_queries, *y_att.shape[1:]) + y_img[:, None]).flatten(0, 1)
            x = F.relu(self.gn1(x))
        if split:
            from . import fused
            if xp is None:
                n, c, h, wd = x.shape
                cin_pad = -(-c // 32) * 32
                xp = x.new_zeros(n, h, wd, cin_pad)                                  # channels innermost, zero tail
                xp[..., :c] = x.permute(0, 2, 3, 1)
            if _mask_head_fused_tail:
                out = self._merged_levels(xp.permute(0, 3, 1, 2), feats, num_queries)
                if out is not None:
                    return out
            x = self._conv_gn_relu(xp.permute(0, 3, 1, 2), self.lay2, self.gn2)
            for conv, gn, feat in ((self.lay3, self.gn3, feats[0]), (self.lay4, self.gn4, feats[1]), (self.lay5, self.gn5, feats[2])):
                # round 6: nearest up-sampling + the broadcast add of the adapter's output in ONE pass (tf_upsample_add_nhwc_f32; the
                # two ATen passes write and re-read the up-sampled tensor: 1.1 GB per 128 queries at the finest level)
                merged = fused.upsample_add(x, feat, num_queries) if _mask_head_fused_tail else None
                x = merged if merged is not None else self._merge(x, feat, num_queries).contiguous(memory_format=torch.channels_last)
                if conv is self.lay5 and _mask_head_fused_tail:
                    # ... and the end of the head -- gn5 + ReLU + out_lay (16 -> 1) -- in one pass over lay5's raw output
                    # (tf_groupnorm_relu_conv3x3_c1_nhwc_f32): the normalised activation is never written, no library convolution
                    # (MIOpen ran a Winograd kernel between two layout transposes here: ~0.65 ms per 128 queries)
                    y = fused.conv3x3(x, self._taps(conv, conv.in_channels), conv.bias, False, 1)
                    out = None if y is None else fused.groupnorm_relu_conv3x3_c1(y, gn, self.out_lay)
                    if out is not None:
                        return out
                x = self._conv_gn_relu(x, conv, gn)
            return self.out_lay(x)
        x = F.relu(self.gn2(self.lay2(x)))
        x = self._merge(x, feats[0], num_queries)
        x = F.relu(self.gn3(self.lay3(x)))
        x = self._merge(x, feats[1], num_queries)
        x = F.relu(self.gn4(self.lay4(x)))
        x = self._merge(x, feats[2], num_queries)
        x = F.relu(self.gn5(self.lay5(x)))
        return self.out_lay(x)


class MHAttentionMap(nn.Module):
    """2-D multi-head attention that returns only the attention weights (softmax over heads x H x W
    jointly, as the reference does at :219), no multiplication by a value."""

    def __init__(self, query_dim, hidden_dim, num_heads, dropout=0.0, bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.hidden_dim = hidden_dim
        self.dropout = nn.Dropout(dropout)
        self.q_linear = nn.Linear(query_dim, hidden_dim, bias=bias)
        self.k_linear = nn.Linear(query_dim, hidden_dim, bias=bias)
        nn.init.zeros_(self.k_linear.bias)
        nn.init.zeros_(self.q_linear.bias)
        nn.init.xavier_uniform_(self.k_linear.weight)
        nn.init.xavier_uniform_(self.q_linear.weight)
        self.normalize_fact = float(hidden_dim / self.num_heads) ** -0.5

    def forward(self, q, k, mask: Optional[Tensor] = None):
        """q [B,Q,C] decoder embeddings, k [B,C,h,w] encoder memory, mask [B,h,w] (True = padding)
        -> [B,Q,heads,h,w]."""
        B, Q = q.shape[:2]
        h, w = k.shape[-2:]
        dh = self.hidden_dim // self.num_heads
        q = self.q_linear(q)
        k = F.conv2d(k, self.k_linear.weight[:, :, None, None], self.k_linear.bias)
        qh = (q * self.normalize_fact).view(B, Q, self.num_heads, dh).transpose(1, 2)   # [B,n,Q,dh]
        kh = k.view(B, self.num_heads, dh, h * w)                                        # [B,n,dh,hw]
        weights = torch.matmul(qh, kh).transpose(1, 2)                                   # [B,Q,n,hw]
        if mask is not None:
            weights = weights.masked_fill(mask.flatten(1)[:, None, None], float("-inf"))
        weights = F.softmax(weights.flatten(2), dim=-1).view(B, Q, self.num_heads, h, w)
        return self.dropout(weights)


class PostProcessSegm(nn.Module):
    """pred_masks -> per-image masks at the original image size: bilinear resize to the padded batch
    size, sigmoid (threshold unless return_probs), crop the padding away, nearest resize."""

    def __init__(self, threshold=0.5):
        super().__init__()
        self.threshold = threshold

    @torch.no_grad()
    def forward(self, results, outputs, orig_target_sizes, max_target_sizes, return_probs=False,
                results_mask=None):
        assert len(orig_target_sizes) == len(max_target_sizes)
        sizes = torch.as_tensor(max_target_sizes).tolist()
        orig = torch.as_tensor(orig_target_sizes).tolist()
        max_h, max_w = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if "pred_masks" not in outputs and "mask_context" in outputs:
            raise KeyError("pred_masks: the detector ran with lazy masks (DETRSegmBase.lazy_masks / lazy_mask_scope) and "
                           "returned `mask_context`; evaluate model.mask_rows(context, hs_rows) or run it without the switch")
        masks_all = outputs["pred_masks"]
        if masks_all.dim() == 5:
            masks_all = masks_all.squeeze(2)
        for i, ((img_h, img_w), out_size) in enumerate(zip(sizes, orig)):
            cur = masks_all[i]
            if results_mask is not None:   # resize only what is kept (the reference filters last)
                cur = cur[results_mask[i]]
            cur = F.interpolate(cur[None], size=(max_h, max_w), mode="bilinear",
                                align_corners=False)[0].sigmoid()
            if not return_probs:
                cur = cur > self.threshold
            cur = F.interpolate(cur[:, :img_h, :img_w].unsqueeze(1).float(), size=tuple(out_size),
                                mode="nearest")
            results[i]["masks"] = cur if return_probs else cur.byte()
        return results


class PostProcessPanoptic(nn.Module):
    """Model outputs -> COCO-panoptic predictions (`png_string`, `segments_info`), reference :266-388.
    Needs `panopticapi` and PIL at call time (neither is a dependency of the tracking path)."""

    def __init__(self, is_thing_map, threshold=0.85):
        super().__init__()
        self.threshold = threshold
        self.is_thing_map = is_thing_map

    def forward(self, outputs, processed_sizes, target_sizes=None):
        from panopticapi.utils import id2rgb, rgb2id   # noqa: import error = missing optional dep
        from PIL import Image
        import numpy as np

        if target_sizes is None:
            target_sizes = processed_sizes
        assert len(processed_sizes) == len(target_sizes)
        out_logits, raw_masks, raw_boxes = \
            outputs["pred_logits"], outputs["pred_masks"], outputs["pred_boxes"]
        assert len(out_logits) == len(raw_masks) == len(target_sizes)
        no_object = out_logits.shape[-1] - 1

        def as_tuple(t):
            return t if isinstance(t, tuple) else tuple(torch.as_tensor(t).cpu().tolist())

        preds = []
        for logits, masks, boxes, size, target_size in zip(out_logits, raw_masks, raw_boxes,
                                                           processed_sizes, target_sizes):
            scores, classes = logits.softmax(-1).max(-1)
            keep = classes.ne(no_object) & (scores > self.threshold)
            scores, classes = scores[keep], classes[keep]
            masks = F.interpolate(masks[keep][None], size=as_tuple(size), mode="bilinear",
                                  align_corners=False)[0]
            boxes = box_ops.box_cxcywh_to_xyxy(boxes[keep])
            h, w = masks.shape[-2:]
            assert len(boxes) == len(classes)
            masks = masks.flatten(1)

            # several predicted masks of one stuff class are merged into the first of them
            stuff_equiv = defaultdict(list)
            for k, label in enumerate(classes.tolist()):
                if not self.is_thing_map[label]:
                    stuff_equiv[label].append(k)
            final_h, final_w = as_tuple(target_size)

            def ids_and_areas(cur_masks, n, dedup=False):
                if cur_masks.shape[0] == 0:
                    m_id = torch.zeros((h, w), dtype=torch.long, device=cur_masks.device)
                else:
                    m_id = cur_masks.transpose(0, 1).softmax(-1).argmax(-1).view(h, w)
                if dedup:
                    for equiv in stuff_equiv.values():
                        for eq_id in equiv[1:]:
                            m_id.masked_fill_(m_id.eq(eq_id), equiv[0])
                seg_img = Image.fromarray(id2rgb(m_id.cpu().numpy()))
                seg_img = seg_img.resize(size=(final_w, final_h), resample=Image.NEAREST)
                ids = torch.from_numpy(rgb2id(np.array(seg_img)))
                return [int(ids.eq(i).sum()) for i in range(n)], seg_img

            area, seg_img = ids_and_areas(masks, len(scores), dedup=True)
            if classes.numel() > 0:
                while True:   # drop empty (<= 4 px) segments until none is left
                    small = torch.as_tensor([a <= 4 for a in area], dtype=torch.bool,
                                            device=keep.device)
                    if not small.any():
                        break
                    scores, classes, masks = scores[~small], classes[~small], masks[~small]
                    area, seg_img = ids_and_areas(masks, len(scores))
            else:
                classes = torch.ones(1, dtype=torch.long, device=classes.device)

            segments_info = [{"id": i, "isthing": self.is_thing_map[int(classes[i])],
                              "category_id": int(classes[i]), "area": a}
                             for i, a in enumerate(area)]
            with io.BytesIO() as buf:
                seg_img.save(buf, format="PNG")
                preds.append({"png_string": buf.getvalue(), "segments_info": segments_info})
        return preds
