"""Sine (2-D and 3-D multi-frame) and learned position encodings.

Same classes / arguments / outputs as models/position_encoding.py of the reference
(PositionEmbeddingSine :84-120, PositionEmbeddingSine3D :12-81, PositionEmbeddingLearned :123-148,
build_position_encoding :151-169).  The sine encodings have no parameters and depend only on the
padding mask; when the mask is known to be all-False (same-size batches, always the case for
batch-1 tracking) the result is cached per (batch, H, W, device) instead of being recomputed with
~25 elementwise kernels per level and frame.
"""
import math

import torch
from torch import nn

from .nested import NestedTensor, is_all_valid


def _sincos_interleave(coord, num_pos_feats, temperature):
    """coord[...] -> [..., num_pos_feats]: sin on even channels, cos on odd, shared frequencies."""
    i = torch.arange(num_pos_feats, dtype=torch.float32, device=coord.device)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
    ang = coord[..., None] / dim_t
    return torch.stack((ang[..., 0::2].sin(), ang[..., 1::2].cos()), dim=-1).flatten(-2)


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def _compute(self, mask):
        not_mask = ~mask
        y = not_mask.cumsum(1, dtype=torch.float32)
        x = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y = (y - 0.5) / (y[:, -1:, :] + eps) * self.scale
            x = (x - 0.5) / (x[:, :, -1:] + eps) * self.scale
        pos_x = _sincos_interleave(x, self.num_pos_feats, self.temperature)
        pos_y = _sincos_interleave(y, self.num_pos_feats, self.temperature)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)

    def forward(self, tensor_list: NestedTensor):
        mask = tensor_list.mask
        assert mask is not None
        if is_all_valid(mask) and not torch.is_grad_enabled():
            key = (tuple(mask.shape), mask.device)
            pos = self._cache.get(key)
            if pos is None:
                pos = self._cache[key] = self._compute(mask)
                pos._tf_cached_geometry = True   # one tensor per geometry: what downstream per-geometry caches may key on
            return pos
        return self._compute(mask)


class PositionEmbeddingSine3D(nn.Module):
    """(frame, y, x) sine encoding for multi-frame attention: output [N, frames, 3*F, H, W]."""

    def __init__(self, num_pos_feats=64, num_frames=2, temperature=10000, normalize=False,
                 scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.frames = num_frames
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def _compute(self, mask):
        n, h, w = mask.shape
        not_mask = ~mask.view(n, 1, h, w).expand(n, self.frames, h, w)
        z = not_mask.cumsum(1, dtype=torch.float32)
        y = not_mask.cumsum(2, dtype=torch.float32)
        x = not_mask.cumsum(3, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6  # NB: no -0.5 shift in the 3-D variant (position_encoding.py:56-58)
            z = z / (z[:, -1:, :, :] + eps) * self.scale
            y = y / (y[:, :, -1:, :] + eps) * self.scale
            x = x / (x[:, :, :, -1:] + eps) * self.scale
        pos = [_sincos_interleave(c, self.num_pos_feats, self.temperature) for c in (z, y, x)]
        return torch.cat(pos, dim=4).permute(0, 1, 4, 2, 3)

    def forward(self, tensor_list: NestedTensor):
        mask = tensor_list.mask
        assert mask is not None
        if is_all_valid(mask) and not torch.is_grad_enabled():
            key = (tuple(mask.shape), mask.device)
            pos = self._cache.get(key)
            if pos is None:
                pos = self._cache[key] = self._compute(mask)
                pos._tf_cached_geometry = True   # one tensor per geometry: what downstream per-geometry caches may key on
            return pos
        return self._compute(mask)


class PositionEmbeddingLearned(nn.Module):
    """Learned absolute encoding: 50 row + 50 column embeddings."""

    def __init__(self, num_pos_feats=256):
        super().__init__()
        self.row_embed = nn.Embedding(50, num_pos_feats)
        self.col_embed = nn.Embedding(50, num_pos_feats)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def forward(self, tensor_list: NestedTensor):
        x = tensor_list.tensors
        h, w = x.shape[-2:]
        x_emb = self.col_embed(torch.arange(w, device=x.device))  # [w, F]
        y_emb = self.row_embed(torch.arange(h, device=x.device))  # [h, F]
        pos = torch.cat([x_emb[None].expand(h, -1, -1), y_emb[:, None].expand(-1, w, -1)], dim=-1)
        return pos.permute(2, 0, 1)[None].repeat(x.shape[0], 1, 1, 1)


def build_position_encoding(args):
    if args.multi_frame_attention and args.multi_frame_encoding:
        n_steps = args.hidden_dim // 3
        sine_cls = PositionEmbeddingSine3D
    else:
        n_steps = args.hidden_dim // 2
        sine_cls = PositionEmbeddingSine
    if args.position_embedding in ('v2', 'sine'):
        return sine_cls(n_steps, normalize=True)
    if args.position_embedding in ('v3', 'learned'):
        return PositionEmbeddingLearned(n_steps)
    raise ValueError(f"not supported {args.position_embedding}")
