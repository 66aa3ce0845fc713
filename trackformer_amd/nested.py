"""Padded-batch container and small numeric helpers used across the forward path.

Mirrors the pieces of the reference's util/misc.py that the hot path needs:
NestedTensor / nested_tensor_from_tensor_list (util/misc.py:309-365) and inverse_sigmoid (:515-519).
"""
from typing import List, Optional

import torch
from torch import Tensor


class NestedTensor(object):
    """A batch of images zero-padded to a common size plus a bool mask (True = padding)."""

    def __init__(self, tensors: Tensor, mask: Optional[Tensor] = None):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        mask = None if self.mask is None else self.mask.to(device)
        return NestedTensor(self.tensors.to(device), mask)

    def decompose(self):
        return self.tensors, self.mask

    def unmasked_tensor(self, index: int):
        """Crop sample `index` back to its un-padded extent (util/misc.py:348-362)."""
        tensor = self.tensors[index]
        mask = self.mask[index]
        if not mask.any():
            return tensor
        first_pad_col = mask[0, :].nonzero(as_tuple=True)[0]
        if len(first_pad_col):
            tensor = tensor[:, :, :first_pad_col[0]]
        first_pad_row = mask[:, 0].nonzero(as_tuple=True)[0]
        if len(first_pad_row):
            tensor = tensor[:, :first_pad_row[0], :]
        return tensor

    def __repr__(self):
        return str(self.tensors)


ALL_VALID_ATTR = "_tf_all_valid"  # host-side knowledge: this padding mask is all False
_all_valid_masks = {}


def all_valid_mask(shape, device) -> Tensor:
    """A cached all-False padding mask, tagged so that consumers can skip mask-dependent work
    (valid ratios, position encodings, masked_fill) without a device->host check."""
    key = (tuple(shape), torch.device(device))
    m = _all_valid_masks.get(key)
    if m is None:
        m = torch.zeros(tuple(shape), dtype=torch.bool, device=device)
        setattr(m, ALL_VALID_ATTR, True)
        _all_valid_masks[key] = m
    return m


def is_all_valid(mask) -> bool:
    return mask is not None and bool(getattr(mask, ALL_VALID_ATTR, False))


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]) -> NestedTensor:
    """list of [C,h,w] images (or a [B,C,H,W] tensor) -> zero padded batch + padding mask."""
    if tensor_list[0].ndim != 3:
        raise ValueError('not supported')
    if isinstance(tensor_list, Tensor) or len({tuple(t.shape) for t in tensor_list}) == 1:
        # all the same size (always true at batch 1): no padding, no per-image copies
        batch = tensor_list if isinstance(tensor_list, Tensor) else torch.stack(list(tensor_list))
        b, _, h, w = batch.shape
        return NestedTensor(batch, all_valid_mask((b, h, w), batch.device))
    c = max(t.shape[0] for t in tensor_list)
    h = max(t.shape[1] for t in tensor_list)
    w = max(t.shape[2] for t in tensor_list)
    first = tensor_list[0]
    batch = torch.zeros((len(tensor_list), c, h, w), dtype=first.dtype, device=first.device)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool, device=first.device)
    for i, img in enumerate(tensor_list):
        batch[i, :img.shape[0], :img.shape[1], :img.shape[2]].copy_(img)
        mask[i, :img.shape[1], :img.shape[2]] = False
    return NestedTensor(batch, mask)


def inverse_sigmoid(x: Tensor, eps: float = 1e-5) -> Tensor:
    """logit with the reference's clamping: log(clamp(x,eps..1) / clamp(1-x,eps..))."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))
