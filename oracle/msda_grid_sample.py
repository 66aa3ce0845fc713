"""oracle/msda_grid_sample.py -- TEST INFRASTRUCTURE ONLY (never imported by trackformer_amd/).

The reference's own pure-CPU MSDeformAttn path, restated: `ms_deform_attn_core_pytorch`
(ops/functions/ms_deform_attn_func.py:34-54) evaluates the operator with `F.grid_sample` on the
per-level value maps (bilinear, zero padding, align_corners=False at grid 2*loc-1 -- the same pixel
mapping x = u*W - 0.5 as the CUDA kernels).  It is what BASELINE.json calls
"the reference's pure-CPU MSDeformAttn path" and what bench.py's cpu_baseline times on the host cores
("kind": "reference-restated"); the C port (oracle/msda_ref.c) stays the parity checker.

Written from the formula, not from the reference text: per level one grid_sample over
[N*M, D, H, W] with a [N*M, Lq, P, 2] grid, weighted by that level's attention weights and accumulated
level by level (the reference stacks all L*P samples and reduces once; the sums differ by fp32
round-off only).  Differentiable (autograd through grid_sample), fp32 / fp64.

Parity status: pinned to the same 9 golden files as the C port (tests/test_oracle.py), which hold
outputs and gradients of the reference's own function.
"""
import torch
import torch.nn.functional as F


def msda_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    """value [N,S,M,D], spatial_shapes [(H,W)]*L, sampling_locations [N,Lq,M,L,P,2] in [0,1],
    attention_weights [N,Lq,M,L,P] -> [N,Lq,M*D]."""
    n, s, m, d = value.shape
    lq, n_levels, n_points = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if torch.is_tensor(spatial_shapes)
                                            else spatial_shapes)]
    assert sum(h * w for h, w in shapes) == s and len(shapes) == n_levels
    heads_first = value.permute(0, 2, 3, 1).reshape(n * m, d, s)          # [N*M, D, S]
    grid_all = (sampling_locations * 2 - 1).permute(0, 2, 1, 3, 4, 5)     # [N, M, Lq, L, P, 2]
    weights = attention_weights.permute(0, 2, 1, 3, 4)                    # [N, M, Lq, L, P]
    out = value.new_zeros(n * m, d, lq)
    start = 0
    for lvl, (h, w) in enumerate(shapes):
        fmap = heads_first[:, :, start:start + h * w].reshape(n * m, d, h, w)
        start += h * w
        grid = grid_all[:, :, :, lvl].reshape(n * m, lq, n_points, 2)
        sampled = F.grid_sample(fmap, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        out = out + (sampled * weights[:, :, :, lvl].reshape(n * m, 1, lq, n_points)).sum(-1)
    return out.reshape(n, m * d, lq).transpose(1, 2).contiguous()


def make_torch_function():
    """Stand-in for trackformer_amd.msda.MSDeformAttnFunction on CPU tensors (`.apply(value, shapes, loc,
    attn, im2col_step)`), used by bench.py's cpu_baseline leg and by tests."""
    class _Fn:
        @staticmethod
        def apply(value, shapes, loc, attn, im2col_step=64):
            return msda_grid_sample(value, shapes, loc, attn)
    return _Fn
