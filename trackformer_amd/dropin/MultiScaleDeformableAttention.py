"""Drop-in replacement for the reference's native extension module `MultiScaleDeformableAttention`.

The reference builds a pybind11/CUDA extension of this name (ops/setup.py:30-66, ops/src/vision.cpp:4-7)
whose only consumer is `import MultiScaleDeformableAttention as MSDA` in
ops/functions/ms_deform_attn_func.py:11.  Putting this directory on sys.path (or installing this
file as a top-level module) makes that import resolve to the MI355X implementation without touching
the reference sources:

    import trackformer_amd.dropin; trackformer_amd.dropin.install()
    # ... reference code: MSDA.ms_deform_attn_forward(value, shapes, loc, weights, im2col_step)

Both functions keep the reference signatures and return types (Tensor, list of 3 Tensors).
"""
from trackformer_amd.msda import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward"]
