"""Makes `import MultiScaleDeformableAttention` resolve to the MI355X implementation.

Two forms of the same module (ms_deform_attn_forward / ms_deform_attn_backward with the reference's signatures,
models/ops/src/vision.cpp:4-7):
  * MultiScaleDeformableAttention.py in this directory: Python over ctypes (trackformer_amd/msda.py) -- install()
  * compiled/MultiScaleDeformableAttention.<abi>.so: a pybind11 torch extension over the C ABI of libtf_msda.so, the form the
    reference ships its plugin in (csrc/msda_ext.cpp, built by build_ext.py) -- install(compiled=True)
"""
import importlib
import os
import sys


def install(compiled=False):
    """Put the module's directory on sys.path (front) and import it.  compiled=True: the compiled extension (built on first use)."""
    here = os.path.dirname(os.path.abspath(__file__))
    if compiled:
        from . import build_ext
        build_ext.build()
        here = build_ext.OUT_DIR
    if here in sys.path:            # to the FRONT, also when the other form's directory was installed in between
        sys.path.remove(here)
    sys.path.insert(0, here)
    sys.modules.pop("MultiScaleDeformableAttention", None)
    return importlib.import_module("MultiScaleDeformableAttention")
