"""Makes `import MultiScaleDeformableAttention` resolve to the MI355X implementation."""
import importlib
import os
import sys


def install():
    """Put this directory on sys.path (front) and import the drop-in module."""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    return importlib.import_module("MultiScaleDeformableAttention")
