"""trackformer_amd -- MI355X (gfx950) native implementation of TrackFormer's per-frame
detection / track-query forward path (see DESIGN.md for scope and INTEGRATION.md for the boundary).

The package never falls back to a CPU implementation: every operator entry point raises if the
HIP library (trackformer_amd/lib/libtf_msda.so) is missing or the tensors are not on a GPU.
"""
__version__ = "0.1.0"
