"""stove_amd -- MI355X (gfx950) native implementation of STOVE's per-frame hot path.

Python modules mirror the reference's module/API surface (model.spn.*, model.video_prediction.*);
the SPN sweep, the scene (glimpse/mask) stage and the GNN dynamics run as hand-written HIP
kernels in libstove_hip.so (C ABI: include/stove_hip.h).  There is no CPU fallback: the ops
raise if the library or a GPU is missing.
"""
__version__ = '0.1.0'
