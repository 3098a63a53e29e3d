"""two-stage-gnn message passing, MI355X-native (import as ``two_stage_gnn_amd``).

Hand-written HIP (gfx950) kernels behind a C ABI (include/tsgnn.h, csrc/), called through ctypes by
Python modules that mirror the reference's nn.Module surface for the message-passing path.
"""
__version__ = "0.1.0"
