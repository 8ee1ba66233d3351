"""tinympc-matlab_amd: MI355X-native TinyMPC ADMM hot path behind the reference's MEX verb surface.

  problems  -- problem data of the BASELINE.json configurations
  _lib      -- ctypes binding of the C-ABI library (include/tinympc_hip.h)
  tinympc   -- TinyMPC class: Python mirror of the reference's MATLAB class (src/TinyMPC.m)
  batch     -- batched verbs + the multi-GPU sharding helpers
"""
from . import problems  # noqa: F401
