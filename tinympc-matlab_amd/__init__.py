"""tinympc-matlab_amd: MI355X-native TinyMPC ADMM hot path behind the reference's MEX verb surface.

  problems  -- problem data of the BASELINE.json configurations
  _lib      -- ctypes binding of the C-ABI library libtinympc_hip.so (include/tinympc_hip.h)
  tinympc   -- TinyMPC class: host-side mirror of the reference's MATLAB class (src/TinyMPC.m)
  batch     -- shard bookkeeping for the multi-GPU batched mode
  csrc/     -- the HIP kernels (gfx950) and the C ABI
  matlab/   -- the drop-in TinyMPC.m class and the MEX shim over the C ABI
"""
from . import batch, problems  # noqa: F401
from ._lib import TinyMPCError, abi_version, device_count, load_library  # noqa: F401
from .tinympc import TinyMPC  # noqa: F401

__all__ = ["TinyMPC", "TinyMPCError", "problems", "batch", "load_library", "abi_version", "device_count"]
